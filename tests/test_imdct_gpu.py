"""GPU parity of vpz_imdct_batch (== Mdct.Reverse, Mdct.cs:15-19) against the CPU oracle, through
the C ABI.  Tolerance for the FAST path is BASELINE.json's: <= 1e-5 max-abs per sample for
|PCM| <~ 1 (sigma = 2^-8 spectra); the EXACT path must be bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import Context
    c = Context(0)
    yield c
    c.close()


def spectra(count, half, seed):
    return (np.random.default_rng(seed).standard_normal((count, half)) * 2.0 ** -8).astype(np.float32)


@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096, 8192])
@pytest.mark.parametrize("count", [1, 2, 5, 37, 64, 513])
def test_fast_matches_oracle(ctx, oracle, n, count):
    from vorbispizza_amd import capi
    x = spectra(count, n // 2, n + count)
    ref = oracle.mdct_reverse(x, n)
    got = ctx.imdct_batch(x, n, capi.IMDCT_FAST)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= TOL
    # the mirrored halves are exact copies, as in Mdct.cs:378-381
    h = n // 2
    assert np.array_equal(got[:, :h], -got[:, :h][:, ::-1])
    assert np.array_equal(got[:, h:], got[:, h:][:, ::-1])


@pytest.mark.parametrize("n", [64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_exact_is_bit_identical(ctx, oracle, n):
    from vorbispizza_amd import capi
    x = spectra(9, n // 2, n)
    x[0] = 0
    x[1, ::2] = -x[1, ::2]
    ref = oracle.mdct_reverse(x, n)
    got = ctx.imdct_batch(x, n, capi.IMDCT_EXACT)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("n", [64, 128])
def test_fast_mode_falls_back_to_exact_for_other_sizes(ctx, oracle, n):
    from vorbispizza_amd import capi
    x = spectra(3, n // 2, n)
    got = ctx.imdct_batch(x, n, capi.IMDCT_FAST)
    assert np.array_equal(got.view(np.uint32), oracle.mdct_reverse(x, n).view(np.uint32))


def test_large_amplitude_relative_error(ctx, oracle):
    """Unit-variance spectra give |y| up to ~70: the error must scale with the signal."""
    from vorbispizza_amd import capi
    x = np.random.default_rng(9).standard_normal((16, 1024)).astype(np.float32)
    ref = oracle.mdct_reverse(x, 2048)
    got = ctx.imdct_batch(x, 2048, capi.IMDCT_FAST)
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()


def test_empty_and_invalid(ctx):
    from vorbispizza_amd import SynthError, capi
    out = ctx.imdct_batch(np.zeros((0, 1024), dtype=np.float32), 2048)
    assert out.shape == (0, 2048)
    for bad in (0, 32, 100, 16384):
        with pytest.raises(SynthError) as e:
            ctx.imdct_batch(np.zeros((1, max(bad // 2, 1)), dtype=np.float32), bad)
        assert e.value.status == capi.E_UNSUPPORTED


def test_special_values_propagate(ctx):
    """NaN / Inf in a row stay in that row (no cross-row contamination)."""
    x = spectra(8, 1024, 5)
    x[3, 17] = np.nan
    got = ctx.imdct_batch(x, 2048)
    assert np.isnan(got[3]).all()
    assert np.isfinite(np.delete(got, 3, axis=0)).all()


def test_device_memory_path_full_size(ctx, oracle):
    """BASELINE config 2 shape on device-resident tensors: 131 072 channel-blocks of N = 2048.
    Checked by linearity-free properties at full size (mirror symmetry, checksum of a seeded
    subset against the oracle)."""
    import torch
    from vorbispizza_amd import capi
    count = 131072
    g = torch.Generator(device="cuda").manual_seed(2048)
    x = torch.randn((count, 1024), generator=g, device="cuda", dtype=torch.float32) * 2.0 ** -8
    y = ctx.imdct_batch(x, 2048, capi.IMDCT_FAST)
    ctx.synchronize()
    assert torch.equal(y[:, :1024], -torch.flip(y[:, :1024], dims=[1]))
    assert torch.equal(y[:, 1024:], torch.flip(y[:, 1024:], dims=[1]))
    idx = torch.tensor([0, 1, 2, 3, 65535, 65536, 99999, 131070, 131071], device="cuda")
    ref = oracle.mdct_reverse(x[idx].cpu().numpy(), 2048)
    assert np.abs(y[idx].cpu().numpy() - ref).max() <= TOL
    # Parseval-type check over the whole batch: sum y^2 = N/2 * 2 * sum X^2 / ... (IMDCT of an
    # orthogonal-up-to-scale basis): energy ratio is N/2 for every row
    ex = (x.double() ** 2).sum(dim=1)
    ey = (y.double() ** 2).sum(dim=1)
    ratio = (ey / ex).cpu().numpy()
    assert np.abs(ratio / 1024.0 - 1).max() < 1e-4
