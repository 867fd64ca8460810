"""BASELINE.json's full sizes (configs[2], configs[3]) through size-independent properties, device memory all
the way: exact homogeneity (x2 in -> x2 out, bit for bit: every operation on the path is linear in the scale
and powers of two are exact), independence of the batch split, sample counts, and a prefix checked against the
oracle."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as ge
    ge.build()
    import torch
    from vorbispizza_amd import Context
    c = Context(0)
    yield c, torch
    c.close()


def _run(ctx, torch, dec, pk, res, posts, counts, samples, channels, splits=()):
    from vorbispizza_amd import capi
    cap = samples + 1024
    out = torch.zeros(channels * cap, device=res.device, dtype=torch.float32)
    dec.reset(-1)
    dec.set_position(0)
    bounds = [0] + list(splits) + [len(pk)]
    done = 0
    for a, b in zip(bounds[:-1], bounds[1:]):
        sub = pk[a:b]
        p_ = None if posts is None else posts[a * channels:b * channels]
        c_ = None if counts is None else counts[a * channels:b * channels]
        # every call writes at the stream's running offset: hand it the tail of the buffer
        view = out[done:]
        w = dec.synth_raw(sub, res, p_, c_, view, None, cap - done, capi.OUT_PLANAR, cap, capi.MEM_DEVICE)
        done += int(w[0])
    ctx.synchronize()
    assert done == samples
    return out.reshape(channels, cap)[:, :samples]


def test_config2_full_size_properties(env, oracle):
    ctx, torch = env
    import bench
    from vorbispizza_amd import Decoder
    dev = torch.device("cuda", 0)
    pk, res, samples, _ = bench.build_synth_ola(torch, dev, 65536)
    dec = Decoder(ctx, 2, 256, 2048)
    y = _run(ctx, torch, dec, pk, res, None, None, samples, 2)
    assert torch.isfinite(y).all() and float(y.abs().max()) < 1.0
    # x2 in -> x2 out, exactly
    y2 = _run(ctx, torch, dec, pk, res * 2.0, None, None, samples, 2)
    assert torch.equal(y2, y * 2.0)
    # the batch split does not matter (state carried across calls == one call)
    y3 = _run(ctx, torch, dec, pk, res, None, None, samples, 2, splits=(1, 30001, 30002, 50000))
    assert torch.equal(y3, y)
    # prefix against the oracle
    n = 300
    opk = [{"flags": int(pk["flags"][f]), "granule": -1, "residue": res[int(pk["residue_offset"][f]):
            int(pk["residue_offset"][f]) + 2 * (1024 if pk["flags"][f] & 1 else 128)].cpu().numpy()} for f in range(n)]
    ref, _, _ = helpers.oracle_decode(oracle, 2, 256, 2048, opk)
    got = y[:, :ref.shape[1]].cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-5
    dec.close()


def test_config3_full_size_properties(env, oracle):
    ctx, torch = env
    import bench
    from vorbispizza_amd import Decoder
    dev = torch.device("cuda", 0)
    pk, res, posts, counts, floors, mappings, samples = bench.build_floor6(torch, dev, 16384)
    dec = Decoder(ctx, 6, 256, 2048, floors=floors, mappings=mappings)
    y = _run(ctx, torch, dec, pk, res, posts, counts, samples, 6)
    assert torch.isfinite(y).all()
    y2 = _run(ctx, torch, dec, pk, res * 2.0, posts, counts, samples, 6)
    assert torch.equal(y2, y * 2.0)
    y3 = _run(ctx, torch, dec, pk, res, posts, counts, samples, 6, splits=(5, 8000, 8001))
    assert torch.equal(y3, y)
    n = 40
    hp, hc = posts[: n * 6].cpu().numpy(), counts[: n * 6].cpu().numpy()
    opk = [{"flags": int(pk["flags"][f]), "granule": -1, "mapping": 0,
            "residue": res[f * 6144:(f + 1) * 6144].cpu().numpy(), "posts": hp[f * 6:(f + 1) * 6],
            "post_count": hc[f * 6:(f + 1) * 6]} for f in range(n)]
    ref, _, _ = helpers.oracle_decode(oracle, 6, 256, 2048, opk, floors=floors, mappings=mappings)
    got = y[:, :ref.shape[1]].cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max()))
    # the same batch as `ReadSamples(Span<float>)` delivers it: interleaved, written by a packet's six waves together --
    # the planar result transposed, bit for bit; and as 16-bit samples, the reference conversion of those floats
    from vorbispizza_amd import capi
    cap = samples + 1024
    for layout, dtype in ((capi.OUT_INTERLEAVED, torch.float32), (capi.OUT_INTERLEAVED_S16, torch.int16)):
        out = torch.zeros(6 * cap, device=dev, dtype=dtype)
        dec.reset(-1)
        dec.set_position(0)
        w = dec.synth_raw(pk, res, posts, counts, out, None, cap, layout, 0, capi.MEM_DEVICE)
        ctx.synchronize()
        assert int(w[0]) == samples
        inter = out[: samples * 6].reshape(samples, 6).t()
        if dtype == torch.float32:
            assert torch.equal(inter.contiguous().view(torch.int32), y.contiguous().view(torch.int32))
        else:
            want = torch.clamp((y * 32768.0).to(torch.int32), -32768, 32767).to(torch.int16)
            assert torch.equal(inter, want)
    dec.close()
