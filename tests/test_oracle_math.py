"""Pins the CPU oracle (oracle/) with mathematics, since the reference holds no golden vectors
for this path: closed-form float64 IMDCT, TDAC reconstruction, output
symmetries, window power complementarity, packet geometry table, coupling truth table, Floor1 DDA
against its closed form."""
import numpy as np
import pytest


def imdct_closed_form(X):
    """y[n] = sum_k X[k] cos(pi/(2N) (2n+1+N/2)(2k+1)), float64 (SURVEY.md a-M)."""
    X = np.asarray(X, dtype=np.float64)
    n2 = X.shape[-1]
    N = 2 * n2
    n = np.arange(N)[:, None]
    k = np.arange(n2)[None, :]
    M = np.cos(np.pi / (2 * N) * (2 * n + 1 + N / 2) * (2 * k + 1))
    return X @ M.T


@pytest.mark.parametrize("N", [256, 512, 1024, 2048, 4096, 8192])
def test_mdct_matches_closed_form(oracle, N):
    rng = np.random.default_rng(N)
    X = rng.standard_normal((3, N // 2)).astype(np.float32)
    y = oracle.mdct_reverse(X, N).astype(np.float64)
    ref = imdct_closed_form(X)
    rel = np.abs(y - ref).max() / np.abs(ref).max()
    assert rel < 5e-7, rel


@pytest.mark.parametrize("N", [64, 128])
def test_mdct_small_blocks_reproduce_reference_quirk(oracle, N):
    """Quirk q1: the reference's step 3 runs too many passes for N < 256 (Mdct.cs:202-209,245),
    so its output is NOT the IMDCT there.  The oracle reproduces that literally."""
    rng = np.random.default_rng(N)
    X = rng.standard_normal((2, N // 2)).astype(np.float32)
    y = oracle.mdct_reverse(X, N).astype(np.float64)
    ref = imdct_closed_form(X)
    assert np.all(np.isfinite(y))
    assert np.abs(y - ref).max() / np.abs(ref).max() > 0.1


@pytest.mark.parametrize("N", [256, 2048])
def test_mdct_output_symmetries(oracle, N):
    """Mdct.cs:378-381: y[n] = -y[N/2-1-n] and y[N/2+n] = y[N-1-n], exactly (mirrored stores)."""
    rng = np.random.default_rng(7)
    y = oracle.mdct_reverse(rng.standard_normal((4, N // 2)).astype(np.float32), N)
    h = N // 2
    assert np.array_equal(y[:, :h], -y[:, :h][:, ::-1])
    assert np.array_equal(y[:, h:], y[:, h:][:, ::-1])


@pytest.mark.parametrize("N", [256, 2048])
def test_mdct_linearity_and_impulse(oracle, N):
    rng = np.random.default_rng(11)
    a = rng.standard_normal((1, N // 2)).astype(np.float32)
    b = rng.standard_normal((1, N // 2)).astype(np.float32)
    ya, yb, yab = (oracle.mdct_reverse(v, N) for v in (a, b, a + b))
    assert np.abs((ya + yb) - yab).max() < 2e-4 * np.abs(yab).max()
    e = np.zeros((1, N // 2), dtype=np.float32)
    e[0, 5] = 1.0
    y = oracle.mdct_reverse(e, N)[0]
    n = np.arange(N)
    assert np.abs(y - np.cos(np.pi / (2 * N) * (2 * n + 1 + N / 2) * 11)).max() < 2e-6


def test_window_slope_power_complementary(oracle):
    for half in (128, 1024):
        s = oracle.window_slope(half).astype(np.float64)
        assert np.abs(s ** 2 + s[::-1] ** 2 - 1).max() < 1e-6
        x = np.arange(half)
        ref = np.sin(0.5 * np.pi * np.sin(0.5 * np.pi * (x + 0.5) / half) ** 2)
        assert np.abs(s - ref).max() < 2e-7


def test_packet_info_table(oracle):
    """Mode.cs:30-66 for 256/2048 (SURVEY.md a-G table)."""
    t = lambda i: (i.LeftStart, i.LeftEnd, i.RightStart, i.RightEnd, i.Length, i.LeftUseSize1, i.SampleCount)
    assert t(oracle.packet_info(256, 2048, 0)) == (0, 128, 128, 256, 128, 0, 128)
    assert t(oracle.packet_info(256, 2048, 0, False, False)) == (0, 128, 128, 256, 128, 0, 128)
    assert t(oracle.packet_info(256, 2048, 1, True, True)) == (0, 1024, 1024, 2048, 1024, 1, 1024)
    assert t(oracle.packet_info(256, 2048, 1, False, True)) == (448, 576, 1024, 2048, 128, 0, 576)
    assert t(oracle.packet_info(256, 2048, 1, True, False)) == (0, 1024, 1472, 1600, 1024, 1, 1472)
    assert t(oracle.packet_info(256, 2048, 1, False, False)) == (448, 576, 1472, 1600, 128, 0, 1024)


def test_coupling_truth_table(oracle):
    """Mapping.cs:235-268; the Vector<T> branch (:205-233) differs only in the sign of zero."""
    m = np.array([2.0, 2.0, -2.0, -2.0, 0.0, -0.0, 3.0, -3.0], dtype=np.float32)
    a = np.array([0.5, -0.5, 0.5, -0.5, 1.0, -1.0, 0.0, 0.0], dtype=np.float32)
    for vf in (True, False):
        nm, na = oracle.apply_coupling(m, a, vf)
        assert list(nm) == [2.0, 1.5, -2.0, -1.5, 0.0, 1.0, 3.0, -3.0]
        assert list(na) == [1.5, 2.0, -1.5, -2.0, 1.0, 0.0, 3.0, -3.0]
    # sign-of-zero difference: M=-0, A<=0 -> newA = M + (+0) = +0 in the vector branch, -0 in the scalar one
    _, na_v = oracle.apply_coupling([-0.0], [-1.0], True)
    _, na_s = oracle.apply_coupling([-0.0], [-1.0], False)
    assert np.signbit(na_s[0]) and not np.signbit(na_v[0])


def test_tdac_reconstruction(oracle):
    """Forward MDCT of windowed overlapping blocks -> oracle IMDCT + OverlapBuffers returns
    (N/4) * x."""
    N = 256
    h = N // 2
    rng = np.random.default_rng(5)
    frames = 6
    x = rng.standard_normal(h * (frames + 1))
    s = oracle.window_slope(h).astype(np.float64)
    w = np.concatenate([s, s[::-1]])
    n = np.arange(N)[:, None]
    k = np.arange(h)[None, :]
    M = np.cos(np.pi / (2 * N) * (2 * n + 1 + N / 2) * (2 * k + 1))
    spectra = np.stack([(w * x[f * h:f * h + N]) @ M for f in range(frames)])
    spectra = (spectra / (N / 4)).astype(np.float32)
    flags = np.zeros(frames, dtype=np.uint8)  # all short blocks
    pcm = oracle.synth_stream_planar(1, N, 2048, flags, np.pad(spectra, ((0, 0), (0, 1024 - h)))[:, None, :])
    assert pcm.shape == (1, (frames - 1) * h)
    assert np.abs(pcm[0] - x[h:h * frames]).max() < 5e-6 * np.abs(x).max() * 8


def test_floor1_dda_matches_closed_form(oracle):
    """Floor1.cs:372-397 DDA == y0 + trunc(dy*k/adx) (C truncating division), and RenderPoint."""
    rng = np.random.default_rng(3)
    xl = [0, 1024, 93, 23, 372, 6, 46, 186, 750, 14, 33, 65, 130, 260, 556, 3, 10, 18, 28, 39, 55,
          79, 111, 158, 220, 312, 464, 650, 850]
    f = oracle.floor1_init(xl, 2)
    db = oracle.inverse_db_table().astype(np.float64)
    order = np.argsort(xl)
    for _ in range(50):
        final_y = rng.integers(0, 128, size=len(xl))
        flags = (rng.random(len(xl)) < 0.7).astype(np.uint8)
        flags[:2] = 1
        res = np.ones(1024, dtype=np.float32)
        import ctypes as C
        fy = np.zeros(64, dtype=np.int32); fy[:len(xl)] = final_y
        fl = np.zeros(64, dtype=np.uint8); fl[:len(xl)] = flags
        oracle.lib().orc_floor1_render(C.byref(f), fy.ctypes.data_as(C.POINTER(C.c_int)),
                                       fl.ctypes.data_as(C.POINTER(C.c_uint8)), len(xl), 1024,
                                       res.ctypes.data_as(C.POINTER(C.c_float)))
        act = [i for i in order if flags[i]]
        curve = np.zeros(1024, dtype=np.int64)
        for a, b in zip(act[:-1], act[1:]):
            x0, x1 = xl[a], min(xl[b], 1024)
            y0, y1 = final_y[a] * 2, final_y[b] * 2
            kk = np.arange(x1 - x0)
            curve[x0:x1] = y0 + np.sign(y1 - y0) * ((abs(y1 - y0) * kk) // (x1 - x0))
        assert np.array_equal(res.astype(np.float64), db[curve].astype(np.float32).astype(np.float64))


def test_floor1_unwrap_posts_example(oracle):
    """Hand-checked amplitude value synthesis (Floor1.cs:287-350): a zero post keeps the
    prediction and clears its step flag; a nonzero post sets both neighbours' flags."""
    f = oracle.floor1_init([0, 128, 64, 32], 2)  # range 128
    posts, flags = oracle.floor1_unwrap(f, [10, 50, 0, 3], 4)
    # post2: predicted = 10 + (40*64)//128 = 30, val 0 -> 30, flag false
    # post3 (x=32, neighbours 0 and 2): predicted = 10 + (20*32)//64 = 20; val 3 odd -> 20 - 2 = 18
    assert list(posts[:4]) == [10, 50, 30, 18]
    assert list(flags[:4]) == [1, 1, 1, 1]  # post3 nonzero re-flags neighbour 2
    posts, flags = oracle.floor1_unwrap(f, [10, 50, 0, 0], 4)
    assert list(posts[:4]) == [10, 50, 30, 20] and list(flags[:4]) == [1, 1, 0, 0]


def test_clip_value(oracle):
    import ctypes as C
    c = C.c_int(0)
    L = oracle.lib()
    up = np.float32(0.99999994)
    assert L.orc_clip_value(1.0, C.byref(c)) == up and c.value == 1
    c = C.c_int(0)
    assert L.orc_clip_value(float(up), C.byref(c)) == up and c.value == 0
    assert L.orc_clip_value(-2.0, C.byref(c)) == -up and c.value == 1
    c = C.c_int(0)
    assert np.isnan(L.orc_clip_value(float("nan"), C.byref(c))) and c.value == 0
