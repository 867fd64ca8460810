"""The HIP path against the committed golden vectors (tests/golden/*.npz) -- no oracle code involved: the vectors
are data.  Float outputs within BASELINE's 1e-5, integers exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import Context
    c = Context(0)
    yield c
    c.close()


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def test_imdct_vectors(ctx):
    from vorbispizza_amd import capi
    v = load("imdct_vectors.npz")
    for n in (256, 2048):
        x, y = v["spectra_%d" % n], v["pcm_%d" % n]
        exact = ctx.imdct_batch(x, n, capi.IMDCT_EXACT)
        assert np.array_equal(exact.view(np.uint32), y.view(np.uint32))        # the reference's own schedule: same bits
        fast = ctx.imdct_batch(x, n, capi.IMDCT_FAST)
        assert np.abs(fast - y).max() <= 1e-5 * max(1.0, float(np.abs(y).max()))  # unit-variance spectra: |y| up to ~72


def test_window_ola_sequence(ctx):
    from vorbispizza_amd import Decoder, capi, make_packets
    v = load("window_ola_sequence.npz")
    flags, spectra, pcm = v["flags"], v["spectra"], v["pcm"]
    pk = make_packets(len(flags))
    res, off = [], 0
    for f in range(len(flags)):
        half = 1024 if flags[f] & 1 else 128
        pk[f]["flags"], pk[f]["granule"], pk[f]["residue_offset"] = flags[f] | capi.PKT_NO_FLOOR, -1, off
        res.append(spectra[f, :, :half].reshape(-1))
        off += 2 * half
    for layout in (capi.OUT_PLANAR, capi.OUT_INTERLEAVED):
        dec = Decoder(ctx, 2, 256, 2048)
        got = dec.synth(pk, np.concatenate(res), out_layout=layout)[0]
        got = got if layout == capi.OUT_PLANAR else got.T
        assert got.shape == pcm.shape and np.abs(got - pcm).max() <= 1e-5
        counts = dec.last_packet_samples(len(flags))
        assert list(counts[1:]) == list(v["packet_info"][1:, 4] - v["packet_info"][1:, 2]) and counts[0] == 0
        dec.close()


def test_coupling_quadrants(ctx):
    """Inverse coupling happens inside the synthesis call: a floor of table index 255 (x 1.0) and an IMDCT follow it.
    The IMDCT is linear, so the de-coupled vectors are recovered by comparing with the synthesis of the expected
    vectors handed over uncoupled -- the same bits, since everything after the coupling is the same arithmetic."""
    from vorbispizza_amd import Decoder, capi, make_packets
    v = load("coupling_quadrants.npz")
    m, a = v["magnitude"], v["angle"]
    n = len(m)
    flat = ([0, 128], 1)  # two posts at y = 255: the curve is table[255] = 1.0 everywhere
    posts = np.zeros((2, 64), dtype=np.int16)
    posts[:, :2] = 255
    counts = np.array([2, 2], dtype=np.uint8)

    def synth(ch0, ch1, coupling):
        dec = Decoder(ctx, 2, 256, 2048, floors=[flat], mappings=[{"coupling": coupling, "channel_floor": [0, 0]}])
        pk = make_packets(2)
        pk["granule"] = -1
        pk["residue_offset"] = [0, 256]
        res = np.zeros(512, dtype=np.float32)
        for f in range(2):
            res[f * 256: f * 256 + n] = ch0
            res[f * 256 + 128: f * 256 + 128 + n] = ch1
        out = dec.synth(pk, res, np.tile(posts, (2, 1)), np.tile(counts, 2))[0]
        dec.close()
        return out

    got = synth(m, a, [(0, 1)])
    want = synth(v["out_magnitude_vector"], v["out_angle_vector"], [])
    assert got.shape == (2, 128) and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.abs(want).max() > 0


def test_floor1_of_3test(ctx):
    from vorbispizza_amd import Decoder
    v = load("floor1_3test_long.npz")
    xlist, mult = [int(x) for x in v["x_list"]], int(v["multiplier"])
    dec = Decoder(ctx, 1, 256, 2048, floors=[(xlist, mult)], mappings=[{"coupling": [], "channel_floor": [0]}])
    raw = np.zeros((len(v["raw_posts"]), 64), dtype=np.int16)
    raw[:, :v["raw_posts"].shape[1]] = v["raw_posts"]
    n_rec = len(raw)
    curve, final_y, flags, active = dec.debug_floor1_indices(raw, np.full(n_rec, 29, np.uint8), np.zeros(n_rec, np.uint8),
                                                             np.ones(n_rec, np.uint8))
    assert np.array_equal(final_y[:, :29], v["final_y"] * mult)
    assert np.array_equal(flags[:, :29], v["step_flags"])
    assert np.array_equal(curve, np.clip(v["table_index"], 0, 255))
    dec.close()


@pytest.mark.parametrize("name", ["1test", "2test", "3test", "issue6test"])
def test_fixture_pcm_heads(ctx, name):
    from vorbispizza_amd import capi
    from vorbispizza_amd.front import VorbisReader
    v = load("fixture_pcm_heads.npz")
    channels, rate, packets, total, pos, clipped, mid = (int(x) for x in v[name + "_meta"])
    rdr = VorbisReader(ctx, os.path.join(GOLDEN, name + ".ogg"))
    assert (rdr.Channels, rdr.SampleRate) == (channels, rate)
    buf = np.zeros(channels * 4096, dtype=np.float32)
    chunks, thrown = [], 0
    while True:
        try:
            n = rdr.ReadSamples(buf)
        except capi.SynthError as e:  # issue6test's trailing empty packet: that one Read throws (StreamDecoder.cs:777-778)
            assert e.status == capi.E_WINDOW_MISMATCH and name == "issue6test"
            thrown += 1
            continue
        if n == 0:
            break
        chunks.append(buf[: n * channels].reshape(n, channels).copy())
    got = np.concatenate(chunks).T
    assert thrown == (1 if name == "issue6test" else 0)
    assert got.shape == (channels, total) and rdr.SamplePosition == pos and rdr.HasClipped == bool(clipped)
    assert np.abs(got[:, :4096] - v[name + "_pcm"]).max() <= 1e-5
    assert np.abs(got[:, mid: mid + 2048] - v[name + "_mid"]).max() <= 1e-5
    rdr.Dispose()
