"""GPU parity of vpz_decoder_synth (Mapping.cs:166-195 + StreamDecoder.cs:640-694, 764-791,
515-638) against the CPU oracle, through the C ABI.  Float outputs: <= 1e-5 max-abs per sample
(|PCM| <~ 1); sample counts, positions and clip flags: exact."""
import numpy as np
import pytest

import helpers
from helpers import (PKT_BLOCK_FLAG, PKT_EOS, PKT_INTERLEAVED, PKT_NEXT_FLAG, PKT_NO_FLOOR, PKT_NOT_DECODED,
                     PKT_PREV_FLAG)

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


def build_batch(flags_list, spectra_list, channels, size0=256, size1=2048, extra_flags=0, stream_ids=None,
                granules=None):
    """flags_list[s]: uint8 flags per frame of stream s; spectra_list[s]: [frames, channels, size1/2]."""
    from vorbispizza_amd import make_packets
    n = sum(len(f) for f in flags_list)
    pk = make_packets(n)
    res, opk = [], [[] for _ in flags_list]
    off = 0
    # interleave the streams round-robin to exercise per-stream ordering
    cursors = [0] * len(flags_list)
    i = 0
    while i < n:
        for s, fl in enumerate(flags_list):
            f = cursors[s]
            if f >= len(fl):
                continue
            half = (size1 if fl[f] & 1 else size0) // 2
            r = spectra_list[s][f, :, :half].reshape(-1)
            pk[i]["stream"] = s if stream_ids is None else stream_ids[s]
            pk[i]["flags"] = fl[f] | extra_flags
            pk[i]["granule"] = -1 if granules is None else granules[s][f]
            pk[i]["residue_offset"] = off
            res.append(r)
            opk[s].append({"flags": int(fl[f] | extra_flags), "residue": r, "granule": int(pk[i]["granule"])})
            off += channels * half
            cursors[s] += 1
            i += 1
    return pk, np.concatenate(res), opk


@pytest.mark.parametrize("channels", [1, 2])
@pytest.mark.parametrize("frames", [1, 2, 7, 40, 300])
def test_mixed_window_switching_matches_oracle(ctx, oracle, channels, frames):
    """BASELINE config 3 at oracle-sized length: Markov short/long chain, NO_FLOOR spectra."""
    from vorbispizza_amd import Decoder
    flags = helpers.markov_block_flags(frames, seed=frames)
    spec = helpers.gaussian_spectra((frames, channels, 1024), seed=100 + frames)
    pk, res, opk = build_batch([flags], [spec], channels, extra_flags=PKT_NO_FLOOR)
    dec = Decoder(ctx, channels, 256, 2048)
    got = dec.synth(pk, res)[0]
    ref, pos, _ = helpers.oracle_decode(oracle, channels, 256, 2048, opk[0])
    assert got.shape == ref.shape
    if ref.size:
        assert np.abs(got - ref).max() <= TOL
    assert dec.position(0) == pos
    dec.close()


def test_every_window_geometry_and_sample_counts(ctx, oracle):
    """All five PacketInfo geometries of Mode.cs:30-66 appear; per-packet SampleCount is exact."""
    from vorbispizza_amd import Decoder
    L, S = PKT_BLOCK_FLAG, 0
    P, N = PKT_PREV_FLAG, PKT_NEXT_FLAG
    #        long pl/nl   long pl/ns   short  short  long ps/ns   short  long ps/nl   long pl/nl
    flags = np.array([L | P | N, L | P, S, S, L, S, L | N, L | P | N], dtype=np.uint8)
    spec = helpers.gaussian_spectra((len(flags), 2, 1024), seed=11)
    dec = Decoder(ctx, 2, 256, 2048)
    total = 0
    expect = [0, 1472, 128, 128, 1024, 128, 576, 1024]  # first packet emits nothing (:679)
    ref_all, _, _ = helpers.oracle_decode(
        oracle, 2, 256, 2048, build_batch([flags], [spec], 2, extra_flags=PKT_NO_FLOOR)[2][0])
    for f in range(len(flags)):  # one packet per call: exercises the saved-state path each time
        pk, res, _ = build_batch([flags[f:f + 1]], [spec[f:f + 1]], 2, extra_flags=PKT_NO_FLOOR)
        out = dec.synth(pk, res)[0]
        assert out.shape[1] == expect[f], (f, out.shape)
        assert np.abs(out - ref_all[:, total:total + expect[f]]).max(initial=0) <= TOL
        total += expect[f]
    assert total == ref_all.shape[1]
    dec.close()


@pytest.mark.parametrize("split", [1, 5, 16, 17, 33])
def test_batch_boundaries_do_not_matter(ctx, oracle, split):
    """Feeding the same stream in differently sized batches gives the same PCM (OLA state carry)."""
    from vorbispizza_amd import Decoder
    frames = 70
    flags = helpers.markov_block_flags(frames, seed=5)
    spec = helpers.gaussian_spectra((frames, 2, 1024), seed=6)
    ref, _, _ = helpers.oracle_decode(oracle, 2, 256, 2048,
                                      build_batch([flags], [spec], 2, extra_flags=PKT_NO_FLOOR)[2][0])
    dec = Decoder(ctx, 2, 256, 2048)
    outs = []
    for a in range(0, frames, split):
        pk, res, _ = build_batch([flags[a:a + split]], [spec[a:a + split]], 2, extra_flags=PKT_NO_FLOOR)
        outs.append(dec.synth(pk, res)[0])
    got = np.concatenate(outs, axis=1)
    assert got.shape == ref.shape and np.abs(got - ref).max() <= TOL
    dec.close()


def test_many_streams_interleaved_output_and_reset(ctx, oracle):
    from vorbispizza_amd import Decoder, capi
    n_streams, channels = 5, 2
    flags = [helpers.markov_block_flags(20 + 3 * s, seed=20 + s) for s in range(n_streams)]
    spec = [helpers.gaussian_spectra((len(flags[s]), channels, 1024), seed=40 + s) for s in range(n_streams)]
    pk, res, opk = build_batch(flags, spec, channels, extra_flags=PKT_NO_FLOOR)
    dec = Decoder(ctx, channels, 256, 2048, n_streams=n_streams)
    for rep in range(2):
        outs = dec.synth(pk, res, out_layout=capi.OUT_INTERLEAVED)
        for s in range(n_streams):
            ref, _, _ = helpers.oracle_decode(oracle, channels, 256, 2048, opk[s], interleave=True)
            assert outs[s].shape == ref.shape
            assert np.abs(outs[s] - ref).max() <= TOL
        dec.reset(-1)  # ResetDecoder: the second pass must start from scratch again
    dec.close()


def test_clip_and_has_clipped(ctx, oracle):
    from vorbispizza_amd import Decoder
    frames = 6
    flags = np.full(frames, PKT_BLOCK_FLAG | PKT_PREV_FLAG | PKT_NEXT_FLAG, dtype=np.uint8)
    spec = helpers.gaussian_spectra((frames, 2, 1024), seed=8, sigma=2.0 ** -5)  # |pcm| well above 1
    spec[:, 1] *= 2.0 ** -6  # second channel stays small
    pk, res, opk = build_batch([flags], [spec], 2, extra_flags=PKT_NO_FLOOR)
    dec = Decoder(ctx, 2, 256, 2048, clip_samples=True)
    got = dec.synth(pk, res)[0]
    ref, _, clipped = helpers.oracle_decode(oracle, 2, 256, 2048, opk[0], clip=True)
    assert clipped and dec.has_clipped(0)
    assert np.abs(got).max() == np.float32(0.99999994)
    inside = np.abs(ref) < 0.999
    assert np.array_equal(np.abs(got) == np.float32(0.99999994), np.abs(ref) == np.float32(0.99999994)) or \
        np.abs(got - ref).max() <= 1e-4
    assert np.abs(got[inside] - ref[inside]).max() <= 1e-4  # sigma 2^-5: error scales with amplitude
    dec2 = Decoder(ctx, 2, 256, 2048, clip_samples=True)
    pk2, res2, _ = build_batch([flags], [spec * 2.0 ** -8], 2, extra_flags=PKT_NO_FLOOR)
    dec2.synth(pk2, res2)
    assert not dec2.has_clipped(0)
    dec.close()
    dec2.close()


def test_eos_granule_trim_and_ignored_tail_packets(ctx, oracle):
    """StreamDecoder.cs:657-666: the EOS packet's granule cuts its valid length; packets after EOS
    are never read (:441-447)."""
    from vorbispizza_amd import Decoder
    frames = 8
    flags = np.full(frames, PKT_BLOCK_FLAG | PKT_PREV_FLAG | PKT_NEXT_FLAG, dtype=np.uint8)
    flags[5] |= PKT_EOS
    gran = [[-1, -1, 2048, -1, -1, 4096 + 300, -1, -1]]
    spec = helpers.gaussian_spectra((frames, 2, 1024), seed=12)
    pk, res, opk = build_batch([flags], [spec], 2, extra_flags=PKT_NO_FLOOR, granules=gran)
    dec = Decoder(ctx, 2, 256, 2048)
    got = dec.synth(pk, res)[0]
    ref, pos, _ = helpers.oracle_decode(oracle, 2, 256, 2048, opk[0])
    assert got.shape == ref.shape == (2, 4096 + 300)
    assert np.abs(got - ref).max() <= TOL
    assert dec.position(0) == pos == 4096 + 300
    dec.close()


def test_failed_eos_packet_drains_tail_unwindowed(ctx, oracle):
    """Quirk q4 (StreamDecoder.cs:451-455)."""
    from vorbispizza_amd import Decoder
    flags = np.array([PKT_BLOCK_FLAG | PKT_PREV_FLAG | PKT_NEXT_FLAG] * 3 + [PKT_NOT_DECODED | PKT_EOS],
                     dtype=np.uint8)
    spec = helpers.gaussian_spectra((4, 1, 1024), seed=13)
    pk, res, opk = build_batch([flags], [spec], 1, extra_flags=PKT_NO_FLOOR)
    dec = Decoder(ctx, 1, 256, 2048)
    got = dec.synth(pk, res)[0]
    ref, _, _ = helpers.oracle_decode(oracle, 1, 256, 2048, opk[0])
    assert got.shape == ref.shape == (1, 3072)
    assert np.abs(got - ref).max() <= TOL
    dec.close()


def test_window_mismatch_is_reported_and_costs_only_that_packet(ctx, oracle):
    """Quirk q11: a long tail (1024) followed by a short block throws in OverlapBuffers
    (StreamDecoder.cs:777-778).  That Read fails, the packet is consumed, the state is untouched."""
    from vorbispizza_amd import Decoder, SynthError, capi
    L3 = PKT_BLOCK_FLAG | PKT_PREV_FLAG | PKT_NEXT_FLAG
    flags = np.array([L3, L3, 0, L3, L3], dtype=np.uint8)  # the short block in the middle does not fit
    spec = helpers.gaussian_spectra((5, 1, 1024), seed=14)
    pk, res, opk = build_batch([flags], [spec], 1, extra_flags=PKT_NO_FLOOR)
    dec = Decoder(ctx, 1, 256, 2048)
    cap = 8192
    out = np.zeros(cap, dtype=np.float32)
    with pytest.raises(SynthError) as e:
        dec.synth_raw(pk, res, None, None, out, None, cap, capi.OUT_PLANAR, cap, capi.MEM_HOST)
    assert e.value.status == capi.E_WINDOW_MISMATCH
    ref, pos, _ = helpers.oracle_decode(oracle, 1, 256, 2048, opk[0])
    assert helpers.oracle_decode.last_mismatches == 1 and ref.shape == (1, 3072)
    assert dec.position(0) == pos == 3072
    assert np.abs(out[:3072] - ref[0]).max() <= TOL
    dec.close()


def test_window_mismatch_is_a_per_packet_status_in_a_batch_of_streams(ctx, oracle):
    """ABI v3: in a batch of several streams the packet that fails the window check (StreamDecoder.cs:777-778 throws out
    of THAT Read) is one entry of vpz_decoder_last_packet_status -- the call succeeds, every other packet of every stream
    is synthesised, and the status points at exactly the packets the oracle's state machine skips."""
    from vorbispizza_amd import Decoder, capi
    L3 = PKT_BLOCK_FLAG | PKT_PREV_FLAG | PKT_NEXT_FLAG
    flag_sets = [np.array([L3, L3, L3, L3], dtype=np.uint8),
                 np.array([L3, L3, 0, L3, L3], dtype=np.uint8),        # packet 2 of this stream does not fit
                 np.array([L3, 0, L3, L3, 0, L3], dtype=np.uint8),     # packets 1 and 4 of this one
                 np.array([L3, L3, L3], dtype=np.uint8)]
    specs = [helpers.gaussian_spectra((len(f), 1, 1024), seed=40 + i) for i, f in enumerate(flag_sets)]
    pk, res, opk = build_batch(flag_sets, specs, 1, extra_flags=PKT_NO_FLOOR)
    dec = Decoder(ctx, 1, 256, 2048, n_streams=4)
    cap = 8192
    out = np.zeros(4 * cap, dtype=np.float32)
    offs = np.arange(4, dtype=np.int64) * cap
    w = dec.synth_raw(pk, res, None, None, out, offs, cap, capi.OUT_PLANAR, cap, capi.MEM_HOST, on_mismatch="ignore")
    status = dec.last_packet_status(len(pk))
    bad = [(int(pk["stream"][i]), i) for i in np.nonzero(status)[0]]
    assert dec.last_mismatches() == 3 and all(status[i] == capi.E_WINDOW_MISMATCH for _, i in bad)
    # (the batch interleaves the streams round robin: a packet's ordinal inside its stream)
    per_stream_index = {s: [np.nonzero(pk["stream"] == s)[0].tolist().index(i) for t, i in bad if t == s] for s in range(4)}
    assert per_stream_index == {0: [], 1: [2], 2: [1, 4], 3: []}
    counts = dec.last_packet_samples(len(pk))
    assert all(counts[i] == 0 for _, i in bad)
    for s in range(4):
        ref, pos, _ = helpers.oracle_decode(oracle, 1, 256, 2048, opk[s])
        assert helpers.oracle_decode.last_mismatches == len(per_stream_index[s])
        assert int(w[s]) == ref.shape[1] and dec.position(s) == pos
        assert np.abs(out[s * cap: s * cap + ref.shape[1]] - ref[0]).max() <= TOL
    dec.close()


def make_floor_packets(rng, frames, channels, flags, interleaved, silent_prob=0.1):
    """Residue (zero above a cutoff bin, like `end < N/2`) + raw floor posts per packet."""
    pks = []
    for f in range(frames):
        bf = flags[f] & 1
        half = 1024 if bf else 128
        xl = helpers.LONG_XLIST if bf else helpers.SHORT_XLIST
        res = (rng.standard_normal((channels, half)) * 6).round().astype(np.float32)
        res[:, int(half * 0.85):] = 0
        res[rng.random((channels, half)) < 0.3] = 0
        posts, counts = helpers.random_posts(rng, xl, 2, channels, silent_prob)
        layout = res.T.reshape(-1) if interleaved else res.reshape(-1)
        pks.append({"flags": int(flags[f]) | (PKT_INTERLEAVED if interleaved else 0), "mapping": int(bf),
                    "residue": layout.copy(), "posts": posts, "post_count": counts, "granule": -1})
    return pks


@pytest.mark.parametrize("channels,coupling,interleaved", [
    (1, [], False), (2, [(0, 1)], True), (2, [(0, 1)], False), (6, [(0, 1), (2, 3)], True),
    (3, [(0, 1), (0, 2)], True), (5, [(4, 0), (1, 2), (0, 3)], False), (7, [(0, 6), (5, 1)], True),
    (34, [(0, 33), (32, 1), (2, 0)], True), (33, [(1, 32)], False),
    (255, [(0, 254), (253, 1), (7, 200)], True), (255, [(254, 0)], False),  # (VPZ_MAX_CHANNELS)
    # VPZ_MAX_COUPLING steps (Mapping.cs reads 8 bits + 1), and a long chain that still fits group mode's step table
    (7, [(i % 7, (i * 3 + 1 + i // 7) % 7) for i in range(300) if i % 7 != (i * 3 + 1 + i // 7) % 7][:256], True),
    (6, [(i % 6, (i * 5 + 1 + i // 6) % 6) for i in range(60) if i % 6 != (i * 5 + 1 + i // 6) % 6], True)])
def test_floor1_and_coupling_match_oracle(ctx, oracle, channels, coupling, interleaved):
    """BASELINE config 4 shape at small size: Residue2-interleaved residue, inverse coupling,
    Floor1 render on the GPU, silent channels (ExecuteChannel false)."""
    from vorbispizza_amd import Decoder, make_packets
    frames = 24
    rng = np.random.default_rng(channels * 10 + len(coupling))
    flags = helpers.markov_block_flags(frames, seed=channels)
    floors = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)]
    mappings = [{"coupling": coupling, "channel_floor": [0] * channels},
                {"coupling": coupling, "channel_floor": [1] * channels}]
    opk = make_floor_packets(rng, frames, channels, flags, interleaved)
    pk = make_packets(frames)
    off = 0
    for f, p in enumerate(opk):
        pk[f]["flags"], pk[f]["mapping"], pk[f]["granule"], pk[f]["residue_offset"] = p["flags"], p["mapping"], -1, off
        off += p["residue"].size
    res = np.concatenate([p["residue"] for p in opk])
    posts = np.concatenate([p["posts"] for p in opk]).astype(np.int16)
    counts = np.concatenate([p["post_count"] for p in opk]).astype(np.uint8)
    dec = Decoder(ctx, channels, 256, 2048, floors=floors, mappings=mappings)
    got = dec.synth(pk, res, posts, counts)[0]
    ref, _, _ = helpers.oracle_decode(oracle, channels, 256, 2048, opk, floors=floors, mappings=mappings)
    assert got.shape == ref.shape
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(got - ref).max() <= TOL * scale
    dec.close()


@pytest.mark.parametrize("channels,n_pairs", [(6, 15), (6, 128), (2, 128), (3, 40)])
def test_many_mappings_with_their_own_coupling(ctx, oracle, channels, n_pairs):
    """a packet's mode picks its mapping, a mapping has its own coupling steps: 30 mappings whose steps still fit group
    mode's table, and 256 -- what a setup header can hold (Mapping count: 6 bits + 1 in the reference, 8 in this ABI) --
    which do not (the separate coupling pass, or the stereo path's own table)"""
    from vorbispizza_amd import Decoder, make_packets
    frames = 40
    rng = np.random.default_rng(channels * 1000 + n_pairs)
    flags = helpers.markov_block_flags(frames, seed=channels + n_pairs)
    floors = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)]
    mappings = []
    for k in range(n_pairs):
        steps = []
        for _ in range(int(rng.integers(0, 3)) if channels > 1 else 0):
            m_ = int(rng.integers(0, channels))
            a_ = int((m_ + 1 + rng.integers(0, channels - 1)) % channels)
            steps.append((m_, a_))
        for bf in (0, 1):
            mappings.append({"coupling": steps, "channel_floor": [bf] * channels})
    opk = make_floor_packets(rng, frames, channels, flags, True)
    pk = make_packets(frames)
    off = 0
    for f, p in enumerate(opk):
        p["mapping"] = 2 * int(rng.integers(0, n_pairs)) + (int(flags[f]) & 1)
        pk[f]["flags"], pk[f]["mapping"], pk[f]["granule"], pk[f]["residue_offset"] = p["flags"], p["mapping"], -1, off
        off += p["residue"].size
    res = np.concatenate([p["residue"] for p in opk])
    posts = np.concatenate([p["posts"] for p in opk]).astype(np.int16)
    counts = np.concatenate([p["post_count"] for p in opk]).astype(np.uint8)
    dec = Decoder(ctx, channels, 256, 2048, floors=floors, mappings=mappings)
    got = dec.synth(pk, res, posts, counts)[0]
    ref, _, _ = helpers.oracle_decode(oracle, channels, 256, 2048, opk, floors=floors, mappings=mappings)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= TOL * max(1.0, float(np.abs(ref).max()))
    dec.close()


def test_floor_curve_is_bit_exact(ctx, oracle):
    """The integer part of Floor1 (UnwrapPosts + DDA) must be bit-exact: with a one-hot residue of
    1.0 every IMDCT output is floor[k] * cos(...), so compare via a flat spectrum trick -- feed
    residue = 1 everywhere through NO coupling and check the floored spectrum the oracle would feed
    to its IMDCT produces identical PCM within float rounding of the transform only."""
    from vorbispizza_amd import Decoder, make_packets
    rng = np.random.default_rng(77)
    frames = 12
    flags = np.full(frames, PKT_BLOCK_FLAG | PKT_PREV_FLAG | PKT_NEXT_FLAG, dtype=np.uint8)
    floors = [(helpers.LONG_XLIST, 2)]
    mappings = [{"coupling": [], "channel_floor": [0]}]
    opk = []
    for f in range(frames):
        posts, counts = helpers.random_posts(rng, helpers.LONG_XLIST, 2, 1)
        opk.append({"flags": int(flags[f]), "mapping": 0, "residue": np.ones(1024, dtype=np.float32) * 2.0 ** -10,
                    "posts": posts, "post_count": counts, "granule": -1})
    pk = make_packets(frames)
    for f in range(frames):
        pk[f]["flags"], pk[f]["residue_offset"], pk[f]["granule"] = flags[f], f * 1024, -1
    dec = Decoder(ctx, 1, 256, 2048, floors=floors, mappings=mappings)
    got = dec.synth(pk, np.concatenate([p["residue"] for p in opk]),
                    np.concatenate([p["posts"] for p in opk]), np.concatenate([p["post_count"] for p in opk]))[0]
    ref, _, _ = helpers.oracle_decode(oracle, 1, 256, 2048, opk, floors=floors, mappings=mappings)
    # one wrong floor bin (a y off by one = 11.5 % amplitude step on one bin) would show as >= 1e-4 here
    assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    dec.close()


def random_xlist(rng, half, posts):
    inner = rng.choice(np.arange(1, half), size=posts - 2, replace=False)
    return [0, half] + [int(v) for v in inner]


@pytest.mark.parametrize("size0,size1", [(2048, 2048), (64, 64), (64, 512), (128, 1024), (256, 256), (512, 4096), (1024, 8192),
                                         (2048, 8192), (256, 4096),
                                         # pairs the general variant of the fused kernel takes
                                         (512, 512), (1024, 1024), (512, 1024), (256, 512), (256, 1024), (512, 2048),
                                         (1024, 2048)])
def test_any_block_size_pair_decodes(ctx, oracle, size0, size1):
    """Every Vorbis block-size pair decodes: sizes out of {256, 512, 1024, 2048} through the fused kernel (its
    general variant when 512 / 1024 take part), the others through the three-pass path (floor, exact IMDCT, OLA)
    whose IMDCT is the reference's own schedule -- for N = 64/128 it reproduces the reference's literal output (q1)."""
    from vorbispizza_amd import Decoder, make_packets
    rng = np.random.default_rng(size0 * 7 + size1)
    channels, frames = 2, 30
    bf = (rng.random(frames) < 0.6).astype(np.uint8) if size0 != size1 else np.zeros(frames, dtype=np.uint8)
    prev = np.concatenate([[1], bf[:-1]])
    nxt = np.concatenate([bf[1:], [1]])
    flags = (bf * PKT_BLOCK_FLAG | prev * PKT_PREV_FLAG * bf | nxt * PKT_NEXT_FLAG * bf).astype(np.uint8)
    h0, h1 = size0 // 2, size1 // 2
    floors = [(random_xlist(rng, h0, min(19, h0 // 2)), 2), (random_xlist(rng, h1, min(29, h1 // 2)), 1)]
    mappings = [{"coupling": [(0, 1)], "channel_floor": [0, 0]}, {"coupling": [(1, 0)], "channel_floor": [1, 1]}]
    opk = []
    for f in range(frames):
        half = h1 if bf[f] else h0
        xl, mult = floors[int(bf[f])]
        res = (rng.standard_normal((channels, half)) * 3).round().astype(np.float32)
        res[:, int(half * 0.8):] = 0
        posts, counts = helpers.random_posts(rng, xl, mult, channels, silent_prob=0.15)
        interleaved = bool(f % 2)
        layout = res.T.reshape(-1) if interleaved else res.reshape(-1)
        opk.append({"flags": int(flags[f]) | (PKT_INTERLEAVED if interleaved else 0), "mapping": int(bf[f]),
                    "residue": layout.copy(), "posts": posts, "post_count": counts, "granule": -1})
    pk = make_packets(frames)
    off = 0
    for f, p in enumerate(opk):
        pk[f]["flags"], pk[f]["mapping"], pk[f]["granule"], pk[f]["residue_offset"] = p["flags"], p["mapping"], -1, off
        off += p["residue"].size
    res = np.concatenate([p["residue"] for p in opk])
    posts = np.concatenate([p["posts"] for p in opk]).astype(np.int16)
    counts = np.concatenate([p["post_count"] for p in opk]).astype(np.uint8)
    ref, pos, _ = helpers.oracle_decode(oracle, channels, size0, size1, opk, floors=floors, mappings=mappings)
    # one batch, then the same stream in three batches (state carried in HBM between calls)
    for splits in ([frames], [7, 1, frames - 8]):
        dec = Decoder(ctx, channels, size0, size1, floors=floors, mappings=mappings)
        outs, a = [], 0
        for n in splits:
            sub = pk[a:a + n].copy()
            base = int(sub["residue_offset"][0])
            sub["residue_offset"] -= base
            end = int(pk["residue_offset"][a + n]) if a + n < frames else res.size
            outs.append(dec.synth(sub, res[base:end], posts[a * channels:(a + n) * channels],
                                  counts[a * channels:(a + n) * channels])[0])
            a += n
        got = np.concatenate(outs, axis=1)
        assert got.shape == ref.shape
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= TOL * scale
        assert dec.position(0) == pos
        dec.close()


floor0_safe_amp = helpers.floor0_safe_amp


@pytest.mark.parametrize("size0,size1,order,bark", [(2048, 2048, 8, 256), (256, 2048, 9, 64), (512, 1024, 16, 128)])
def test_floor0_matches_oracle(ctx, oracle, size0, size1, order, bark):
    """Type-0 (LSP) floor, Floor0.cs:164-225: channel 0 uses a floor 0, channel 1 a floor 1, coupled.
    cosf / sqrtf / expf are the device library's, so the bar is relative (1e-5 of the signal)."""
    from vorbispizza_amd import Decoder, make_packets
    rng = np.random.default_rng(order)
    channels, frames = 2, 12
    bf = (rng.random(frames) < 0.6).astype(np.uint8) if size0 != size1 else np.zeros(frames, dtype=np.uint8)
    prev = np.concatenate([[1], bf[:-1]])
    nxt = np.concatenate([bf[1:], [1]])
    flags = (bf * PKT_BLOCK_FLAG | prev * PKT_PREV_FLAG * bf | nxt * PKT_NEXT_FLAG * bf).astype(np.uint8)
    h0, h1 = size0 // 2, size1 // 2
    f0 = {"order": order, "rate": 44100, "bark_map_size": bark, "amp_bits": 6, "amp_ofs": 100}
    floors = [f0, (random_xlist(rng, h0, 11), 2), (random_xlist(rng, h1, 19), 2)]
    mappings = [{"coupling": [(0, 1)], "channel_floor": [0, 1]}, {"coupling": [(0, 1)], "channel_floor": [0, 2]}]
    opk, amps, coeffs = [], [], []
    for f in range(frames):
        half = h1 if bf[f] else h0
        xl, mult = floors[2 if bf[f] else 1]
        res = (rng.standard_normal((channels, half)) * 3).round().astype(np.float32)
        posts, counts = helpers.random_posts(rng, xl, mult, channels)
        # LSP frequencies in (0, pi), increasing; amplitude like Unpack computes it: amp * ampOfs / (2^bits - 1)
        coeff = np.zeros((channels, 32), dtype=np.float32)
        coeff[0, :order] = np.sort(rng.uniform(0.05, 3.0, order))
        # an amplitude that keeps the curve in a sane dB range: amp / sqrt(p+q) - ampOfs <= 0 everywhere
        amp0 = floor0_safe_amp(coeff[0, :order], bark, 100.0) * rng.uniform(0.3, 1.0) if f != 3 else 0.0  # frame 3 silent
        amp = np.array([np.float32(amp0), 0], dtype=np.float32)
        counts[0] = 1 if amp[0] != 0 else 0                            # ExecuteChannel of the floor-0 channel
        opk.append({"flags": int(flags[f]), "mapping": int(bf[f]), "residue": res.reshape(-1).copy(), "posts": posts,
                    "post_count": counts, "granule": -1, "f0_amp": amp, "f0_coeff": coeff})
        amps.append(amp)
        coeffs.append(coeff)
    pk = make_packets(frames)
    off = 0
    for f, p in enumerate(opk):
        pk[f]["flags"], pk[f]["mapping"], pk[f]["granule"], pk[f]["residue_offset"] = p["flags"], p["mapping"], -1, off
        off += p["residue"].size
    res = np.concatenate([p["residue"] for p in opk])
    posts = np.concatenate([p["posts"] for p in opk]).astype(np.int16)
    counts = np.concatenate([p["post_count"] for p in opk]).astype(np.uint8)
    ref, pos, _ = helpers.oracle_decode(oracle, channels, size0, size1, opk, floors=floors, mappings=mappings)
    dec = Decoder(ctx, channels, size0, size1, floors=floors, mappings=mappings)
    dec.set_floor0_data(np.concatenate(amps), np.concatenate(coeffs))
    got = dec.synth(pk, res, posts, counts)[0]
    assert got.shape == ref.shape and np.isfinite(ref).all()
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max()))
    dec.close()


@pytest.mark.parametrize("size0,size1", [(512, 1024), (256, 1024), (1024, 2048), (512, 512)])
@pytest.mark.parametrize("channels", [2, 3])
def test_general_sizes_layouts_streams_and_eos(ctx, oracle, size0, size1, channels):
    """The general variant of the fused kernel (block sizes 512 / 1024 in the mix): several streams with
    different window sequences in one batch, planar vs interleaved output (stereo takes the wave-pair store)
    bit for bit, an EOS trim on the last packet, all against the oracle."""
    from vorbispizza_amd import Decoder, capi, make_packets
    n_streams, frames = 3, 70
    per_stream, pks, spectra = [], [], []
    off = 0
    for s in range(n_streams):
        flags = helpers.markov_block_flags(frames, seed=100 * size0 + 10 * channels + s, start_long=bool(s % 2))
        if size0 == size1:
            flags &= ~np.uint8(PKT_BLOCK_FLAG | PKT_PREV_FLAG | PKT_NEXT_FLAG)
        pk = make_packets(frames)
        opk = []
        for f in range(frames):
            half = (size1 if flags[f] & 1 else size0) // 2
            x = helpers.gaussian_spectra((channels, half), seed=7 * s + f)
            pk[f]["stream"], pk[f]["flags"], pk[f]["granule"], pk[f]["residue_offset"] = s, flags[f] | capi.PKT_NO_FLOOR, -1, off
            off += x.size
            spectra.append(x.reshape(-1))
            opk.append({"flags": int(flags[f]) | capi.PKT_NO_FLOOR, "granule": -1, "residue": x.reshape(-1)})
        # EOS with a granule 37 samples short of the natural end: the last packet is trimmed (:658-666)
        natural, _, _ = helpers.oracle_decode(oracle, channels, size0, size1, opk)
        g = natural.shape[1] - 37
        pk[frames - 1]["flags"] |= capi.PKT_EOS
        pk[frames - 1]["granule"] = g
        opk[-1]["flags"] |= capi.PKT_EOS
        opk[-1]["granule"] = g
        ref, pos, _ = helpers.oracle_decode(oracle, channels, size0, size1, opk)
        assert ref.shape[1] == g
        per_stream.append((ref, pos))
        pks.append(pk)
    # interleave the streams' packets in the batch (packets of one stream stay in order)
    order = np.argsort(np.concatenate([np.arange(frames) * n_streams + s for s in range(n_streams)]), kind="stable")
    pk_all = np.concatenate(pks)[order]
    res = np.concatenate(spectra)
    dec = Decoder(ctx, channels, size0, size1, n_streams=n_streams)
    planar = dec.synth(pk_all, res, out_layout=capi.OUT_PLANAR)
    dec.reset(-1)
    for s in range(n_streams):
        dec.set_position(0, stream=s)
    inter = dec.synth(pk_all, res, out_layout=capi.OUT_INTERLEAVED)
    for s in range(n_streams):
        ref, pos = per_stream[s]
        assert planar[s].shape == ref.shape
        assert np.abs(planar[s] - ref).max() <= TOL
        assert np.array_equal(inter[s], planar[s].T)
        assert dec.position(s) == pos
    dec.close()


@pytest.mark.parametrize("order,bark", [(16, 128), (9, 100), (30, 1024), (16, 256)])
@pytest.mark.parametrize("interleaved", [True, False])
def test_floor0_inside_the_stereo_fast_path_equals_the_separate_floor0_pass(ctx, oracle, order, bark, interleaved):
    """A stereo 256 / 2048 stream with type-0 floors: the stereo fast path applies the floor itself (the record's curve over
    the bark indices from floor0_curve_kernel, looked up per bin -- Floor0.cs:188-222 multiplies a run of bins of one bark
    index by one q), instead of planar temp + floor0_apply_kernel + the one-channel kernel (VPZ_NO_F0_FUSED=1).  Same bits;
    and against the oracle.  Coupled, a silent channel, window switching, two calls (the overlap state crosses them)."""
    from test_host_paths_gpu import env
    from vorbispizza_amd import Decoder, capi, make_packets
    rng = np.random.default_rng(order * 7 + bark)
    frames = 36
    flags = helpers.markov_block_flags(frames, seed=order, p_ls=0.2, p_sl=0.3)
    f0 = {"order": order, "rate": 44100, "bark_map_size": bark, "amp_bits": 6, "amp_ofs": 100}
    floors = [f0, f0]
    mappings = [{"coupling": [(0, 1)], "channel_floor": [0, 0]}, {"coupling": [(0, 1)], "channel_floor": [1, 1]}]
    pk = make_packets(frames)
    res, amps, coeffs, opk = [], [], [], []
    off = 0
    for f in range(frames):
        half = 1024 if flags[f] & 1 else 128
        r = np.round(rng.standard_normal((2, half)) * 3.0).astype(np.float32)
        coeff = np.zeros((2, order), dtype=np.float32)
        amp = np.zeros(2, dtype=np.float32)
        for c in range(2):
            coeff[c] = np.sort(rng.uniform(0.05, 3.0, size=order)).astype(np.float32)
            amp[c] = helpers.floor0_safe_amp(coeff[c], bark, 100.0) * rng.uniform(0.3, 1.0)
        if f == 5:
            amp[1] = 0.0  # a silent channel inside the coupled pair
        pk[f]["flags"] = flags[f] | (capi.PKT_INTERLEAVED if interleaved else 0)
        pk[f]["mapping"] = flags[f] & 1
        pk[f]["granule"] = -1
        pk[f]["residue_offset"] = off
        res.append(r.T.reshape(-1) if interleaved else r.reshape(-1))
        off += 2 * half
        amps.append(amp)
        coeffs.append(coeff)
        opk.append({"flags": int(pk[f]["flags"]), "granule": -1, "mapping": int(pk[f]["mapping"]), "residue": res[-1],
                    "posts": np.zeros((2, 64), dtype=np.int16), "post_count": (amp != 0).astype(np.uint8),
                    "f0_amp": amp, "f0_coeff": coeff})
    res = np.concatenate(res)
    amps, coeffs = np.concatenate(amps), np.concatenate(coeffs)
    posts = np.zeros((frames * 2, 64), dtype=np.int16)
    counts = (amps != 0).astype(np.uint8)
    outs = {}
    for name, kv in (("fused", dict(VPZ_NO_F0_FUSED=None)), ("separate", dict(VPZ_NO_F0_FUSED=1))):
        with env(**kv):
            dec = Decoder(ctx, 2, 256, 2048, floors=floors, mappings=mappings)
            parts = []
            for a, b in ((0, 20), (20, frames)):
                dec.set_floor0_data(amps[2 * a:2 * b], coeffs[2 * a:2 * b])
                sub = pk[a:b].copy()
                parts.append(dec.synth(sub, res, posts[2 * a:2 * b], counts[2 * a:2 * b], out_layout=capi.OUT_PLANAR)[0])
            dec.close()
        outs[name] = np.concatenate(parts, axis=1)
    assert outs["fused"].shape == outs["separate"].shape and outs["fused"].shape[1] > 0
    assert np.array_equal(outs["fused"].view(np.uint32), outs["separate"].view(np.uint32))
    if bark <= 128:  # (a bark map larger than a block's half makes the reference index its w map out of range: no oracle there)
        ref, _, _ = helpers.oracle_decode(oracle, 2, 256, 2048, opk, floors=floors, mappings=mappings)
        assert ref.shape == outs["fused"].shape
        assert np.abs(outs["fused"] - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("size0,size1", [(512, 4096), (1024, 8192), (256, 4096), (2048, 8192), (4096, 4096), (4096, 8192), (8192, 8192),
                                         (2048, 4096), (256, 8192)])
@pytest.mark.parametrize("layout", ["planar", "interleaved", "planar_s16"])
def test_the_fused_kernel_for_4096_and_8192_blocks_equals_the_three_pass_path(ctx, monkeypatch, size0, size1, layout):
    """synth_big_kernel (one pass over HBM: Floor1 render + multiply, the 4096 / 8192-point wave transforms, window + overlap-add,
    store) against the three-pass path it replaces (VPZ_NO_BIG=1: generic_floor_kernel -> gathered IMDCT -> generic_ola_kernel),
    which the oracle tests of this file pin: the same values bit for bit (zeros of a silent channel: +0.0 here), floored and coupled packets, silent channels, planar and
    Residue2-interleaved input, window switching, three batches with the state carried between the calls, three channels."""
    from vorbispizza_amd import Decoder, capi, make_packets
    rng = np.random.default_rng(size0 * 11 + size1)
    channels, frames = 3, 26
    bf = (rng.random(frames) < 0.6).astype(np.uint8) if size0 != size1 else np.zeros(frames, dtype=np.uint8)
    prev = np.concatenate([[1], bf[:-1]])
    nxt = np.concatenate([bf[1:], [1]])
    flags = (bf * PKT_BLOCK_FLAG | prev * PKT_PREV_FLAG * bf | nxt * PKT_NEXT_FLAG * bf).astype(np.uint8)
    h0, h1 = size0 // 2, size1 // 2
    floors = [(random_xlist(rng, h0, min(19, h0 // 2)), 2), (random_xlist(rng, h1, min(61, h1 // 2)), 1)]
    mappings = [{"coupling": [(0, 1)], "channel_floor": [0, 0, 0]}, {"coupling": [(1, 0), (2, 1)], "channel_floor": [1, 1, 1]}]
    pk = make_packets(frames)
    res_parts, posts_parts, count_parts, off = [], [], [], 0
    for f in range(frames):
        half = h1 if bf[f] else h0
        xl, mult = floors[int(bf[f])]
        res = (rng.standard_normal((channels, half)) * 3).round().astype(np.float32)
        res[:, int(half * 0.85):] = 0
        posts, counts = helpers.random_posts(rng, xl, mult, channels, silent_prob=0.15)
        no_floor = f % 5 == 4
        interleaved = bool(f % 2) and not no_floor
        pk[f]["flags"] = int(flags[f]) | (PKT_INTERLEAVED if interleaved else 0) | (capi.PKT_NO_FLOOR if no_floor else 0)
        pk[f]["mapping"], pk[f]["granule"], pk[f]["residue_offset"] = int(bf[f]), -1, off
        if no_floor:
            res = (res * 2.0 ** -6).astype(np.float32)
        res_parts.append(res.T.reshape(-1).copy() if interleaved else res.reshape(-1))
        posts_parts.append(posts)
        count_parts.append(counts)
        off += res.size
    res = np.concatenate(res_parts)
    posts = np.concatenate(posts_parts).astype(np.int16)
    counts = np.concatenate(count_parts).astype(np.uint8)
    out_layout = {"planar": capi.OUT_PLANAR, "interleaved": capi.OUT_INTERLEAVED, "planar_s16": capi.OUT_PLANAR_S16}[layout]

    def decode(no_big, splits):
        if no_big:
            monkeypatch.setenv("VPZ_NO_BIG", "1")
        else:
            monkeypatch.delenv("VPZ_NO_BIG", raising=False)
        dec = Decoder(ctx, channels, size0, size1, floors=floors, mappings=mappings, clip_samples=True)
        outs, a = [], 0
        for n in splits:
            sub = pk[a:a + n].copy()
            base = int(sub["residue_offset"][0])
            end = int(pk[a + n]["residue_offset"]) if a + n < frames else res.size
            sub["residue_offset"] -= base
            out = dec.synth(sub, res[base:end], posts[a * channels:(a + n) * channels], counts[a * channels:(a + n) * channels],
                            out_layout=out_layout)[0]
            outs.append(out)
            a += n
        clipped = dec.has_clipped()
        dec.close()
        axis = 0 if layout == "interleaved" else 1
        return np.concatenate(outs, axis=axis), clipped

    want, want_clip = decode(True, [frames])
    assert want.size > 0
    for splits in ([frames], [5, 1, frames - 6]):
        got, got_clip = decode(False, splits)
        assert got.shape == want.shape, (got.shape, want.shape)
        # value for value; the only bit patterns that may differ are zeros: a silent channel's block is all +0.0 here, as in the
        # reference (Mapping.cs:190-194 clears it) and in the other fused kernels, where the three-pass path transforms zeros and
        # leaves zeros of both signs
        assert np.array_equal(got, want), (size0, size1, layout, splits)
        if layout != "planar_s16":
            differing = got.view(np.uint32) != want.view(np.uint32)
            assert (got[differing] == 0).all() and not np.signbit(got[differing]).any(), (size0, size1, layout, splits)
        assert got_clip == want_clip
