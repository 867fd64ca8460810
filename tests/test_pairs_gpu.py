"""The pair route (synth_pairs.hip: the stereo fast path's kernel once more, a workgroup per PAIR of channels) for streams with 4, 6,
8, ... channels whose coupling steps join the channels two by two -- against the routes it replaces, bit for bit: group mode of
synth_kernel (VPZ_NO_PAIRS=1; up to 8 channels) and the separate coupling pass (VPZ_NO_GROUP=1 too), and against the oracle.
Mapping.cs:166-195 (a step touches its two channels and no other), Residue2.cs:42-51, StreamDecoder.cs:515-638, 764-791."""
import numpy as np
import pytest

import helpers
from test_dual_gpu import same_bits
from test_host_paths_gpu import env, run, stream_major_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import Context
    c = Context(0)
    yield c
    c.close()


# (VPZ_PAIRS=1: the pairs wherever they can run -- left to itself the decoder keeps group mode where that is as fast or faster: the
# Residue2 vector in, interleaved PCM out, up to eight channels)
ROUTES = (("pairs", dict(VPZ_PAIRS=1, VPZ_NO_PAIRS=None, VPZ_NO_GROUP=None)), ("group", dict(VPZ_PAIRS=None, VPZ_NO_PAIRS=1, VPZ_NO_GROUP=None)),
          ("separate", dict(VPZ_PAIRS=None, VPZ_NO_PAIRS=1, VPZ_NO_GROUP=1)), ("default", dict(VPZ_PAIRS=None, VPZ_NO_PAIRS=None, VPZ_NO_GROUP=None)))
FLOORS = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)]

# channel count, steps of the short blocks' mapping, steps of the long blocks' mapping
SETUPS = [
    (6, [(0, 1), (2, 3)], [(0, 1), (2, 3)]),                  # BASELINE configs[3]'s shape: adjacent pairs, two channels on their own
    (6, [(0, 2), (3, 4)], [(0, 2), (3, 4)]),                  # 5.1 as libvorbis couples it: L-R and the rears, not adjacent
    (4, [(0, 1), (1, 0), (2, 3)], [(3, 2)]),                  # a pair coupled twice in one mapping, the other way round in the other
    (8, [(0, 1), (2, 3), (4, 5), (6, 7)], [(7, 6), (1, 0)]),
    (6, [], [(4, 1)]),                                        # coupled in one mapping only; 0-2 and 3-5 are pairs of lone channels
    (4, [], []),                                              # no coupling at all
    (10, [(0, 1), (9, 2)], [(2, 9), (5, 4)]),                 # beyond group mode's eight channels
]


@pytest.mark.parametrize("channels,steps0,steps1", SETUPS)
@pytest.mark.parametrize("interleaved", [True, False])
@pytest.mark.parametrize("host", ["serial", "parallel"])
def test_pairs_equal_group_mode_and_the_separate_pass(ctx, oracle, channels, steps0, steps1, interleaved, host):
    """Floor1 + coupling, both input layouts, window switching with streaks of short blocks, silent channels, every output layout,
    two calls per stream (the overlap state of every channel crosses the call), explicit descriptors (serial host pass) and compact
    runs (parallel one)."""
    from vorbispizza_amd import capi
    n_streams, frames = 6, 44
    pk, res, posts, counts = stream_major_batch(n_streams, frames, channels, seed=7300 + channels + 10 * interleaved + len(steps0),
                                                floor=True, interleaved=interleaved, p_ls=0.15, p_sl=0.3, silent_prob=0.12)
    pk["mapping"] = pk["flags"] & 1
    mappings = [{"coupling": steps0, "channel_floor": [0] * channels}, {"coupling": steps1, "channel_floor": [1] * channels}]
    hostkv = dict(VPZ_PAR_MIN_PACKETS=1 << 40) if host == "serial" else dict(VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=4)
    outs = {}
    for layout in (capi.OUT_PLANAR, capi.OUT_INTERLEAVED, capi.OUT_INTERLEAVED_S16, capi.OUT_PLANAR_S16):
        for name, kv in ROUTES:
            with env(**dict(kv, **hostkv)):
                outs[name] = run(ctx, pk, res, posts, counts, n_streams, channels, FLOORS, mappings, layout=layout, splits=2)
        same_bits(outs["pairs"], outs["group"], "pairs vs group, layout %d" % layout)
        same_bits(outs["pairs"], outs["separate"], "pairs vs separate, layout %d" % layout)
        same_bits(outs["pairs"], outs["default"], "pairs vs the decoder's own choice, layout %d" % layout)
        assert np.abs(outs["pairs"][0].astype(np.float64)).max() > 0
    # ... and one stream against the oracle
    s_, per = 3, frames
    opk = []
    for i in range(s_ * per, (s_ + 1) * per):
        half = 1024 if pk["flags"][i] & 1 else 128
        off = int(pk["residue_offset"][i])
        opk.append({"flags": int(pk["flags"][i]), "granule": -1, "mapping": int(pk["mapping"][i]),
                    "residue": res[off: off + channels * half], "posts": posts[i * channels:(i + 1) * channels],
                    "post_count": counts[i * channels:(i + 1) * channels]})
    ref, _, _ = helpers.oracle_decode(oracle, channels, 256, 2048, opk, floors=FLOORS, mappings=mappings)
    with env(**dict(ROUTES[0][1], **hostkv)):
        got = run(ctx, pk, res, posts, counts, n_streams, channels, FLOORS, mappings, layout=capi.OUT_PLANAR, splits=2)
    cap = per * 1024 + 64
    pcm = got[0][s_ * channels * cap:(s_ + 1) * channels * cap].reshape(channels, cap)[:, :ref.shape[1]]
    assert got[1][s_] == ref.shape[1] and ref.shape[1] > 0
    assert np.abs(pcm - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("interleaved", [True, False])
def test_already_floored_packets_of_six_channels(ctx, interleaved):
    """VPZ_PKT_NO_FLOOR packets: no coupling, no curve -- three pairs of independent channels, batches of short blocks."""
    from vorbispizza_amd import capi
    n_streams, frames, channels = 5, 60, 6
    pk, res, _, _ = stream_major_batch(n_streams, frames, channels, seed=7500 + interleaved, floor=False, interleaved=interleaved,
                                       p_ls=0.1, p_sl=0.2)
    for layout in (capi.OUT_PLANAR, capi.OUT_INTERLEAVED):
        outs = {}
        for name, kv in (ROUTES[0], ROUTES[2]):
            with env(**dict(kv, VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=3)):
                outs[name] = run(ctx, pk, res, None, None, n_streams, channels, (), (), layout=layout, splits=3)
        same_bits(outs["pairs"], outs["separate"], "already floored, layout %d" % layout)
        assert np.abs(outs["pairs"][0]).max() > 0


def test_all_long_streams_in_chained_runs_of_pairs(ctx):
    """What BASELINE configs[3] looks like to the host: all-long six-channel streams, enough frames for chained runs (the later run
    of a workgroup takes its predecessor's tail over in LDS), three calls, an end-of-stream trim on the last packet of some."""
    from vorbispizza_amd import capi
    n_streams, frames, channels = 5, 130, 6
    pk, res, posts, counts = stream_major_batch(n_streams, frames, channels, seed=7600, floor=True, interleaved=True,
                                                p_ls=0.0, p_sl=1.0, silent_prob=0.05)
    pk["mapping"] = pk["flags"] & 1
    mappings = [{"coupling": [(0, 1), (2, 3)], "channel_floor": [0] * channels}, {"coupling": [(0, 1), (2, 3)], "channel_floor": [1] * channels}]
    for s in range(0, n_streams, 2):
        last = s * frames + frames - 1
        pk[last]["flags"] |= helpers.PKT_EOS
        pk[last]["granule"] = max(1, (frames - 2) * 1024 - 77)
    for layout in (capi.OUT_PLANAR, capi.OUT_INTERLEAVED):
        with env(VPZ_NO_PAIRS=1, VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=3):
            ref = run(ctx, pk, res, posts, counts, n_streams, channels, FLOORS, mappings, layout=layout, splits=3)
        for run_len in (None, 4, 13):
            with env(VPZ_PAIRS=1, VPZ_NO_PAIRS=None, VPZ_DUAL_RUN=run_len, VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=3):
                got = run(ctx, pk, res, posts, counts, n_streams, channels, FLOORS, mappings, layout=layout, splits=3)
            same_bits(got, ref, "chained runs of %r, layout %d" % (run_len, layout))
        assert np.abs(ref[0]).max() > 0


def test_setups_whose_steps_do_not_pair_the_channels_keep_their_route(ctx):
    """A channel with two partners (in one mapping, or a different one per mapping), an odd channel count: not pairs -- group mode as
    before, same bits as the separate pass."""
    from vorbispizza_amd import capi
    for channels, steps0, steps1 in ((6, [(0, 1), (1, 2)], [(0, 1)]), (4, [(0, 1)], [(0, 2)]), (5, [(0, 1), (2, 3)], [(0, 1), (2, 3)])):
        n_streams, frames = 3, 30
        pk, res, posts, counts = stream_major_batch(n_streams, frames, channels, seed=7700 + channels, floor=True, interleaved=True)
        pk["mapping"] = pk["flags"] & 1
        mappings = [{"coupling": steps0, "channel_floor": [0] * channels}, {"coupling": steps1, "channel_floor": [1] * channels}]
        with env(VPZ_PAIRS=1, VPZ_NO_PAIRS=None, VPZ_NO_GROUP=None):
            a = run(ctx, pk, res, posts, counts, n_streams, channels, FLOORS, mappings, layout=capi.OUT_PLANAR, splits=2)
        with env(VPZ_NO_PAIRS=1, VPZ_NO_GROUP=1):
            b = run(ctx, pk, res, posts, counts, n_streams, channels, FLOORS, mappings, layout=capi.OUT_PLANAR, splits=2)
        same_bits(a, b, "%d channels, %r / %r" % (channels, steps0, steps1))


def test_the_route_that_runs_is_the_route_asked_for(ctx, capfd):
    """VPZ_HOST_PROFILE=1 makes every synth call name its route on stderr: the pairs when forced (VPZ_PAIRS=1) in either input layout; left
    to itself the decoder takes them for planar packets to planar PCM and keeps group mode for the Residue2 vector and for interleaved
    PCM; beyond eight channels (no group mode) the pairs for planar packets, the separate coupling pass for the Residue2 vector;
    VPZ_NO_PAIRS=1: never; a stereo stream: the stereo kernel as before."""
    from vorbispizza_amd import capi
    mappings6 = [{"coupling": [(0, 1), (2, 3)], "channel_floor": [f] * 6} for f in (0, 1)]
    mappings10 = [{"coupling": [(0, 1), (9, 2)], "channel_floor": [f] * 10} for f in (0, 1)]
    mappings2 = [{"coupling": [(0, 1)], "channel_floor": [f] * 2} for f in (0, 1)]
    cases = [  # channels, mappings, interleaved in, layout, env, route
        (6, mappings6, True, capi.OUT_PLANAR, dict(VPZ_PAIRS=1), "pairs"),
        (6, mappings6, True, capi.OUT_INTERLEAVED, dict(VPZ_PAIRS=1), "pairs"),
        (6, mappings6, False, capi.OUT_PLANAR, dict(VPZ_PAIRS=None), "pairs"),
        (6, mappings6, True, capi.OUT_PLANAR, dict(VPZ_PAIRS=None), "group"),
        (6, mappings6, False, capi.OUT_INTERLEAVED, dict(VPZ_PAIRS=None), "group"),
        (6, mappings6, False, capi.OUT_PLANAR, dict(VPZ_PAIRS=None, VPZ_NO_PAIRS=1), "group"),
        (10, mappings10, False, capi.OUT_PLANAR, dict(VPZ_PAIRS=None), "pairs"),
        (10, mappings10, True, capi.OUT_PLANAR, dict(VPZ_PAIRS=None), "separate"),
        (2, mappings2, True, capi.OUT_PLANAR, dict(VPZ_PAIRS=None), "stereo"),
    ]
    for channels, mappings, ilv, layout, kv, route in cases:
        pk, res, posts, counts = stream_major_batch(2, 12, channels, seed=7800 + channels, floor=True, interleaved=ilv)
        pk["mapping"] = pk["flags"] & 1
        capfd.readouterr()
        with env(**dict(dict(VPZ_NO_PAIRS=None, VPZ_NO_GROUP=None, VPZ_NO_DUAL=None), VPZ_HOST_PROFILE=1, **kv)):
            run(ctx, pk, res, posts, counts, 2, channels, FLOORS, mappings, layout=layout)
        err = capfd.readouterr().err
        assert "route %s," % route in err, (channels, ilv, layout, kv, route, err[-300:])
