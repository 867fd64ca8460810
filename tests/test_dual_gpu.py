"""The stereo fast path (synth_dual.hip: one wavefront synthesises both channels of a stream -- 16-byte loads of the
Residue2 vector, coupling in registers, two transforms side by side, dense L R L R stores) against the routes it
replaces, bit for bit: group mode of synth_kernel (VPZ_NO_DUAL=1) and the separate coupling pass (VPZ_NO_GROUP=1 too);
and against the oracle.  Residue2.cs:42-51, Mapping.cs:166-195, StreamDecoder.cs:764-791."""
import numpy as np
import pytest

import helpers
from helpers import PKT_EOS
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


ROUTES = (("dual", dict(VPZ_NO_DUAL=None, VPZ_NO_GROUP=None)), ("group", dict(VPZ_NO_DUAL=1, VPZ_NO_GROUP=None)),
          ("separate", dict(VPZ_NO_DUAL=1, VPZ_NO_GROUP=1)))


def same_bits(a, b, what=""):
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3], what
    x, y = (a[0], b[0]) if a[0].dtype == np.int16 else (a[0].view(np.uint32), b[0].view(np.uint32))
    bad = np.nonzero(x != y)[0]
    assert len(bad) == 0, "%s: %d elements differ, first at %d (%r vs %r), last at %d; totals %r" % (
        what, len(bad), bad[0], a[0][bad[0]], b[0][bad[0]], bad[-1], a[1][:6])


@pytest.mark.parametrize("interleaved", [True, False])
@pytest.mark.parametrize("steps", [[(0, 1)], [(1, 0)], [(0, 1), (1, 0), (0, 1)], []])
@pytest.mark.parametrize("host", ["serial", "parallel"])
def test_floored_stereo_dual_equals_group_and_separate(ctx, oracle, interleaved, steps, host):
    """Floor1 + coupling, both input layouts, streams with streaks of short blocks and silent channels, every output
    layout, two calls per stream (the overlap state crosses the call)."""
    from vorbispizza_amd import capi
    n_streams, frames, channels = 20, 70, 2
    pk, res, posts, counts = stream_major_batch(n_streams, frames, channels, seed=9100 + len(steps) + 10 * interleaved,
                                                floor=True, interleaved=interleaved, p_ls=0.2, p_sl=0.25, silent_prob=0.12)
    pk["mapping"] = pk["flags"] & 1
    floors = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)]
    mappings = [{"coupling": steps, "channel_floor": [0, 0]}, {"coupling": steps, "channel_floor": [1, 1]}]
    hostkv = dict(VPZ_PAR_MIN_PACKETS=1 << 40) if host == "serial" else dict(VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=4)
    for layout in (capi.OUT_PLANAR, capi.OUT_INTERLEAVED, capi.OUT_INTERLEAVED_S16, capi.OUT_PLANAR_S16):
        outs = {}
        for name, kv in ROUTES:
            with env(**dict(kv, **hostkv)):
                outs[name] = run(ctx, pk, res, posts, counts, n_streams, channels, floors, mappings, layout=layout, splits=2)
        same_bits(outs["dual"], outs["group"], "dual vs group, layout %d" % layout)
        same_bits(outs["dual"], outs["separate"], "dual vs separate, layout %d" % layout)
        assert np.abs(outs["dual"][0].astype(np.float64)).max() > 0
    # ... and one stream against the oracle
    s_, per = 7, frames
    opk = []
    for i in range(s_ * per, (s_ + 1) * per):
        half = 1024 if pk["flags"][i] & 1 else 128
        off = int(pk["residue_offset"][i])
        opk.append({"flags": int(pk["flags"][i]), "granule": -1, "mapping": int(pk["mapping"][i]),
                    "residue": res[off: off + channels * half], "posts": posts[i * channels:(i + 1) * channels],
                    "post_count": counts[i * channels:(i + 1) * channels]})
    ref, _, _ = helpers.oracle_decode(oracle, channels, 256, 2048, opk, floors=floors, mappings=mappings)
    with env(**dict(ROUTES[0][1], **hostkv)):
        got = run(ctx, pk, res, posts, counts, n_streams, channels, floors, mappings, layout=capi.OUT_PLANAR, splits=2)
    cap = per * 1024 + 64
    pcm = got[0][s_ * channels * cap:(s_ + 1) * channels * cap].reshape(channels, cap)[:, :ref.shape[1]]
    assert got[1][s_] == ref.shape[1] and ref.shape[1] > 0
    assert np.abs(pcm - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("interleaved", [True, False])
@pytest.mark.parametrize("p_sl", [0.1, 0.6])
def test_already_floored_stereo_dual_equals_the_one_channel_kernel(ctx, interleaved, p_sl):
    """VPZ_PKT_NO_FLOOR packets (BASELINE configs[2]'s shape): no coupling, no curve -- but batches of short blocks,
    which only the stereo fast path forms for them."""
    from vorbispizza_amd import capi
    n_streams, frames, channels = 18, 120, 2
    pk, res, _, _ = stream_major_batch(n_streams, frames, channels, seed=5200 + interleaved, floor=False, interleaved=interleaved,
                                       p_ls=0.2, p_sl=p_sl)
    for layout in (capi.OUT_PLANAR, capi.OUT_INTERLEAVED, capi.OUT_INTERLEAVED_S16, capi.OUT_PLANAR_S16):
        outs = {}
        for name, kv in ROUTES[:2]:
            with env(**dict(kv, VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=3)):
                outs[name] = run(ctx, pk, res, None, None, n_streams, channels, layout=layout, splits=3)
        same_bits(outs["dual"], outs["group"], "dual vs one wave per channel, layout %d" % layout)
        assert np.abs(outs["dual"][0].astype(np.float64)).max() > 0


def test_unaligned_output_rows_and_eos_trim(ctx, oracle):
    """Output rows that start off a 16-byte boundary take the element-wise store path; an EOS-trimmed last packet cuts a
    frame where groups of four samples no longer fit.  Dual against group, and the trimmed stream against the oracle."""
    from vorbispizza_amd import Decoder, capi
    channels, n_streams, frames = 2, 6, 40
    pk, res, posts, counts = stream_major_batch(n_streams, frames, channels, seed=31, floor=True, interleaved=True, p_ls=0.3,
                                                p_sl=0.3, silent_prob=0.0)
    pk["mapping"] = pk["flags"] & 1
    floors = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)]
    mappings = [{"coupling": [(0, 1)], "channel_floor": [0, 0]}, {"coupling": [(0, 1)], "channel_floor": [1, 1]}]
    # every stream ends with EOS and a granule position that cuts 1..300 samples off its last packet
    idx = np.arange(len(pk)).reshape(n_streams, frames)
    full = None
    cap = frames * 1024 + 64
    results = {}
    for name, kv in ROUTES[:2]:
        with env(**dict(kv, VPZ_PAR_MIN_PACKETS=1 << 40)):
            dec = Decoder(ctx, channels, 256, 2048, floors=floors, mappings=mappings, n_streams=n_streams)
            out = np.zeros(n_streams * channels * cap + 8, dtype=np.float32)
            offs = np.arange(n_streams, dtype=np.int64) * channels * cap + 1  # rows start 4 bytes off
            if full is None:  # sample counts without a trim
                full = dec.synth_raw(pk, res, posts, counts, out, offs, cap, capi.OUT_INTERLEAVED, 0, capi.MEM_HOST).copy()
                dec.reset(-1)
                for s in range(n_streams):
                    dec.set_position(0, stream=s)
            sub = pk.copy()
            for s in range(n_streams):
                last = idx[s, -1]
                sub["flags"][last] |= PKT_EOS
                sub["granule"][last] = int(full[s]) - (1 + 53 * s)
            out[:] = 0
            w = dec.synth_raw(sub, res, posts, counts, out, offs, cap, capi.OUT_INTERLEAVED, 0, capi.MEM_HOST)
            results[name] = (out.copy(), w.copy())
            dec.close()
    assert np.array_equal(results["dual"][1], results["group"][1])
    last_samples = [128 if not (pk["flags"][idx[s, -1]] & 1) else None for s in range(n_streams)]  # (a short last packet
    cut = [int(full[s] - results["dual"][1][s]) for s in range(n_streams)]                          # holds 128 samples)
    assert all(c == 1 + 53 * s or (m is not None and c == min(1 + 53 * s, m)) for s, (c, m) in enumerate(zip(cut, last_samples)))
    assert sum(c == 1 + 53 * s for s, c in enumerate(cut)) >= 3
    assert np.array_equal(results["dual"][0].view(np.uint32), results["group"][0].view(np.uint32))
    s_ = 3
    opk = []
    for i in idx[s_]:
        half = 1024 if pk["flags"][i] & 1 else 128
        off = int(pk["residue_offset"][i])
        fl = int(pk["flags"][i]) | (PKT_EOS if i == idx[s_, -1] else 0)
        opk.append({"flags": fl, "granule": int(full[s_]) - (1 + 53 * s_) if i == idx[s_, -1] else -1,
                    "mapping": int(pk["mapping"][i]), "residue": res[off: off + channels * half],
                    "posts": posts[i * channels:(i + 1) * channels], "post_count": counts[i * channels:(i + 1) * channels]})
    ref, _, _ = helpers.oracle_decode(oracle, channels, 256, 2048, opk, floors=floors, mappings=mappings, interleave=True)
    n = int(results["dual"][1][s_])
    got = results["dual"][0][1 + s_ * channels * cap: 1 + s_ * channels * cap + n * channels].reshape(n, channels)
    assert ref.shape == got.shape
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_streams_come_and_go_between_calls(ctx, seed):
    """The saved overlap state lives in two copies that a stream flips only when it has frames in a call: streams that
    skip calls, start late, or end early (EOS) while others go on -- random subsets of the streams in every call, random
    amounts of packets -- through the stereo fast path (parallel host pass, compact runs, batches) against the separate
    coupling pass with the serial host pass, bit for bit, sample counts and positions included."""
    from vorbispizza_amd import Decoder, capi
    rng = np.random.default_rng(seed)
    channels, n_streams, frames = 2, 12, 64
    pk, res, posts, counts = stream_major_batch(n_streams, frames, channels, seed=9000 + seed, floor=True, interleaved=True,
                                                p_ls=0.25, p_sl=0.2, silent_prob=0.1)
    pk["mapping"] = pk["flags"] & 1
    floors = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)]
    mappings = [{"coupling": [(0, 1)], "channel_floor": [0, 0]}, {"coupling": [(0, 1)], "channel_floor": [1, 1]}]
    idx = np.arange(len(pk)).reshape(n_streams, frames)
    # the schedule: per call, per stream, how many of its remaining packets it sends (0: it sits the call out)
    cursor = np.zeros(n_streams, dtype=int)
    calls = []
    while (cursor < frames).any() and len(calls) < 40:
        sel = []
        for s in range(n_streams):
            left = frames - cursor[s]
            if left == 0 or rng.random() < 0.35:
                continue
            take = int(min(left, rng.integers(1, 14)))
            sel.append(idx[s, cursor[s]: cursor[s] + take])
            cursor[s] += take
        if sel:
            calls.append(np.concatenate(sel))
    cap = frames * 1024 + 64
    results = {}
    for name, kv in (("dual", dict(VPZ_NO_DUAL=None, VPZ_NO_GROUP=None, VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=4)),
                     ("separate", dict(VPZ_NO_DUAL=1, VPZ_NO_GROUP=1, VPZ_PAR_MIN_PACKETS=1 << 40))):
        with env(**kv):
            dec = Decoder(ctx, channels, 256, 2048, floors=floors, mappings=mappings, n_streams=n_streams)
            out = np.zeros(n_streams * channels * cap, dtype=np.float32)
            offs = np.arange(n_streams, dtype=np.int64) * channels * cap
            total = np.zeros(n_streams, dtype=np.int64)
            per_packet = []
            for sel in calls:
                sub = pk[sel].copy()
                p = np.ascontiguousarray(posts.reshape(len(pk), channels, 64)[sel].reshape(-1, 64))
                c = np.ascontiguousarray(counts.reshape(len(pk), channels)[sel].reshape(-1))
                w = dec.synth_raw(sub, res, p, c, out, offs + total * channels, cap - int(total.max()), capi.OUT_INTERLEAVED, 0,
                                  capi.MEM_HOST)
                per_packet.append(dec.last_packet_samples(len(sub)))
                total += w
            results[name] = (out, total.copy(), np.concatenate(per_packet), [dec.position(s) for s in range(n_streams)])
            dec.close()
    same_bits(results["dual"], results["separate"], "streams coming and going, seed %d" % seed)
    assert len(calls) >= 8 and results["dual"][1].min() > 0


@pytest.mark.parametrize("floor,interleaved_in", [(False, False), (True, True), (True, False)])
def test_chained_runs_give_the_bits_of_recomputed_ones(ctx, floor, interleaved_in):
    """The stereo fast path chains the runs of a workgroup (round 5: the later run overlaps its first frame with the tail its
    neighbour leaves in LDS and emits that frame last, nothing is recomputed -- kPreNeighbour) wherever two runs of one stream meet in
    the steady state, and cuts all-long batches into short runs.  Whatever the run length (VPZ_DUAL_RUN 4 ... 33, the default), all-long
    streams and window-switching ones, streams shorter than a run, three calls with the state carried and an end-of-stream trim on
    the last packet of some streams: the PCM, the counts and the positions are those of the cut that recomputes every predecessor
    (VPZ_NO_CHAIN=1) -- float32 and 16-bit, planar and interleaved."""
    from vorbispizza_amd import capi
    channels = 2
    floors = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)] if floor else ()
    maps = [{"coupling": [(0, 1)], "channel_floor": [0, 0]}, {"coupling": [(1, 0)], "channel_floor": [1, 1]}] if floor else ()
    for n_streams, frames, p_ls in ((3, 260, 0.0), (7, 90, 0.04), (40, 23, 0.0), (2, 515, 0.3)):
        pk, res, posts, counts = stream_major_batch(n_streams, frames, channels, seed=9000 + n_streams + frames, floor=floor,
                                                    interleaved=interleaved_in, p_ls=p_ls, p_sl=0.4, silent_prob=0.1)
        if floor:
            pk["mapping"] = pk["flags"] & 1
        # an end-of-stream trim on every other stream's last packet (StreamDecoder.cs:658-666): a granule 100 samples short
        per = frames
        for s in range(0, n_streams, 2):
            last = s * per + per - 1
            pk[last]["flags"] |= PKT_EOS
            pk[last]["granule"] = max(1, (per - 2) * 1024 - 100)
        for layout in (capi.OUT_PLANAR, capi.OUT_INTERLEAVED, capi.OUT_INTERLEAVED_S16):
            with env(VPZ_NO_CHAIN=1, VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=3):
                ref = run(ctx, pk, res, posts, counts, n_streams, channels, floors, maps, layout=layout, splits=3)
            assert np.abs(ref[0].astype(np.float64)).max() > 0
            for run_len in (None, 4, 5, 8, 13, 33):
                with env(VPZ_NO_CHAIN=None, VPZ_DUAL_RUN=run_len, VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=3):
                    got = run(ctx, pk, res, posts, counts, n_streams, channels, floors, maps, layout=layout, splits=3)
                same_bits(got, ref, "chained runs of %r, %d streams x %d frames, layout %d" % (run_len, n_streams, frames, layout))
