"""The in-process multi-device dispatcher (include/vorbispizza_multi.h, host/vorbis_multi.cpp): ONE process, one context
group per device, streams partitioned contiguously, no collective (VorbisReader.cs:56-85: one reader,
N independent StreamDecoders).  On a one-GPU box the N-device path is exercised with N groups on device 0: the PCM must be
bit-equal to what one group produces and to the stream-by-stream decode, whatever the partition."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import Context
    c = Context(0)
    yield c
    c.close()


def single_stream_pcm(ctx, raw, s16=False):
    """interleaved PCM of one container through the plain ABI: front end + one decoder + one synth call"""
    from vorbispizza_amd import Decoder, capi
    from vorbispizza_amd.front import OggVorbisFile
    f = OggVorbisFile(raw)
    pk, res, posts, counts = f.decode_packets()
    dec = Decoder(ctx, f.channels, f.block_size0, f.block_size1, floors=f.floors, mappings=f.mappings)
    if f.floor0_data is not None:
        dec.set_floor0_data(*f.floor0_data)
    out = dec.synth(pk, res, posts, counts, out_layout=capi.OUT_INTERLEAVED_S16 if s16 else capi.OUT_INTERLEAVED,
                    on_mismatch="ignore")[0]
    dec.close()
    return out


def library(kinds, n):
    raws = [open(os.path.join(GOLDEN, k), "rb").read() for k in kinds]
    return [raws[i % len(raws)] for i in range(n)]


def run_dispatcher(device_ids, raws, s16=False, capacity_slack=2048, **opt):
    from vorbispizza_amd import multi
    from vorbispizza_amd.front import OggVorbisFile
    infos = {}
    for r in set(raws):
        f = OggVorbisFile(r)
        infos[r] = (f.channels, int(f.total_samples))
    caps = np.array([infos[r][1] + capacity_slack for r in raws], dtype=np.int64)
    sizes = np.array([c * infos[r][0] for c, r in zip(caps, raws)], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    pcm = np.full(int(sizes.sum()), 7, dtype=np.int16) if s16 else np.full(int(sizes.sum()), np.float32(7.0), dtype=np.float32)
    datas = [np.frombuffer(r, dtype=np.uint8) for r in raws]
    d = multi.Dispatcher(device_ids, **opt)
    try:
        results, stats = d.decode_library(datas, pcm, offs, caps, s16=s16)
    finally:
        d.close()
    return pcm, offs, results, stats, infos


@pytest.mark.parametrize("groups", [1, 2, 4, 8])
@pytest.mark.parametrize("s16", [False, True])
def test_partitioned_library_equals_the_stream_by_stream_decode(ctx, groups, s16):
    raws = library(("3test.ogg", "issue6test.ogg", "2test.ogg", "1test.ogg"), 22)
    pcm, offs, results, stats, infos = run_dispatcher([0] * groups, raws, s16=s16, host_threads=6, streams_per_call=4)
    refs = {r: single_stream_pcm(ctx, r, s16=s16) for r in set(raws)}
    for k, r in enumerate(raws):
        ref = refs[r]
        C_ = infos[r][0]
        assert results["status"][k] == 0 and results["channels"][k] == C_
        assert results["samples"][k] == ref.shape[0] <= infos[r][1]  # (issue6test.ogg's last packet is skipped: 63 samples short)
        got = pcm[offs[k]: offs[k] + ref.shape[0] * C_].reshape(-1, C_)
        assert np.array_equal(got.view(np.uint16 if s16 else np.uint32), ref.view(np.uint16 if s16 else np.uint32)), (k, groups)
        # stream k -> group k * groups / n (shard_range)
        lo = [len(raws) * g // groups for g in range(groups + 1)]
        assert lo[results["device_slot"][k]] <= k < lo[results["device_slot"][k] + 1]
    assert sum(stats.device_streams[g] for g in range(groups)) == len(raws)
    assert sum(stats.device_samples[g] for g in range(groups)) == sum(int(results["samples"][k]) * infos[r][0] for k, r in enumerate(raws))
    assert stats.wall_s > 0 and all(stats.device_wall_s[g] > 0 for g in range(groups))
    # issue6test.ogg's trailing packet fails the window check (StreamDecoder.cs:777-778): a per-stream count, not a failure
    name = open(os.path.join(GOLDEN, "issue6test.ogg"), "rb").read()
    assert all(results["skipped_packets"][k] == (1 if r == name else 0) for k, r in enumerate(raws))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("threads_per_group", [2, 3])
def test_eight_groups_with_the_few_thread_defaults(ctx, threads_per_group):
    """The shape of the 8-GPU job on a 16-CPU box, rehearsed on one GPU: EIGHT context groups (all on device 0), two or three host
    threads each and every other option left to its default -- below 8 threads per device that is 2 contexts and 12 slots per
    group.  No deadlock (the timeout), the PCM of the one-group decode bit for bit, every group its contiguous share, and the
    page-locked memory the slots hold is reported and stays bounded (what vpzm_stats.pinned_mib is for)."""
    raws = library(("3test.ogg", "issue6test.ogg"), 72)
    base = run_dispatcher([0], raws, host_threads=4)
    pcm, offs, results, stats, infos = run_dispatcher([0] * 8, raws, host_threads=8 * threads_per_group)
    assert (results["status"] == 0).all()
    assert np.array_equal(base[0].view(np.uint32), pcm.view(np.uint32))
    assert np.array_equal(base[2]["samples"], results["samples"])
    assert stats.threads_per_device == threads_per_group
    assert [int(stats.device_streams[g]) for g in range(8)] == [9] * 8
    assert all(results["device_slot"][k] == k // 9 for k in range(72))
    # 9 streams per group in calls of up to 16: at most 12 slots per group hold arrays, each a sub-batch of these short files
    assert 0 < stats.pinned_mib < 8 * 12 * 64, stats.pinned_mib


def test_partition_does_not_change_a_single_bit(ctx):
    """... and the whole PCM array of the job is the same for 1, 2, 3 and 5 groups, any sub-batch size, any thread count."""
    raws = library(("3test.ogg", "issue6test.ogg"), 17)
    base = run_dispatcher([0], raws, host_threads=4, streams_per_call=16)
    for groups, spc, thr, ctxs in ((2, 3, 2, 1), (3, 16, 8, 2), (5, 1, 3, 3)):
        other = run_dispatcher([0] * groups, raws, host_threads=thr, streams_per_call=spc, contexts_per_device=ctxs)
        assert np.array_equal(base[0].view(np.uint32), other[0].view(np.uint32)), (groups, spc)
        assert np.array_equal(base[2]["samples"], other[2]["samples"])


def test_a_bad_container_or_a_small_area_costs_only_its_stream(ctx):
    from vorbispizza_amd import multi
    good = open(os.path.join(GOLDEN, "3test.ogg"), "rb").read()
    raws = [good, b"not an ogg file at all" * 10, good, good[: len(good) // 3], good]
    from vorbispizza_amd.front import OggVorbisFile
    f = OggVorbisFile(good)
    n_s, C_ = int(f.total_samples), f.channels
    caps = np.array([n_s + 64, n_s + 64, 1000, n_s + 64, n_s + 64], dtype=np.int64)  # stream 2: area too small
    offs = (np.arange(5, dtype=np.int64) * (n_s + 64) * C_)
    pcm = np.zeros(int(5 * (n_s + 64) * C_), dtype=np.float32)
    d = multi.Dispatcher([0, 0], host_threads=3, streams_per_call=2)
    results, _ = d.decode_library([np.frombuffer(r, dtype=np.uint8) for r in raws], pcm, offs, caps)
    d.close()
    assert results["status"][0] == 0 and results["status"][4] == 0
    assert results["status"][1] == multi.E_OPEN
    assert results["status"][2] == multi.E_CAPACITY
    ref = single_stream_pcm(ctx, good)
    for k in (0, 4):
        assert results["samples"][k] == ref.shape[0]
        assert np.array_equal(pcm[offs[k]: offs[k] + ref.size].view(np.uint32), ref.reshape(-1).view(np.uint32))
    # the truncated file is a valid prefix: it decodes to fewer samples (or fails to open) -- never touches its neighbours
    assert results["status"][3] in (0, multi.E_OPEN) and results["samples"][3] < n_s
    assert not pcm[offs[1]: offs[1] + 100].any() and not pcm[offs[2]: offs[2] + 100].any()


def test_streams_with_different_channel_counts_and_floor_types_in_one_library(ctx):
    import synthetic_streams as ss
    raws = []
    for name in ("stereo_floor0", "six_channels_51", "mono_floor1_res1", "three_channels_two_submaps"):
        stream, rng = ss.ALL[name]()
        ogg, _ = stream.build(rng, 30)
        raws.append(bytes(ogg))
    raws = raws + [open(os.path.join(GOLDEN, "3test.ogg"), "rb").read()] + raws
    pcm, offs, results, stats, infos = run_dispatcher([0, 0, 0], raws, host_threads=4, streams_per_call=2)
    for k, r in enumerate(raws):
        ref = single_stream_pcm(ctx, r)
        C_ = infos[r][0]
        assert results["status"][k] == 0 and results["samples"][k] == ref.shape[0], k
        got = pcm[offs[k]: offs[k] + ref.shape[0] * C_].reshape(-1, C_)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), k


def test_a_dispatcher_is_reusable_and_says_what_it_is(ctx):
    from vorbispizza_amd import multi
    raws = library(("3test.ogg",), 5)
    from vorbispizza_amd.front import OggVorbisFile
    f = OggVorbisFile(raws[0])
    n_s, C_ = int(f.total_samples), f.channels
    d = multi.Dispatcher([0, 0], host_threads=2, streams_per_call=2)
    assert multi.lib().vpzm_device_count(d._h) == 2
    outs = []
    for _ in range(3):
        pcm = np.zeros(5 * (n_s + 8) * C_, dtype=np.float32)
        res, _ = d.decode_library([np.frombuffer(r, dtype=np.uint8) for r in raws], pcm, np.arange(5, dtype=np.int64) * (n_s + 8) * C_,
                                  np.full(5, n_s + 8, dtype=np.int64))
        assert (res["status"] == 0).all()
        outs.append(pcm)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)) and np.array_equal(outs[0].view(np.uint32), outs[2].view(np.uint32))
    d.close()
    with pytest.raises(multi.MultiError):
        multi.Dispatcher([99])  # no such device


def _same_setup_streams(lengths, name="stereo_coupled_res2"):
    import synthetic_streams as ss
    raws = []
    for n in lengths:
        stream, rng = ss.ALL[name]()
        ogg, _ = stream.build(rng, n)
        raws.append(bytes(ogg))
    return raws


def test_files_of_one_setup_and_different_lengths_share_a_call(ctx):
    """one encoder setting = one setup header, whatever the song's length: every stream of the call has its own area
    (vpz_decoder_set_stream_capacities), none is held to the smallest one"""
    raws = _same_setup_streams((20, 50, 35, 50, 8, 41))
    for s16 in (False, True):
        pcm, offs, results, stats, infos = run_dispatcher([0], raws, s16=s16, capacity_slack=0, host_threads=2, streams_per_call=8)
        assert (results["status"] == 0).all(), results["status"]
        assert len({infos[r][1] for r in raws}) == 5
        for k, r in enumerate(raws):
            ref = single_stream_pcm(ctx, r, s16=s16)
            assert results["samples"][k] == ref.shape[0] == infos[r][1]
            got = pcm[offs[k]: offs[k] + ref.shape[0] * 2].reshape(-1, 2)
            assert np.array_equal(got.view(np.uint16 if s16 else np.uint32), ref.view(np.uint16 if s16 else np.uint32)), k
        # (no area was written past its end: the next stream's first sample is where it belongs, the last area ends the array)
        assert offs[-1] + results["samples"][-1] * 2 == pcm.size


def test_a_call_is_also_cut_by_the_residue_it_holds(ctx, monkeypatch):
    """sub-batches close at streams_per_call streams OR at a residue budget (64 Mi values; here a few thousand, so that
    long streams ride alone and short ones in twos and threes): the PCM does not know"""
    raws = _same_setup_streams((20, 50, 3, 35, 50, 8, 41, 2, 2, 2)) + library(("3test.ogg", "1test.ogg"), 5)
    want = run_dispatcher([0, 0], raws, capacity_slack=0, host_threads=3, streams_per_call=6)
    monkeypatch.setenv("VPZM_MAX_CALL_VALUES", "30000")
    got = run_dispatcher([0, 0], raws, capacity_slack=0, host_threads=3, streams_per_call=6)
    monkeypatch.delenv("VPZM_MAX_CALL_VALUES")
    assert (want[2]["status"] == 0).all()
    for field in ("status", "samples", "packets", "skipped_packets", "channels"):
        assert np.array_equal(got[2][field], want[2][field]), field
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))


def test_streams_of_every_length_from_one_packet_up(ctx):
    """one packet (no sample yet: the first block only primes the overlap), two, three ... in sub-batches that mix them"""
    lengths = list(range(1, 26)) + [40, 1, 2, 33, 1]
    raws = _same_setup_streams(lengths) + _same_setup_streams(lengths[::3], name="mono_floor1_res1")
    order = np.random.default_rng(11).permutation(len(raws))
    raws = [raws[i] for i in order]
    pcm, offs, results, stats, infos = run_dispatcher([0, 0, 0], raws, capacity_slack=0, host_threads=5, streams_per_call=7)
    assert (results["status"] == 0).all(), results["status"]
    refs = {}
    for k, r in enumerate(raws):
        if r not in refs:
            refs[r] = single_stream_pcm(ctx, r)
        ref, C_ = refs[r], infos[r][0]
        assert results["samples"][k] == ref.shape[0] == infos[r][1], (k, results["samples"][k], ref.shape[0], infos[r][1])
        got = pcm[offs[k]: offs[k] + ref.shape[0] * C_].reshape(-1, C_)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), k
    assert min(infos[r][1] for r in raws) == 0


def test_after_a_failed_call_every_member_gets_a_call_of_its_own(ctx, monkeypatch):
    """"a stream that fails costs only itself" also when it is the synth call of its sub-batch that fails: the members are
    then synthesised one by one.  (A damaged stream that loses its end-of-stream trim does that for real:
    tests/test_hostile_input_gpu.py.)  Here the dispatcher counts EVERY sub-batch's call as failed (VPZM_FAIL_BATCH_CALLS), over
    streams of all kinds: the member-by-member path must give the very PCM, sample counts and skipped-packet counts of the batched one."""
    import synthetic_streams as ss
    raws = _same_setup_streams((20, 50, 35)) + library(("issue6test.ogg", "3test.ogg", "issue6test.ogg"), 3)
    for name in ("stereo_floor0", "six_channels_51", "stereo_floor0"):
        stream, rng = ss.ALL[name]()
        ogg, _ = stream.build(rng, 25)
        raws.append(bytes(ogg))
    want = run_dispatcher([0, 0], raws, capacity_slack=0, host_threads=3, streams_per_call=4)
    monkeypatch.setenv("VPZM_FAIL_BATCH_CALLS", "1")
    got = run_dispatcher([0, 0], raws, capacity_slack=0, host_threads=3, streams_per_call=4)
    monkeypatch.delenv("VPZM_FAIL_BATCH_CALLS")
    assert (want[2]["status"] == 0).all() and int(want[2]["skipped_packets"].sum()) == 2
    for field in ("status", "samples", "packets", "skipped_packets", "channels"):
        assert np.array_equal(got[2][field], want[2][field]), field
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))


def test_callers_on_several_threads_take_a_dispatcher_in_turn(ctx):
    # ctypes drops the GIL for the call: without the dispatcher's own lock two callers would share slots and contexts
    import threading
    from vorbispizza_amd import multi
    from vorbispizza_amd.front import OggVorbisFile
    kinds = (("3test.ogg",), ("2test.ogg", "1test.ogg"), ("3test.ogg", "2test.ogg"), ("1test.ogg",))
    d = multi.Dispatcher([0, 0], host_threads=4, streams_per_call=3)
    jobs = []
    for kind in kinds:
        raws = library(kind, 9)
        info = [(OggVorbisFile(r).channels, int(OggVorbisFile(r).total_samples)) for r in raws]
        caps = np.array([n + 16 for _, n in info], dtype=np.int64)
        sizes = np.array([c * (n + 16) for c, n in info], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        jobs.append(dict(raws=raws, caps=caps, offs=offs, info=info, pcm=np.full(int(sizes.sum()), np.float32(7.0), dtype=np.float32)))

    def call(j):
        try:
            j["res"], _ = d.decode_library([np.frombuffer(r, dtype=np.uint8) for r in j["raws"]], j["pcm"], j["offs"], j["caps"])
        except Exception as e:  # noqa: BLE001 (handed to the asserting thread)
            j["err"] = e

    threads = [threading.Thread(target=call, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    d.close()
    singles = {}
    for j in jobs:
        assert "err" not in j, j.get("err")
        assert (j["res"]["status"] == 0).all()
        for k, r in enumerate(j["raws"]):
            if r not in singles:
                singles[r] = single_stream_pcm(ctx, r)
            ref, C_ = singles[r], j["info"][k][0]
            assert j["res"]["samples"][k] == ref.shape[0]
            got = j["pcm"][j["offs"][k]:j["offs"][k] + ref.shape[0] * C_].reshape(-1, C_)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), k


def test_an_empty_library_and_one_of_nothing_but_garbage(ctx):
    from vorbispizza_amd import multi
    d = multi.Dispatcher([0, 0], host_threads=3)
    res, st = d.decode_library([], np.zeros(4, dtype=np.float32), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
    assert len(res) == 0 and st.device_streams[0] == 0 and st.device_samples[0] == 0
    junk = [np.frombuffer(bytes([i]) * 300 + b"OggS" + bytes(60), dtype=np.uint8) for i in range(7)]
    pcm = np.full(70, np.float32(7.0), dtype=np.float32)
    res, st = d.decode_library(junk, pcm, np.arange(7, dtype=np.int64) * 10, np.full(7, 5, dtype=np.int64))
    assert (res["status"] == multi.E_OPEN).all() and (res["samples"] == 0).all() and (pcm == np.float32(7.0)).all()
    # ... and the dispatcher is as good as new
    raw = library(("1test.ogg",), 1)[0]
    ref = single_stream_pcm(ctx, raw)
    pcm = np.zeros(ref.size + 64, dtype=np.float32)
    res, st = d.decode_library([np.frombuffer(raw, dtype=np.uint8)], pcm, np.zeros(1, dtype=np.int64), np.full(1, ref.shape[0] + 64, dtype=np.int64))
    assert res["status"][0] == 0 and res["samples"][0] == ref.shape[0]
    assert np.array_equal(pcm[: ref.size].view(np.uint32), ref.reshape(-1).view(np.uint32))
    d.close()


def test_more_setups_than_a_context_keeps_decoders_for(ctx):
    """A context keeps the decoders of its last 8 setups (vorbis_multi.cpp: kDecodersPerContext): a library of 12 stereo streams with
    12 different setup headers, decoded twice on ONE context -- decoders are created, evicted and created again, the PCM stays that of
    the stream-by-stream decode."""
    import synthetic_streams as ss
    raws = []
    for seed in range(100, 112):
        stream, rng = ss.stereo_coupled_res2(seed=seed)
        ogg, _ = stream.build(rng, 12)
        raws.append(bytes(ogg))
    assert len(set(raws)) == 12
    raws = raws + raws
    pcm, offs, results, stats, infos = run_dispatcher([0], raws, host_threads=3, streams_per_call=1, contexts_per_device=1)
    for k, r in enumerate(raws):
        ref = single_stream_pcm(ctx, r)
        C_ = infos[r][0]
        assert results["status"][k] == 0 and results["samples"][k] == ref.shape[0], k
        got = pcm[offs[k]: offs[k] + ref.shape[0] * C_].reshape(-1, C_)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), k
