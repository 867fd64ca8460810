#!/usr/bin/env python3
"""bench.py -- decoded PCM Msamples/s of the MI355X Vorbis synthesis back end.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic, HBM-resident input: north_star's workload -- batched
stereo N = 2048 inverse MDCT + window + overlap-add to PCM (`Mdct.Reverse` + `OverlapBuffers` + `StoreContiguous`:
Mdct.cs:15-19, StreamDecoder.cs:764-791, 594-638) through the fused kernel, 65 536 all-long frames x 2 channels per GPU
(BASELINE configs[1]'s batch, taken all the way to PCM), one vpz_decoder_synth call = one kernel launch per step; weak scaling:
every rank owns its own batch; streams are independent, so there is no collective on the data path.  One sample = one
float32 PCM value of one channel; 8 algorithmic bytes per sample (4 read + 4 written).

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     algorithmic bytes of the fused kernel / its HIP-event duration (events on the library's stream) vs 8 TB/s,
  "cpu_baseline": the CPU oracle (restated reference, kind "port") on the same workload, timed on the host cores,
  "extra_workloads": N = 1: configs[1] (Mdct.Reverse alone, the contract line of rounds 1-4, with its own roofline), configs[2]
                  (mixed 256/2048 window switching), configs[3] (6 channels, Residue2-interleaved, coupled, Floor1 on the GPU;
                  fractions from the bytes that have to move), configs[4] one GPU's share, configs[0]; every N: configs[4]'s whole
                  job -- 1024 real stereo streams PARTITIONED over the ranks (vorbispizza_amd/sharding.py), GPU stage and end to
                  end, with a PCM checksum that does not depend on the partition.
The stdout line carries the figures only (it has to fit the driver's 8 KB stdout tail); the same record with every note goes to
gpurun_out/bench_detail.json and to stderr.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

FRAMES = 65536
CHANNELS = 2
N = 2048


def pmc_traffic_bytes(workload="headline"):
    """HBM bytes per launch of a workload's kernels as measured with the PMC counters (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate passes, gfx950 corrections): counters cannot be collected inside this process, so the figure
    comes from profiles/traffic_stamp.json, written on the GPU box by tools/stamp_traffic.py together with the SHA-256
    of the kernels' sources.  Null when there is no stamp or when the sources have changed since it was taken."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "traffic_stamp.json")
    try:
        stamp = json.load(open(path))
        entry = stamp.get("workloads", {}).get(workload) or (stamp if workload == "headline" else None)
        h = hashlib.sha256()
        for rel in entry["kernel_sources"]:
            h.update(open(os.path.join(ROOT, rel), "rb").read())
        if h.hexdigest() != entry["kernel_sources_sha256"]:
            return None
        return int(entry["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError, TypeError):
        return None


HOST_THREADS_CAP = None  # --host-threads N: an explicit cap (none by default: every core the process may run on is used)


def cpu_quota():
    """CPUs' worth of CPU time the container may use (cgroup cpu.max / cfs quota), or None when there is no quota."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, -(-int(q) // int(p)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, -(-q // p))
    except (OSError, ValueError):
        return None


def cores_available():
    """CPUs this process may use: its affinity mask, capped by the container's CPU-time quota (a 256-core host that gives the
    job 16 CPUs' worth shows 256 cores in the mask; more runnable threads than the quota only thrash inside it) -- what the
    box gives the job, not what the machine has.  The library's pools count the same way (vpzh_default_threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    q = cpu_quota()
    return min(n, q) if q else n


def host_threads(whole_node=False):
    """Host threads of ONE rank (CPU baselines, entropy-decode pool): the cores this process may run on, divided by the
    ranks torchrun started on this node (LOCAL_WORLD_SIZE) -- eight ranks must not each claim the whole node, that would
    oversubscribe exactly the end-to-end figure the 8-GPU run reports.  whole_node: the single-process legs, which run
    while the other ranks wait at a barrier, take every core.  No silent cap (rounds 1-3 stopped at 16): --host-threads
    sets one explicitly.  The library's own pools divide the same way (vpzh_default_threads)."""
    n = cores_available()
    local_world = 1 if whole_node else max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    n = max(1, n // local_world)
    return min(n, HOST_THREADS_CAP) if HOST_THREADS_CAP else n


def host_report(threads):
    """What every CPU-side figure of the line says about the host it ran on (north_star: "core count stated")."""
    return {"cores_available": cores_available(), "threads_used": threads, "cpu_quota": cpu_quota(),
            "cores_in_affinity_mask": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "local_world_size": max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)),
            "machine_cores": os.cpu_count()}


def cpu_baseline_imdct(seconds_1thread=4.0, seconds_all=8.0):
    """Times the oracle's Mdct.Reverse restatement on the host cores: 1 thread, then `host_threads()`
    threads.  Every thread re-transforms its own 512-block buffer (6 MiB) until its deadline, so the
    memory footprint is bounded regardless of the core count."""
    import ctypes as C

    import oracle
    L = oracle.lib()
    threads = host_threads()
    rng = np.random.default_rng(2048)
    chunk = 512
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))

    def worker(seconds, result, idx):
        x = (rng.standard_normal((chunk, N // 2)) * 2.0 ** -8).astype(np.float32) if idx == 0 else xs[idx]
        o = np.empty((chunk, N), dtype=np.float32)
        done = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            L.orc_mdct_reverse_batch(N, chunk, fp(x), fp(o))
            done += chunk
        result[idx] = (done, time.perf_counter() - t0)

    xs = [(np.random.default_rng(i).standard_normal((chunk, N // 2)) * 2.0 ** -8).astype(np.float32)
          for i in range(threads)]
    r1 = [None]
    worker(seconds_1thread, r1, 0)
    rate1 = r1[0][0] * (N // 2) / r1[0][1] / 1e6
    res = [None] * threads
    ths = [threading.Thread(target=worker, args=(seconds_all, res, i)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    wall = time.perf_counter() - t0
    blocks = sum(r[0] for r in res)
    rate_all = blocks * (N // 2) / wall / 1e6
    return {
        "value": round(rate_all, 2), "unit": "Msamples/s", "cores": threads, "kind": "port",
        "cores_available": cores_available(), "threads_used": threads, "machine_cores": os.cpu_count(), "cpu_quota": cpu_quota(),
        "value_1thread": round(rate1, 2),
        "sample": "oracle Mdct.Reverse restatement (gcc -O2 -ffp-contract=off, scalar), N=2048: %d channel-blocks on "
                  "1 thread in %.1f s, then %d channel-blocks on %d threads in %.1f s"
                  % (r1[0][0], r1[0][1], blocks, threads, wall),
    }


def cpu_baseline_fused(which, seconds=1.0, reps=3):
    """The restated reference (oracle, C, scalar, Mdct tables warm) on the FUSED workloads, beside the GPU figures of
    configs[2] / [3] / [4] and north_star's line: every thread pushes its own copy of a bounded sample of the workload through
    oracle.FlooredStream (orc_synth_stream_floored: Residue2 de-interleave, inverse coupling, Floor1, IMDCT, window + overlap-add,
    store) until its deadline; 1 thread and every core the rank may use, `reps` repetitions each, the median counts."""
    import helpers
    import oracle
    from vorbispizza_amd import capi, make_packets
    threads = host_threads()

    def sample(i):
        rng = np.random.default_rng(1000 + i)
        if which in ("configs2", "north_star_line"):
            frames = 1024
            flags = helpers.markov_block_flags(frames, seed=3 + i)
            if which == "north_star_line":
                flags = np.full(frames, capi.PKT_BLOCK_FLAG | capi.PKT_PREV_FLAG | capi.PKT_NEXT_FLAG, dtype=np.uint8)
            halves = np.where(flags & 1, 1024, 128).astype(np.int64)
            pk = make_packets(frames)
            pk["flags"] = flags | capi.PKT_NO_FLOOR
            pk["granule"] = -1
            pk["residue_offset"] = np.concatenate([[0], np.cumsum(halves * CHANNELS)[:-1]])
            res = (rng.standard_normal(int((halves * CHANNELS).sum())) * 2.0 ** -8).astype(np.float32)
            return [oracle.FlooredStream(CHANNELS, 256, 2048, pk, res, None, None)]
        if which == "configs3":
            frames, C6 = 192, 6
            pk = make_packets(frames)
            pk["flags"] = capi.PKT_BLOCK_FLAG | capi.PKT_PREV_FLAG | capi.PKT_NEXT_FLAG | capi.PKT_INTERLEAVED
            pk["granule"] = -1
            pk["residue_offset"] = np.arange(frames, dtype=np.int64) * (1024 * C6)
            res = np.round(rng.standard_normal((frames, 1024, C6)) * 4.0).astype(np.float32)
            res[:, FLOOR6_RESIDUE_END:, :] = 0
            posts = np.zeros((frames * C6, 64), dtype=np.int16)
            posts[:, 0] = rng.integers(20, 60, size=frames * C6)
            posts[:, 1] = rng.integers(10, 40, size=frames * C6)
            v = rng.integers(0, 10, size=(frames * C6, 27))
            v[rng.random(v.shape) < 0.35] = 0
            posts[:, 2:29] = v
            counts = np.full(frames * C6, 29, dtype=np.uint8)
            return [oracle.FlooredStream(C6, 256, 2048, pk, res.reshape(-1), posts, counts, floors=[(helpers.LONG_XLIST, 2)],
                                         mappings=[{"coupling": [(0, 1), (2, 3)], "channel_floor": [0] * C6}])]
        from vorbispizza_amd.front import OggVorbisFile
        out = []
        for name, _ in REAL_FIXTURES:  # configs[4]: the two real stereo fixtures, entropy-decoded by the front end
            f = OggVorbisFile(os.path.join(ROOT, "tests", "golden", name))
            pk, r, po, co = f.decode_packets()
            out.append(oracle.FlooredStream(f.channels, f.block_size0, f.block_size1, pk, r, po, co, floors=f.floors, mappings=f.mappings))
        return out

    work = [sample(i) for i in range(threads)]

    def worker(streams, deadline, result, idx):
        done = 0
        while time.perf_counter() < deadline:
            for st in streams:
                done += st.run() * st.channels
        result[idx] = done

    def rate(n_threads):
        res = [0] * n_threads
        t0 = time.perf_counter()
        ths = [threading.Thread(target=worker, args=(work[i], t0 + seconds, res, i)) for i in range(n_threads)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return sum(res) / (time.perf_counter() - t0) / 1e6

    r1 = sorted(rate(1) for _ in range(reps))[reps // 2]
    ra = sorted(rate(threads) for _ in range(reps))[reps // 2]
    return {"Msamples_per_s_1thread": round(r1, 2), "Msamples_per_s_all_threads": round(ra, 2), "kind": "port",
            "cores_available": cores_available(), "threads_used": threads, "repetitions": reps,
            "sample": "oracle restatement (orc_synth_stream_floored, gcc -O2 -ffp-contract=off, scalar, Mdct tables warm), median of "
                      "%d runs of %.1f s; per thread: %s" % (reps, seconds, {
                          "configs2": "1024 frames of the mixed 256/2048 stereo sequence", "north_star_line": "1024 all-long stereo frames",
                          "configs3": "192 frames of the 6-channel Residue2 + coupling + Floor1 workload",
                          "configs4": "3test.ogg + issue6test.ogg, decoded spectra"}[which])}


def build_synth_ola(torch, device, frames=FRAMES, all_long=False, seed=3):
    """BASELINE configs[2]: one stereo stream, 65 536 frames, Markov block flags (seed 3); all_long: every block long
    with long windows on both sides (north_star's "batched stereo N=2048 IMDCT+window+OLA")."""
    import helpers
    from vorbispizza_amd import capi, make_packets
    flags = helpers.markov_block_flags(frames, seed=3)
    if all_long:
        flags = np.full(frames, capi.PKT_BLOCK_FLAG | capi.PKT_PREV_FLAG | capi.PKT_NEXT_FLAG, dtype=np.uint8)
    halves = np.where(flags & 1, 1024, 128).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(halves * CHANNELS)])
    pk = make_packets(frames)
    pk["flags"] = flags | capi.PKT_NO_FLOOR
    pk["granule"] = -1
    pk["residue_offset"] = offs[:-1]
    g = torch.Generator(device=device).manual_seed(seed)
    residue = torch.randn(int(offs[-1]), generator=g, device=device, dtype=torch.float32) * 2.0 ** -8
    # samples per channel: every packet but the first emits RightStart - LeftStart
    samples = 0
    for f in range(1, frames):
        bf, pf, nf = flags[f] & 1, bool(flags[f] & 2), bool(flags[f] & 4)
        if not bf:
            samples += 128
        else:
            samples += (1024 if nf else 1472) - (0 if pf else 448)
    return pk, residue, samples, int(offs[-1])


FLOOR6_RESIDUE_END = 410  # configs[3]: the residue's `_end` in bins per channel (a Residue2 over 6 channels with end = 2460)


def build_floor6(torch, device, frames=16384, declare_support=True):
    """BASELINE configs[3]: 6 channels, Residue2-interleaved residue zero above a cutoff, coupling
    (0,1),(2,3), 29-post Floor1 rendered on the GPU, N = 2048.  declare_support: the mapping carries the residue's support
    (ABI v4, `residue_end`: what the setup header of such a stream says), so the zeros above it are not loaded."""
    import helpers
    from vorbispizza_amd import capi, make_packets
    C6 = 6
    rng = np.random.default_rng(6)
    pk = make_packets(frames)
    pk["flags"] = capi.PKT_BLOCK_FLAG | capi.PKT_PREV_FLAG | capi.PKT_NEXT_FLAG | capi.PKT_INTERLEAVED
    pk["granule"] = -1
    pk["residue_offset"] = np.arange(frames, dtype=np.int64) * (1024 * C6)
    g = torch.Generator(device=device).manual_seed(6)
    res = torch.round(torch.randn((frames, 1024, C6), generator=g, device=device) * 4.0)
    res[:, FLOOR6_RESIDUE_END:, :] = 0  # ~60 % zeros above the cutoff bin (end < N/2)
    posts = np.zeros((frames * C6, 64), dtype=np.int16)
    posts[:, 0] = rng.integers(20, 60, size=frames * C6)
    posts[:, 1] = rng.integers(10, 40, size=frames * C6)
    v = rng.integers(0, 10, size=(frames * C6, 27))
    v[rng.random(v.shape) < 0.35] = 0
    posts[:, 2:29] = v
    counts = np.full(frames * C6, 29, dtype=np.uint8)
    floors = [(helpers.LONG_XLIST, 2)]
    mappings = [{"coupling": [(0, 1), (2, 3)], "channel_floor": [0] * C6}]
    if declare_support:
        mappings[0].update(residue_begin=(0, 0), residue_end=(128, FLOOR6_RESIDUE_END))
    return pk, res.reshape(-1).contiguous(), torch.from_numpy(posts).to(device), torch.from_numpy(counts).to(device), \
        floors, mappings, (frames - 1) * 1024


def build_real_streams(torch, device, name, copies, int16=False):
    """`copies` independent streams of one fixture: CPU entropy decode once (C++ front end), batch
    arrays replicated with their own stream ids (BASELINE configs[4]: per-GPU share of 1024 streams)."""
    from vorbispizza_amd.front import OggVorbisFile
    t0 = time.perf_counter()
    f = OggVorbisFile(os.path.join(ROOT, "tests", "golden", name))
    pk, res, posts, counts = f.decode_packets(int16=int16)  # (int16: the residue as 16-bit integers, ABI v5 -- exact for these files)
    t_front = time.perf_counter() - t0
    n = len(pk)
    pk_all = np.tile(pk, copies)
    pk_all["stream"] = np.repeat(np.arange(copies, dtype=np.int32), n)
    pk_all["residue_offset"] += np.repeat(np.arange(copies, dtype=np.int64) * res.size, n)
    d_res = torch.from_numpy(res).to(device).repeat(copies)
    d_posts = torch.from_numpy(posts).to(device).repeat(copies, 1)
    d_counts = torch.from_numpy(counts).to(device).repeat(copies)
    return f, pk_all, d_res, d_posts, d_counts, t_front


REAL_FIXTURES = (("3test.ogg", 288094), ("issue6test.ogg", 548160))  # (file, decoded samples per channel)
TOTAL_REAL_STREAMS = 1024  # BASELINE configs[4]; global stream s plays fixture s % 2


def time_real_streams(ctx, torch, device, copies, steps=5, warmup=2, plan=None, before_timing=None, reduce_time=None,
                      repeats=3, int16=False):
    """GPU-stage rate over real stereo streams, interleaved output, decoded spectra device-resident.
    `copies` streams of each fixture, or `plan` = [global ids playing fixture 0, global ids playing fixture 1]
    (one decoder group per fixture: streams of a group share a setup header).  Returns (seconds per step, samples,
    host entropy-decode seconds a real host would spend, {global id: (samples, PCM checksum)})."""
    from vorbispizza_amd import Decoder, capi, sharding
    if plan is None:
        plan = [list(range(0, 2 * copies, 2)), list(range(1, 2 * copies, 2))]
    # ONE decoder over all the streams: the fixtures share channel count and block sizes, their floors and mappings are
    # merged (sharding.merge_setups) and a packet's mapping index is shifted by its fixture's base -- one unwrap and one
    # synth launch per step, with runs twice as long, instead of a pair per fixture.  VPZ_BENCH_SPLIT_SETUPS=1: the old way.
    # (measured: 128 streams 0.366 -> 0.320 ms, 256: 0.657 -> 0.635, 512: 1.238 -> 1.213; at 1024 streams either way fills
    # the GPU for milliseconds and the two calls per step overlap their host halves better: 2.35 vs 2.49 ms -- so a batch
    # is merged up to 512 streams)
    split = bool(os.environ.get("VPZ_BENCH_SPLIT_SETUPS")) or \
        (sum(len(ids) for ids in plan) > 512 and not os.environ.get("VPZ_BENCH_MERGE_SETUPS"))
    parts = []
    total_samples = 0
    t_front_total = 0.0
    for (name, samples), ids in zip(REAL_FIXTURES, plan):
        n = len(ids)
        if n == 0:
            continue
        f, pk, res, posts, counts, t_front = build_real_streams(torch, device, name, n, int16=int16)
        parts.append((f, pk, res, posts, counts, samples, ids))
        total_samples += n * samples * f.channels
        t_front_total += t_front * n  # a real host decodes every stream; we decoded one copy
    groups = []
    if split or len(parts) == 1 or len({(p[0].channels, p[0].block_size0, p[0].block_size1) for p in parts}) != 1:
        for f, pk, res, posts, counts, samples, ids in parts:
            n = len(ids)
            dec = Decoder(ctx, f.channels, f.block_size0, f.block_size1, floors=f.floors, mappings=f.mappings, n_streams=n)
            cap = samples + 2048
            out = torch.empty(n * cap * f.channels, device=device, dtype=torch.float32)
            offs = np.arange(n, dtype=np.int64) * cap * f.channels
            groups.append((dec, pk, res, posts, counts, out, offs, cap, [samples] * n, f.channels, ids))
    else:
        f0 = parts[0][0]
        floors, mappings, bases = sharding.merge_setups([(p[0].floors, p[0].mappings) for p in parts])
        n_all = sum(len(p[6]) for p in parts)
        dec = Decoder(ctx, f0.channels, f0.block_size0, f0.block_size1, floors=floors, mappings=mappings, n_streams=n_all)
        cap = max(p[5] for p in parts) + 2048
        pks, s0, r0 = [], 0, 0
        for (f, pk, res, posts, counts, samples, ids), base in zip(parts, bases):
            pk = pk.copy()
            pk["stream"] += s0
            pk["residue_offset"] += r0
            pk["mapping"] += base
            pks.append(pk)
            s0 += len(ids)
            r0 += res.numel()
        out = torch.empty(n_all * cap * f0.channels, device=device, dtype=torch.float32)
        offs = np.arange(n_all, dtype=np.int64) * cap * f0.channels
        groups.append((dec, np.concatenate(pks), torch.cat([p[2] for p in parts]), torch.cat([p[3] for p in parts]),
                       torch.cat([p[4] for p in parts]), out, offs, cap,
                       [p[5] for p in parts for _ in p[6]], f0.channels, [i for p in parts for i in p[6]]))

    def step():
        for dec, pk, res, posts, counts, out, offs, cap, samples, ch, ids in groups:
            dec.reset(-1)
            # (issue6test.ogg's trailing empty packet fails the window check -- StreamDecoder.cs:777-778 throws out of that
            # one Read --: a per-packet status since ABI v3, the call and every stream's PCM are valid)
            # (VPZ_BENCH_REAL_LAYOUT=planar: an A/B switch for tools/, never the bench's default -- the reference's ReadSamples(Span<float>)
            # delivers interleaved samples)
            planar = os.environ.get("VPZ_BENCH_REAL_LAYOUT") == "planar"
            w = dec.synth_raw(pk, res, posts, counts, out, offs, cap, capi.OUT_PLANAR if planar else capi.OUT_INTERLEAVED,
                              cap if planar else 0, capi.MEM_DEVICE, on_mismatch="ignore")
            assert [int(v) for v in w] == samples, "sample counts"
            assert dec.last_mismatches() == sum(1 for v in samples if v == REAL_FIXTURES[1][1]), "skipped packets"

    for _ in range(warmup):
        step()
    ctx.synchronize()
    dt = None
    for _ in range(repeats):  # best of `repeats` loops; with several ranks each loop starts at a barrier and counts
        if before_timing:     # with the slowest rank's time
            before_timing()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.synchronize()
        t = (time.perf_counter() - t0) / steps
        if reduce_time:
            t = reduce_time(t)
        dt = t if dt is None else min(dt, t)
    # per-stream PCM checksum: the sum of the stream's float32 bit patterns as integers (exact, order independent)
    results = {}
    for dec, pk, res, posts, counts, out, offs, cap, samples, ch, ids in groups:
        view = out.view(len(ids), cap * ch).view(torch.int32)
        for row, (sid, smp) in enumerate(zip(ids, samples)):
            results[sid] = (smp, int(view[row, : smp * ch].sum(dtype=torch.int64).item()))
        dec.close()
    return dt, total_samples, t_front_total, results


def end_to_end_real_streams(ctx, torch, copies, threads, sub=16, synth_lanes=2, plan=None, s16=False, streamed=True):
    """configs[4] end to end for one GPU's share: every one of the 2 x `copies` streams is opened and
    entropy-decoded on the host (`threads` host threads of the host library, vpzh_decode_many_progress: one stream at a time
    each -- the reference's model of one decoder per thread), straight into pinned batch buffers; the streams go to the GPU in
    sub-batches of `sub` streams, one host-memory synth call each (H2D + kernels + D2H), issued by the calling thread as
    soon as the sub-batch's streams are reported complete while the pool keeps decoding the following ones (streamed=False:
    one vpzh_decode_many call per sub-batch, i.e. a fork-join of the decode threads in front of every synth call).  `synth_lanes` contexts
    (one HIP stream each, one issuing thread each) take the sub-batches in turn, so the H2D copy of one
    sub-batch overlaps the D2H copy of the previous one (PCIe is full duplex).
    s16: PCM leaves the GPU as the 16-bit samples the reference's tests derive (VPZ_OUT_INTERLEAVED_S16): half the
    bytes over the link on the way back.
    Returns (samples, (wall, wall of the decode stage alone, summed synth-call time))."""
    from vorbispizza_amd import Context
    from concurrent.futures import ThreadPoolExecutor
    from vorbispizza_amd import Decoder, capi
    from vorbispizza_amd.front import OggVorbisFile

    def pinned(n, dtype):
        return torch.empty(n, dtype=dtype, pin_memory=True).numpy()

    per_kind = [copies, copies] if plan is None else [len(ids) for ids in plan]
    lanes = [ctx] + [Context(ctx.device) for _ in range(max(1, synth_lanes) - 1)]
    groups = []
    for (name, samples), copies in zip(REAL_FIXTURES, per_kind):
        if copies == 0:
            continue
        sub_k = max(d for d in range(1, min(sub, copies) + 1) if copies % d == 0)  # streams per synth call
        data = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
        probe = OggVorbisFile(data)
        n, C_, rf = probe.audio_packets, probe.channels, probe.info.residue_floats
        g = {"data": data, "probe": probe, "n": n, "C": C_, "rf": rf, "samples": samples, "cap": samples + 2048,
             "copies": copies, "sub": sub_k, "bytes": np.frombuffer(data, dtype=np.uint8),
             "pk": capi.make_packets(n * copies), "res": pinned(rf * copies, torch.float32),
             "posts": pinned(n * copies * C_ * 64, torch.int16).reshape(n * copies * C_, 64),
             "counts": pinned(n * copies * C_, torch.uint8),
             "out": pinned(copies * (samples + 2048) * C_, torch.int16 if s16 else torch.float32),
             "decs": [Decoder(lanes[b % len(lanes)], C_, probe.block_size0, probe.block_size1, floors=probe.floors,
                              mappings=probe.mappings, n_streams=sub_k) for b in range(copies // sub_k)]}
        groups.append(g)

    from vorbispizza_amd import front

    def decode_sub(g, b):
        # one sub-batch: its streams opened and entropy-decoded by the host library's own threads (vpzh_decode_many: one
        # stream at a time per thread, no Python in the loop), straight into the pinned batch arrays
        n, rf, sub = g["n"], g["rf"], g["sub"]
        lo = b * sub
        front.decode_many([g["bytes"]] * sub, [(lo + j) * n for j in range(sub)], [(lo + j) * rf for j in range(sub)],
                          g["pk"], g["res"], g["posts"], g["counts"], threads=threads, stream_id0=0, residue_origin=lo * rf)

    def synth_sub(g, b):
        n, C_, rf, cap, sub = g["n"], g["C"], g["rf"], g["cap"], g["sub"]
        lo, hi = b * sub, (b + 1) * sub
        dec = g["decs"][b]
        dec.reset(-1)
        offs = np.arange(sub, dtype=np.int64) * cap * C_
        w = dec.synth_raw(g["pk"][lo * n:hi * n], g["res"][lo * rf:hi * rf], g["posts"][lo * n * C_:hi * n * C_],
                          g["counts"][lo * n * C_:hi * n * C_], g["out"][lo * cap * C_:hi * cap * C_], offs, cap,
                          capi.OUT_INTERLEAVED_S16 if s16 else capi.OUT_INTERLEAVED, 0, capi.MEM_HOST, on_mismatch="ignore")
        assert all(int(v) == g["samples"] for v in w), "sample counts"

    def timed_synth(g, b):
        t = time.perf_counter()
        synth_sub(g, b)
        return time.perf_counter() - t

    def decode_group(g, done):
        # all of the group's streams in ONE vpzh_decode_many_progress call (no fork-join per sub-batch: a thread that is done
        # with its stream takes the next one of the whole group); `done` tells the issuing thread which streams are complete
        n, rf, copies = g["n"], g["rf"], g["copies"]
        front.decode_many([g["bytes"]] * copies, [j * n for j in range(copies)], [j * rf for j in range(copies)],
                          g["pk"], g["res"], g["posts"], g["counts"], threads=threads, stream_id0=0, residue_origin=0, done=done)
        return time.perf_counter()

    def rebase_sub(g, b):
        # the synth call of a sub-batch sees its own slices: stream ids and residue offsets from the slices' start
        n, rf, sub = g["n"], g["rf"], g["sub"]
        lo, hi = b * sub, (b + 1) * sub
        g["pk"]["stream"][lo * n:hi * n] -= lo
        g["pk"]["residue_offset"][lo * n:hi * n] -= lo * rf

    best = None
    with ThreadPoolExecutor(max_workers=len(lanes)) as synth_pool, ThreadPoolExecutor(max_workers=1) as decode_pool:
        for _ in range(3):
            t0 = time.perf_counter()
            pending = []
            t_dec_done = t0
            for g in groups:
                if not streamed:
                    for b in range(g["copies"] // g["sub"]):
                        decode_sub(g, b)  # (the GIL is released inside: the synth calls of earlier sub-batches run meanwhile)
                        t_dec_done = time.perf_counter()
                        pending.append(synth_pool.submit(timed_synth, g, b))
                    continue
                done = np.zeros(g["copies"], dtype=np.int32)
                decoding = decode_pool.submit(decode_group, g, done)
                for b in range(g["copies"] // g["sub"]):
                    mine = done[b * g["sub"]:(b + 1) * g["sub"]]
                    while not mine.all():
                        if decoding.done():
                            decoding.result()  # (raises what the decode raised; otherwise the flags are all set by now)
                        time.sleep(2e-5)
                    assert (mine == 1).all()
                    rebase_sub(g, b)
                    pending.append(synth_pool.submit(timed_synth, g, b))
                t_dec_done = decoding.result()
            t_syn = sum(f.result() for f in pending)
            t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t_dec_done - t0, t_syn)
    for g in groups:
        for d in g["decs"]:
            d.close()
    for c in lanes[1:]:
        c.close()
    total = sum(g["copies"] * g["samples"] * g["C"] for g in groups)
    return total, best


def end_to_end_rank_dispatcher(torch, device_index, per_kind, threads, s16=False, repeats=3):
    """A rank's shard of configs[4]'s job -- per_kind[k] streams of fixture k, container bytes in host memory -> interleaved PCM in
    page-locked host memory -- through the product's own host path: ONE vpzm_decode_library call on the rank's device (open, entropy
    decode -- as 16-bit integers where the setup header guarantees them, ABI v5 --, host-memory synth calls on four contexts).
    Returns (samples, (wall, wall until the last stream was entropy-decoded, summed synth-call time))."""
    from vorbispizza_amd import multi
    raws = [np.frombuffer(open(os.path.join(ROOT, "tests", "golden", name), "rb").read(), dtype=np.uint8) for name, _ in REAL_FIXTURES]
    kinds = [k for k, n in enumerate(per_kind) for _ in range(n)]
    if not kinds:
        return 0, (0.0, 0.0, 0.0)
    datas = [raws[k] for k in kinds]
    caps = np.array([REAL_FIXTURES[k][1] + 2048 for k in kinds], dtype=np.int64)
    sizes = caps * 2
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    pcm = torch.empty(int(sizes.sum()), dtype=torch.int16 if s16 else torch.float32, pin_memory=True).numpy()
    d = multi.Dispatcher([device_index], host_threads=threads)
    best = None
    try:
        for _ in range(repeats):
            results, stats = d.decode_library(datas, pcm, offs, caps, s16=s16)
            assert (results["status"] == 0).all(), "dispatcher: a stream failed: %s" % d.last_error()
            assert all(int(results["samples"][i]) == REAL_FIXTURES[k][1] for i, k in enumerate(kinds)), "sample counts"
            if best is None or stats.wall_s < best[0]:
                best = (stats.wall_s, stats.device_decode_s[0], stats.device_synth_s[0])
    finally:
        d.close()
    del pcm
    return int(results["samples"].sum()) * 2, best


def dispatcher_whole_job(torch, device_ids, threads, s16=False, repeats=3, streams=None, checksum=True):
    """configs[4]'s whole job -- 1024 stereo streams, container bytes in host memory -> interleaved PCM in (page-locked) host
    memory -- through the in-process multi-device dispatcher of libvorbispizza_host.so (include/vorbispizza_multi.h): ONE
    process, one context group per entry of device_ids, streams partitioned contiguously (stream s plays fixture s % 2, as in
    the per-process job), no collective.  Returns a dict for the bench line; the checksum is the per-process job's."""
    from vorbispizza_amd import multi, sharding
    streams = TOTAL_REAL_STREAMS if streams is None else streams
    raws = [np.frombuffer(open(os.path.join(ROOT, "tests", "golden", name), "rb").read(), dtype=np.uint8) for name, _ in REAL_FIXTURES]
    caps1 = [smp + 2048 for _, smp in REAL_FIXTURES]
    datas = [raws[i % 2] for i in range(streams)]
    caps = np.array([caps1[i % 2] for i in range(streams)], dtype=np.int64)
    sizes = caps * 2
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    pcm = torch.empty(int(sizes.sum()), dtype=torch.int16 if s16 else torch.float32, pin_memory=True).numpy()
    # (the library's defaults: 16 streams per call, 4 contexts with 8+ threads per device; `threads` None: its own count of decode
    # threads -- the CPUs the process may use plus one per context)
    d = multi.Dispatcher(device_ids, host_threads=threads or 0)
    best = None
    try:
        for _ in range(repeats):
            results, stats = d.decode_library(datas, pcm, offs, caps, s16=s16)
            assert (results["status"] == 0).all(), "dispatcher: a stream failed: %s" % d.last_error()
            assert all(int(results["samples"][i]) == REAL_FIXTURES[i % 2][1] for i in range(streams)), "sample counts"
            if best is None or stats.wall_s < best[0]:
                n = len(device_ids)
                best = (stats.wall_s, [round(stats.device_wall_s[g] * 1e3, 2) for g in range(min(n, 16))],
                        [round(stats.device_decode_s[g] * 1e3, 2) for g in range(min(n, 16))],
                        [round(stats.device_synth_s[g] * 1e3, 2) for g in range(min(n, 16))],
                        [int(stats.device_streams[g]) for g in range(min(n, 16))], stats.threads_per_device, int(stats.pinned_mib))
    finally:
        d.close()
    tot = int(results["samples"].sum()) * 2
    out = {"Msamples_per_s": round(tot / best[0] / 1e6, 1), "wall_ms": round(best[0] * 1e3, 2), "devices": list(device_ids),
           "per_device_ms": {"wall": best[1], "until_last_stream_entropy_decoded": best[2], "summed_synth_calls": best[3]},
           "streams_per_device": best[4], "decode_threads_per_device": best[5], "page_locked_slot_memory_MiB": best[6],
           "host": dict(host_report(threads if threads else host_threads(whole_node=True)), decode_threads_per_device=best[5]),
           "pcm": "int16" if s16 else "float32", "collectives_on_the_data_path": 0, "samples_total": tot}
    if checksum and not s16:
        sums = []
        for i in range(streams):
            smp = int(results["samples"][i])
            sums.append(int(pcm[offs[i]: offs[i] + smp * 2].view(np.int32).sum(dtype=np.int64)))
        out["pcm_checksum"] = "%016x" % sharding.combine_stream_checksums(sums)
    del pcm
    return out


METRIC = "decoded PCM Msamples/s (batched stereo N=2048)"
NS_KERNEL = "synth_dual_kernel<false, false, 0, false>"
NS_WORKLOAD = ("north_star: batched stereo 44.1 kHz N=2048 IMDCT + window + overlap-add to PCM through the fused kernel "
               "(Mdct.Reverse + OverlapBuffers + StoreContiguous, Mdct.cs:15-19 + StreamDecoder.cs:764-791, 594-638): 65536 all-long "
               "frames x 2 ch per GPU (BASELINE configs[1]'s batch taken all the way to PCM), spectra N(0, 2^-8) device-resident, "
               "planar float32 PCM out, 8 B per sample")


class NorthStarLine:
    """The contract workload: one stereo stream of `frames` all-long N = 2048 packets (VPZ_PKT_NO_FLOOR: the spectra are what
    Mdct.Reverse gets), device-resident, through ONE vpz_decoder_synth call per step -- inverse MDCT, window, overlap-add, planar
    store; the stream's state is reset in front of every step (a host-side epoch, no device work)."""

    def __init__(self, torch, ctx, device, seed, frames=FRAMES):
        from vorbispizza_amd import Decoder, capi
        self.capi = capi
        self.pk, self.residue, self.samples, self.res_floats = build_synth_ola(torch, device, frames, all_long=True, seed=seed)
        self.dec = Decoder(ctx, CHANNELS, 256, 2048)
        self.cap = self.samples + 1024
        self.out = torch.empty(CHANNELS * self.cap, device=device, dtype=torch.float32)
        self.alg_bytes = 4 * self.res_floats + 4 * self.samples * CHANNELS  # spectra read once + PCM written once
        self.pcm_values = self.samples * CHANNELS

    def step(self):
        self.dec.reset(-1)
        w = self.dec.synth_raw(self.pk, self.residue, None, None, self.out, None, self.cap, self.capi.OUT_PLANAR, self.cap,
                               self.capi.MEM_DEVICE)
        assert int(w[0]) == self.samples, (int(w[0]), self.samples)

    def close(self):
        self.dec.close()
        del self.residue, self.out


def cpu_baseline_contract():
    """`cpu_baseline` of the contract line: the oracle (restated reference, scalar C) on the SAME workload -- all-long stereo
    N = 2048 frames through orc_synth_stream_floored (IMDCT + window + overlap-add + store) -- 1 thread and every core the rank
    may use; a bounded sample (1024 frames per thread, repeated until the deadline)."""
    r = cpu_baseline_fused("north_star_line", seconds=2.0, reps=3)
    return {"value": r["Msamples_per_s_all_threads"], "unit": "Msamples/s", "cores": r["threads_used"], "kind": "port",
            "value_1thread": r["Msamples_per_s_1thread"], "cores_available": r["cores_available"], "cpu_quota": cpu_quota(),
            "machine_cores": os.cpu_count(),
            "sample": "oracle (orc_synth_stream_floored, gcc -O2 -ffp-contract=off, scalar): 1024 all-long stereo N=2048 frames per "
                      "thread, repeated for 2 s, median of 3; 1 thread, then %d threads" % r["threads_used"]}


def imdct_only_extra(torch, ctx, device, seed, steps, cpu=True):
    """BASELINE configs[1] as it was the contract line in rounds 1-4: Mdct.Reverse alone (no window, no overlap-add: its output
    is not PCM yet), 131 072 channel-blocks, 12 288 B each."""
    from vorbispizza_amd import capi
    count = FRAMES * CHANNELS
    g = torch.Generator(device=device).manual_seed(seed)
    spectra = torch.randn((count, N // 2), generator=g, device=device, dtype=torch.float32) * 2.0 ** -8
    out = torch.empty((count, N), device=device, dtype=torch.float32)
    for _ in range(3):
        ctx.imdct_batch(spectra, N, capi.IMDCT_FAST, out=out)
    ctx.synchronize()
    ctx.timer_start()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.imdct_batch(spectra, N, capi.IMDCT_FAST, out=out)
    kernel_ms = ctx.timer_stop() / steps
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / steps
    alg = count * (4 * (N // 2) + 4 * N)
    ach = alg / (kernel_ms * 1e-3) / 1e9
    e = {"Msamples_per_s": round(count * (N // 2) / dt / 1e6, 1), "ms_per_step": round(dt * 1e3, 4),
         "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                      "traffic": pmc_traffic_bytes("headline"), "kernel": "imdct2048_kernel", "kernel_ms": round(kernel_ms, 4),
                      "algorithmic_bytes_per_launch": alg},
         "note": "vpz_imdct_batch, Mdct.Reverse semantics only; the contract line of rounds 1-4"}
    if cpu:
        c = cpu_baseline_imdct(2.0, 4.0)
        e["cpu_oracle"] = {"Msamples_per_s_1thread": c["value_1thread"], "Msamples_per_s_all_threads": c["value"], "threads_used": c["cores"]}
    del spectra, out
    torch.cuda.empty_cache()
    return e


DROP_FROM_STDOUT = ("note", "sample", "how", "host", "end_to_end_path", "python_pipeline_note", "per_rank_ms", "kind", "repetitions",
                    "cores_in_affinity_mask", "machine_cores", "local_world_size")


def slim(x, top=True):
    """The stdout line keeps the figures; notes and per-rank detail go to the detail file (the driver's record keeps 8 KB of stdout)."""
    if isinstance(x, dict):
        return {k: slim(v, False) for k, v in x.items() if k not in DROP_FROM_STDOUT or top}
    return x


def emit(result, extras):
    """ONE JSON line on stdout (contract fields + roofline + cpu_baseline + the extras' figures, short enough for the driver's
    stdout tail); the same record with every note under gpurun_out/bench_detail.json and on stderr."""
    full = dict(result)
    if extras:
        full["extra_workloads"] = extras
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as f:
            json.dump(full, f, indent=1)
    except OSError:
        pass
    sys.stderr.write("bench detail: " + json.dumps(full) + "\n")
    sys.stderr.flush()
    line = dict(result)
    if extras:
        line["extra_workloads"] = {k: slim(v, False) for k, v in extras.items()}
    print(json.dumps(line, separators=(",", ":")))
    sys.stdout.flush()


def main_single_process(args):
    """bench.py --gpus N --single-process: the alternative launcher -- ONE process, N devices, no torchrun, no
    torch.distributed.  The contract workload runs as N host threads, one vpz_context, one decoder and one batch per device,
    started together and timed from the first launch to the last completion (the in-process form of "barrier, K steps,
    max over ranks"); the 1024-stream job runs through the in-process dispatcher.  Prints one JSON line of the same shape."""
    import torch
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import Context
    n_dev = args.gpus
    have = torch.cuda.device_count()
    rehearsal = os.environ.get("VPZ_BENCH_REHEARSAL") == "1"  # every context on device 0 (a one-GPU box)
    if n_dev > have and not rehearsal:
        raise SystemExit("--gpus %d: only %d device(s) visible (VPZ_BENCH_REHEARSAL=1 puts every context on device 0)" % (n_dev, have))
    ids = [0] * n_dev if rehearsal else list(range(n_dev))
    ctxs, lines = [], []
    frames = args.frames
    for r, dev in enumerate(ids):
        ctxs.append(Context(dev))
        lines.append(NorthStarLine(torch, ctxs[r], torch.device("cuda", dev), 3 + r, frames))
    for dev in set(ids):
        torch.cuda.synchronize(dev)
    start = threading.Barrier(n_dev + 1)
    done = threading.Barrier(n_dev + 1)
    kernel_ms = [0.0] * n_dev

    def lane(r):
        for _ in range(max(0, args.ramp_steps) + args.warmup):  # (the clocks' ramp of a fresh process, then the W warm-up steps: see main())
            lines[r].step()
        ctxs[r].synchronize()
        start.wait()
        ctxs[r].timer_start()
        for _ in range(args.steps):
            lines[r].step()
        kernel_ms[r] = ctxs[r].timer_stop() / args.steps
        ctxs[r].synchronize()
        done.wait()

    ths = [threading.Thread(target=lane, args=(r,)) for r in range(n_dev)]
    for t in ths:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    done.wait()
    elapsed = time.perf_counter() - t0
    for t in ths:
        t.join()
    value = n_dev * lines[0].pcm_values * args.steps / elapsed / 1e6
    alg_bytes = lines[0].alg_bytes
    achieved = alg_bytes / (max(kernel_ms) * 1e-3) / 1e9
    result = {
        "metric": METRIC, "value": round(value, 1), "unit": "Msamples/s",
        "n_gpus": n_dev, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "launcher": "single process: one host thread + one vpz_context per device (bench.py --single-process)",
        "config": {"workload": NS_WORKLOAD, "frames_per_gpu": frames, "channels": CHANNELS, "block_size": N,
                   "sharding": "independent batch per GPU, no collective", "devices": ids,
                   "device_state": "steady: %d untimed steps of the workload in front of the W warm-up steps" % max(0, args.ramp_steps)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": pmc_traffic_bytes("north_star_line") if frames == FRAMES else None, "kernel": NS_KERNEL,
                     "kernel_ms": round(max(kernel_ms), 4), "kernel_ms_per_device": [round(k, 4) for k in kernel_ms],
                     "algorithmic_bytes_per_launch": alg_bytes},
    }
    for ln in lines:
        ln.close()
    for c in ctxs:
        c.close()
    torch.cuda.empty_cache()
    extras = {}
    if not args.no_extras:
        thr = host_threads(whole_node=True) if HOST_THREADS_CAP else None  # (None: the dispatcher's own count)
        extras = {"configs[4] job, ONE process (vpzm_decode_library), %d device(s), f32 PCM" % n_dev: dispatcher_whole_job(torch, ids, thr),
                  "configs[4] job, ONE process, s16 PCM": dispatcher_whole_job(torch, ids, thr, s16=True)}
    emit(result, extras)


def cpu_plumbing_2test():
    """BASELINE configs[0]: TestFiles/2test.ogg decoded on the CPU only (front end + oracle)."""
    import helpers
    import oracle
    from vorbispizza_amd.front import OggVorbisFile
    t0 = time.perf_counter()
    f = OggVorbisFile(os.path.join(ROOT, "tests", "golden", "2test.ogg"))
    pk, res, posts, counts = f.decode_packets()
    t1 = time.perf_counter()
    pcm, pos, _ = helpers.oracle_decode(oracle, f.channels, f.block_size0, f.block_size1,
                                        helpers.packets_for_oracle(f, pk, res, posts, counts),
                                        floors=f.floors, mappings=f.mappings)
    t2 = time.perf_counter()
    return {"file": "tests/golden/2test.ogg (mono 44.1 kHz, 310 packets)", "samples": int(pcm.shape[1]),
            "expected_samples": 315790, "front_end_ms": round((t1 - t0) * 1e3, 2),
            "oracle_synthesis_ms": round((t2 - t1) * 1e3, 2),
            "note": "CPU only: C++ front end + C oracle driven packet by packet from Python"}


def time_decoder(ctx, dec, torch, pk, residue, posts, counts, samples, channels, steps, warmup, repeats=3, layout=None):
    """Seconds per vpz_decoder_synth call: `repeats` timed loops of `steps` calls each, the best loop counts (the
    side workloads share the box's host cores with other tenants; one disturbed loop must not decide the figure).
    The contract line in main() is NOT measured this way: it times exactly --steps steps once."""
    from vorbispizza_amd import capi
    out = torch.empty(channels * (samples + 1024), device=residue.device, dtype=torch.float32)
    cap = samples + 1024
    layout = capi.OUT_PLANAR if layout is None else layout

    def step():
        dec.reset(-1)
        w = dec.synth_raw(pk, residue, posts, counts, out, None, cap, layout, cap, capi.MEM_DEVICE)
        assert int(w[0]) == samples, (int(w[0]), samples)

    for _ in range(warmup):
        step()
    ctx.synchronize()
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        best = dt if best is None else min(best, dt)
    # The device time of every call of a back-to-back sequence: one HIP event on the LIBRARY's stream (wrapped as a torch external
    # stream -- an event recorded on torch's own stream would not see the kernels) behind every call, nothing synchronised in
    # between; consecutive events are one call's kernels plus the gap to the next call's.  What the spread of a fixed batch is:
    # the run cutting lands on one cut for it, call after call (VPZ_HOST_PROFILE=1 logs it), the rest is the device's.
    ext = torch.cuda.ExternalStream(int(ctx.stream), device=residue.device)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 2)]
    step()
    evs[0].record(ext)
    for i in range(steps + 1):
        step()
        evs[i + 1].record(ext)
    ctx.synchronize()
    us = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(1, steps + 1)]  # (the first interval holds the queue's start-up)
    time_decoder.last_spread = {"device_us_per_call_min": round(min(us), 1), "device_us_per_call_mean": round(sum(us) / len(us), 1),
                                "device_us_per_call_max": round(max(us), 1), "max_over_min": round(max(us) / min(us), 3),
                                "calls": len(us), "how": "HIP events on the library's stream behind each of consecutive calls"}
    return best, out


def fused_entry(samples_total, dt, byt, traffic=None, cpu=None, note=None, **more):
    e = {"Msamples_per_s": round(samples_total / dt / 1e6, 1), "ms_per_step": round(dt * 1e3, 3),
         "GBps": round(byt / dt / 1e9, 1), "frac_of_8TBps": round(byt / dt / 1e9 / HBM_PEAK_GBS, 4), "bytes": byt}
    if traffic is not None:
        e["traffic"] = traffic
    e.update(more)
    if getattr(time_decoder, "last_spread", None):
        s = time_decoder.last_spread
        e["device_us_per_call"] = [s["device_us_per_call_min"], s["device_us_per_call_mean"], s["device_us_per_call_max"]]
        time_decoder.last_spread = None
    if cpu:
        e["cpu_oracle"] = {"Msamples_per_s_1thread": cpu["Msamples_per_s_1thread"], "Msamples_per_s_all_threads": cpu["Msamples_per_s_all_threads"],
                           "threads_used": cpu["threads_used"], "sample": cpu["sample"]}
    if note:
        e["note"] = note
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[1]/[2]/[3]/[4] side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames", type=int, default=FRAMES, help="frames of the contract workload (default: BASELINE's 65536)")
    ap.add_argument("--ramp-steps", type=int, default=200, help="untimed steps of the contract workload between its cold measurement and "
                    "the one reported (the device's clocks ramp up over the first few milliseconds of load; 0: none)")
    ap.add_argument("--extras-frames", type=int, default=FRAMES, help="frames of the configs[2] side measurement")
    ap.add_argument("--extras-frames6", type=int, default=16384, help="frames of the configs[3] side measurement")
    ap.add_argument("--host-threads", type=int, default=0, help="explicit cap on the host threads of a rank (default: every "
                    "core the process may run on, divided by LOCAL_WORLD_SIZE)")
    ap.add_argument("--single-process", action="store_true", help="alternative launcher: ONE process drives --gpus N devices "
                    "(one host thread + one context per device, the in-process dispatcher for the 1024-stream job) instead of "
                    "one process per GPU under torchrun")
    args = ap.parse_args()
    global HOST_THREADS_CAP
    HOST_THREADS_CAP = args.host_threads if args.host_threads > 0 else None
    if args.single_process:
        return main_single_process(args)

    import torch
    import __graft_entry__ as ge

    from vorbispizza_amd import sharding
    world, rank, local_rank = sharding.env_world()
    distributed = world > 1
    # Rehearsal of the N > 1 path on a one-GPU box (never what the driver runs): VPZ_BENCH_REHEARSAL=1 puts every rank
    # on device 0 and takes gloo for the barrier / reductions (RCCL refuses two ranks on one GPU).
    rehearsal = distributed and os.environ.get("VPZ_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if distributed:
        torch.cuda.set_device(local_rank)
        sharding.init("gloo" if rehearsal else "nccl")  # RCCL; only the barrier, the max-over-ranks of the timing
                                                        # and the merge of per-stream counts / checksums use it
    if rank == 0:
        ge.build()
    if distributed:
        sharding.barrier()
    from vorbispizza_amd import Context, Decoder, capi, make_packets

    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    ctx = Context(local_rank)
    red_device = "cpu" if rehearsal else device  # where the tiny reduction tensors live

    # ---------------- the contract line: north_star's workload -- batched stereo N = 2048 IMDCT + window + overlap-add to PCM,
    # device-resident, one fused kernel launch per step (weak scaling: every rank owns its own batch, no collective)
    line = NorthStarLine(torch, ctx, device, 3 + rank, args.frames)
    torch.cuda.synchronize()

    def measure():
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; (wall seconds, kernel ms per step)"""
        for _ in range(args.warmup):
            line.step()
        ctx.synchronize()
        if distributed:
            sharding.barrier()
        torch.cuda.synchronize()
        ctx.timer_start()  # HIP events on the stream the kernel is launched on
        t0 = time.perf_counter()
        for _ in range(args.steps):
            line.step()
        k_ms = ctx.timer_stop() / args.steps  # also synchronises the stream; the K launches back to back incl. their gaps
        ctx.synchronize()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if distributed:
            el = sharding.max_over_ranks(el, red_device)
            k_ms = sharding.max_over_ranks(k_ms, red_device)
            sharding.barrier()
        return el, k_ms

    # A process that has just started finds the device at idle clocks: the first ~25 calls (6 ms) of ANY cut of this workload run
    # 8-12 % slower than the calls after them (profiles/r5_bench_cold_and_steady.txt: 0.254-0.268 ms per step over steps 6-25,
    # 0.229-0.232 over 200).  The contract line is the device's steady state -- what a decode service runs in --: the same W + K
    # measurement is taken twice, once cold (reported as `cold_start`) and once behind `ramp_steps` untimed steps of the workload.
    cold_elapsed, cold_kernel_ms = measure()
    ramp_steps = max(0, args.ramp_steps)
    for _ in range(ramp_steps):
        line.step()
    ctx.synchronize()
    elapsed, kernel_ms = measure()

    value = world * line.pcm_values * args.steps / elapsed / 1e6
    alg_bytes = line.alg_bytes  # 4 B x spectra values read + 4 B x PCM values written = 8 B per sample
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    result = {
        "metric": METRIC,
        "value": round(value, 1),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": NS_WORKLOAD, "frames_per_gpu": args.frames, "channels": CHANNELS, "block_size": N,
                   "pcm_values_per_step_per_gpu": line.pcm_values, "sharding": "independent batch per GPU, no collective",
                   "device_state": "steady: %d untimed steps of the workload between the cold measurement and this one" % ramp_steps},
        "cold_start": {"ms_per_step": round(cold_elapsed / args.steps * 1e3, 4), "kernel_ms": round(cold_kernel_ms, 4),
                       "value": round(world * line.pcm_values * args.steps / cold_elapsed / 1e6, 1),
                       "frac": round(line.alg_bytes / (cold_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "note": "the same W warm-up + K timed steps taken first, on the device as the fresh process found it (idle clocks)"},
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic_bytes("north_star_line") if args.frames == FRAMES else None,
            "kernel": NS_KERNEL, "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
            "bytes_per_sample": round(alg_bytes / line.pcm_values, 3),
        },
    }
    line.close()
    del line
    torch.cuda.empty_cache()

    extras = {}
    cpu = not args.no_cpu_baseline
    if rank == 0 and world == 1:
        if cpu:
            result["cpu_baseline"] = cpu_baseline_contract()
        if not args.no_extras:
            # configs[1]: Mdct.Reverse alone (the contract line of rounds 1-4), with its own roofline
            extras["configs[1] IMDCT only, N=2048, 131072 ch-blocks"] = imdct_only_extra(torch, ctx, device, 2048 + rank, 20, cpu)
            # configs[2]
            pk, residue, samples, res_floats = build_synth_ola(torch, device, args.extras_frames)
            dec = Decoder(ctx, CHANNELS, 256, 2048)
            dt, _ = time_decoder(ctx, dec, torch, pk, residue, None, None, samples, CHANNELS, 40, 3)
            byt = 4 * res_floats + 4 * samples * CHANNELS
            extras["configs[2] mixed 256/2048 + window + OLA, stereo, %d frames" % args.extras_frames] = fused_entry(
                samples * CHANNELS, dt, byt, traffic=pmc_traffic_bytes("configs2") if args.extras_frames == FRAMES else None,
                cpu=cpu_baseline_fused("configs2") if cpu else None,
                note="whole vpz_decoder_synth call incl. the host state machine; best of 3 loops of 40 calls; planar out")
            dec.close()
            del residue
            torch.cuda.empty_cache()
            # configs[3]: `bytes` = what has to move -- with the support declared (ABI v4) the upper half of every vector is
            # neither loaded nor staged; the full-vector figure of rounds 1-3 is the secondary one
            pk, res6, posts, counts, floors, mappings, samples6 = build_floor6(torch, device, args.extras_frames6)
            dec = Decoder(ctx, 6, 256, 2048, floors=floors, mappings=mappings)
            dt, _ = time_decoder(ctx, dec, torch, pk, res6, posts, counts, samples6, 6, 40, 3)
            byt_full = 4 * res6.numel() + 4 * samples6 * 6 + posts.numel() * 2
            byt = byt_full - 4 * res6.numel() // 2
            extras["configs[3] 6ch Residue2 + coupling + Floor1, N=2048, %d frames, planar out" % args.extras_frames6] = fused_entry(
                samples6 * 6, dt, byt, cpu=cpu_baseline_fused("configs3") if cpu else None,
                frac_full_vector_accounting=round(byt_full / dt / 1e9 / HBM_PEAK_GBS, 4),
                note="2 kernels (Floor1 unwrap; fused de-interleave + coupling + floor + IMDCT + OLA); whole call; best of 3 loops of "
                     "40.  The mapping declares the residue's support (residue_end = %d of 1024 bins): `bytes` counts the lower "
                     "half of every vector (what is loaded), frac_full_vector_accounting the whole vector as rounds 1-3 did"
                     % FLOOR6_RESIDUE_END)
            # ... as `ReadSamples(Span<float>)` hands a 5.1 stream over: interleaved (IStreamDecoder.cs:126)
            dt, _ = time_decoder(ctx, dec, torch, pk, res6, posts, counts, samples6, 6, 40, 3, layout=capi.OUT_INTERLEAVED)
            extras["configs[3] interleaved out"] = fused_entry(samples6 * 6, dt, byt)
            dec.close()
            # ... with the support NOT declared (what ABI v3 could say): every zero loaded and multiplied
            pk_n, res_n, posts_n, counts_n, floors_n, mappings_n, _ = build_floor6(torch, device, args.extras_frames6, declare_support=False)
            dec_n = Decoder(ctx, 6, 256, 2048, floors=floors_n, mappings=mappings_n)
            dt_n, _ = time_decoder(ctx, dec_n, torch, pk_n, res_n, posts_n, counts_n, samples6, 6, 40, 3)
            extras["configs[3] support not declared"] = fused_entry(samples6 * 6, dt_n, byt_full)
            dec_n.close()
            # ... and as planar [6][1024] packets (what residue types 0 / 1 hand over): the pair route -- the stereo kernel, a workgroup per
            # pair of channels (synth_pairs.hip); on the Residue2 vector above the decoder keeps group mode
            res_p = res6.reshape(args.extras_frames6, 1024, 6).transpose(1, 2).contiguous().reshape(-1)
            pk_p = pk.copy()
            pk_p["flags"] &= np.uint8(~capi.PKT_INTERLEAVED & 0xFF)
            dec_p = Decoder(ctx, 6, 256, 2048, floors=floors, mappings=mappings)
            dt_p, _ = time_decoder(ctx, dec_p, torch, pk_p, res_p, posts, counts, samples6, 6, 40, 3)
            extras["configs[3] planar packets (pair route)"] = fused_entry(samples6 * 6, dt_p, byt)
            dec_p.close()
            del res_n, posts_n, counts_n, res6, res_p, posts, counts
            torch.cuda.empty_cache()
            # the other long block sizes the reference takes (Mdct.cs:15-19): 4096 and 8192, mixed with their short blocks, stereo, already
            # floored (configs[2]'s shape) -- synth_big_kernel, one pass over HBM
            big = {}
            import helpers
            for size0, size1 in ((512, 4096), (1024, 8192)):
                frames_b = args.extras_frames * 2048 // size1
                flags_b = helpers.markov_block_flags(frames_b, seed=3)
                halves = np.where(flags_b & 1, size1 // 2, size0 // 2).astype(np.int64)
                offs_b = np.concatenate([[0], np.cumsum(halves * 2)])
                pk_b = make_packets(frames_b)
                pk_b["flags"] = flags_b | capi.PKT_NO_FLOOR
                pk_b["granule"] = -1
                pk_b["residue_offset"] = offs_b[:-1]
                g = torch.Generator(device=device).manual_seed(size1)
                res_b = torch.randn(int(offs_b[-1]), generator=g, device=device) * 2.0 ** -8
                dec_b = Decoder(ctx, 2, size0, size1)
                cap_b = int(halves.sum()) + 2 * size1
                out_b = torch.empty(2 * cap_b, device=device)
                dec_b.reset(-1)
                smp_b = int(dec_b.synth_raw(pk_b, res_b, None, None, out_b, None, cap_b, capi.OUT_PLANAR, cap_b, capi.MEM_DEVICE)[0])
                del out_b
                dt_b, _ = time_decoder(ctx, dec_b, torch, pk_b, res_b, None, None, smp_b, 2, 20, 3)
                byt_b = 4 * int(offs_b[-1]) + 4 * smp_b * 2
                big["%d/%d" % (size0, size1)] = {"ms_per_step": round(dt_b * 1e3, 3), "frac_of_8TBps": round(byt_b / dt_b / 1e9 / HBM_PEAK_GBS, 4),
                                                 "bytes": byt_b, "frames": frames_b}
                time_decoder.last_spread = None
                dec_b.close()
                del res_b
            extras["long blocks of 4096 / 8192 samples, stereo, mixed with short blocks (synth_big_kernel)"] = big
            torch.cuda.empty_cache()
            # configs[4], one GPU's share: 128 stereo streams (64 x 3test.ogg + 64 x issue6test.ogg)
            dt, tot, t_front, _ = time_real_streams(ctx, torch, device, 64, steps=40, warmup=3)
            extras["configs[4] one GPU's share: 128 real stereo streams, interleaved out, GPU stage"] = fused_entry(
                tot, dt, 8 * tot, traffic=pmc_traffic_bytes("configs4_share"), cpu=cpu_baseline_fused("configs4") if cpu else None,
                cpu_entropy_decode_s_1thread=round(t_front, 3),
                note="64x 3test.ogg + 64x issue6test.ogg, decoded spectra device-resident; one decoder over both fixtures' setups "
                     "(sharding.merge_setups); Floor1 unwrap + the stereo fast path, one launch each per step")
            # ... the same share with the residue as 16-bit integers in device memory (ABI v5; exact for these files): the floored
            # stereo kernel reads them in place -- 2 B in + 4 B out per sample
            dt16, tot16, _, _ = time_real_streams(ctx, torch, device, 64, steps=40, warmup=3, int16=True)
            extras["configs[4] share, int16 residue (6 B per sample)"] = fused_entry(tot16, dt16, 6 * tot16)
            thr = host_threads()
            thr_disp = thr if HOST_THREADS_CAP else thr + 4  # (the dispatcher's own rule: the CPUs plus one per context)
            tot_e, (t_all, t_dec, t_syn) = end_to_end_rank_dispatcher(torch, ctx.device, [64, 64], thr_disp)
            tot_s, (t_all_s, t_dec_s, t_syn_s) = end_to_end_rank_dispatcher(torch, ctx.device, [64, 64], thr_disp, s16=True)
            extras["configs[4] share end to end (container bytes -> PCM in host memory)"] = {
                "f32_Msamples_per_s": round(tot_e / t_all / 1e6, 1), "s16_Msamples_per_s": round(tot_s / t_all_s / 1e6, 1),
                "host_threads": thr_disp, "f32_decode_done_ms": round(t_dec * 1e3, 2), "s16_decode_done_ms": round(t_dec_s * 1e3, 2),
                "host": host_report(thr),
                "note": "ONE vpzm_decode_library call: sub-batches of 16 streams on 4 contexts, int16 residue over the link (ABI v5); best of 3"}
            extras["configs[0] 2test.ogg, CPU only"] = cpu_plumbing_2test()
    if not args.no_extras:
        # ---------------- configs[4], the whole job: 1024 stereo streams partitioned over the ranks (strong scaling).
        # Every rank decodes ONLY its contiguous shard of global stream ids (no data-path collective); what crosses
        # ranks is two int64 vectors: per-stream sample counts and PCM checksums.
        plan = sharding.plan_stream_shard(TOTAL_REAL_STREAMS, world, rank, n_kinds=len(REAL_FIXTURES))
        sync = (lambda: sharding.barrier()) if distributed else None
        reduce_t = (lambda t: sharding.max_over_ranks(t, red_device)) if distributed else None
        dt, tot_local, _, local = time_real_streams(ctx, torch, device, 0, steps=5, warmup=2, plan=plan,
                                                    before_timing=sync, reduce_time=reduce_t)
        samples, sums = sharding.merge_stream_results(TOTAL_REAL_STREAMS, local, red_device)
        # the same two fixtures decoded alone (one stream, one call): what a single-GPU run of the job produces
        _, _, _, solo = time_real_streams(ctx, torch, device, 1, steps=1, warmup=0, repeats=1)
        expect = sharding.combine_stream_checksums([solo[s % 2][1] for s in range(TOTAL_REAL_STREAMS)])
        job_sum = sharding.combine_stream_checksums(sums)
        tot = sum(samples) * 2
        thr = host_threads()
        if distributed:
            sharding.barrier()
        # end to end: the rank's shard through the product's host path (the dispatcher on the rank's own device)
        per_kind = [len(ids) for ids in plan]
        thr_disp = thr if HOST_THREADS_CAP else thr + 4
        tot_e_local, (t_all, t_dec, t_syn) = end_to_end_rank_dispatcher(torch, ctx.device, per_kind, thr_disp)
        if distributed:
            sharding.barrier()
        _, (t_all16, t_dec16, t_syn16) = end_to_end_rank_dispatcher(torch, ctx.device, per_kind, thr_disp, s16=True)
        rank_ms = sharding.gather_floats([t_all * 1e3, t_dec * 1e3, t_syn * 1e3, t_all16 * 1e3, t_dec16 * 1e3, t_syn16 * 1e3],
                                         red_device)
        if distributed:
            sharding.barrier()
            t_all = sharding.max_over_ranks(t_all, red_device)
            t_all16 = sharding.max_over_ranks(t_all16, red_device)
        per_rank = [sum(samples[s] for s in range(*sharding.shard_range(TOTAL_REAL_STREAMS, world, r))) * 2
                    for r in range(world)]
        extras["configs[4] whole job: 1024 real stereo streams over %d GPU(s), interleaved out" % world] = {
            "gpu_stage_Msamples_per_s": round(tot / dt / 1e6, 1), "gpu_stage_ms_per_step": round(dt * 1e3, 3),
            "gpu_stage_frac_of_8TBps_per_gpu": round(8 * tot / world / dt / 1e9 / HBM_PEAK_GBS, 4),
            "end_to_end_Msamples_per_s": round(tot / t_all / 1e6, 1), "end_to_end_ms": round(t_all * 1e3, 2),
            "end_to_end_s16_Msamples_per_s": round(tot / t_all16 / 1e6, 1), "end_to_end_s16_ms": round(t_all16 * 1e3, 2),
            "host_threads_per_rank": thr, "samples_per_rank": per_rank,
            "pcm_checksum": "%016x" % job_sum, "checksum_equals_single_stream_decode": job_sum == expect,
            "collectives_on_the_data_path": 0,
            "end_to_end_path": "every rank: ONE vpzm_decode_library call on its device (libvorbispizza_host.so: open, entropy decode -- "
                               "int16 residue where the setup header guarantees integers, ABI v5 --, host-memory synth calls on 4 contexts)",
            "per_rank_ms": {"end_to_end_wall": [round(r[0], 2) for r in rank_ms],
                            "cpu_open_and_entropy_decode_wall": [round(r[1], 2) for r in rank_ms],
                            "synth_host_memory_calls_incl_h2d_d2h": [round(r[2], 2) for r in rank_ms],
                            "s16_end_to_end_wall": [round(r[3], 2) for r in rank_ms],
                            "s16_cpu_open_and_entropy_decode_wall": [round(r[4], 2) for r in rank_ms],
                            "s16_synth_host_memory_calls": [round(r[5], 2) for r in rank_ms]},
            "host": host_report(thr), "samples_total": tot,
            "streams_per_rank": [sharding.shard_range(TOTAL_REAL_STREAMS, world, r)[1] -
                                 sharding.shard_range(TOTAL_REAL_STREAMS, world, r)[0] for r in range(world)],
            "note": "512x 3test.ogg + 512x issue6test.ogg; strong scaling (1024 streams in total).  GPU stage: decoded spectra "
                    "device-resident, best of 3 loops of 5 steps (barrier before, max over ranks after each); end to end: container "
                    "bytes in host memory -> PCM in host memory incl. CPU entropy decode, best of 3 per rank, max over ranks"}
        # ---------------- the same job in ONE process: rank 0 drives all `world` devices through the in-process dispatcher
        # (include/vorbispizza_multi.h) while the other ranks wait -- both launchers' curves from one driver run
        if distributed:
            sharding.barrier()
        if rank == 0:
            ids = [0] * world if rehearsal else list(range(world))
            thr_all = host_threads(whole_node=True)
            try:
                # (no --host-threads: the dispatcher counts its decode threads itself -- the CPUs this process may use plus one per
                # context; under torchrun the library's own count is this RANK's share of the CPUs: the whole node's is passed instead)
                thr_disp = thr_all if (HOST_THREADS_CAP or world > 1) else None
                one = dispatcher_whole_job(torch, ids, thr_disp)
                one16 = dispatcher_whole_job(torch, ids, thr_disp, s16=True, checksum=False)
                one["checksum_equals_the_per_process_job"] = one.get("pcm_checksum") == "%016x" % job_sum
                extras["configs[4] job in ONE process (vpzm_decode_library), %d device(s), f32 PCM" % world] = one
                extras["configs[4] job in ONE process, s16 PCM"] = one16
            except Exception as e:  # (the leg is an extra: its failure must not cost the contract line)
                extras["configs[4] job in ONE process: FAILED"] = repr(e)
        if distributed:
            sharding.barrier()
    ctx.close()
    if distributed:
        sharding.finalize()
    if rank == 0:
        emit(result, extras)


if __name__ == "__main__":
    main()
