"""world_size-2 gloo test of the N > 1 path: stream partitioning (no data-path collective), the
barrier and the max-over-ranks timing reduction bench.py uses."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
os.environ["VPZ_NO_TORCH"] = "1"
from vorbispizza_amd import sharding
world, rank, local_rank = sharding.init("gloo")
assert world == 2
n_streams = 1024 + 3
lo, hi = sharding.shard_range(n_streams, world, rank)
owned = list(range(lo, hi))
assert all(sharding.owner_of(s, n_streams, world) == rank for s in owned)
# every rank "processes" its own streams: the per-rank work is just a checksum here
local_samples = sum(1000 + s for s in owned)
sharding.barrier()
t0 = time.perf_counter()
time.sleep(0.05 * (rank + 1))          # rank 1 is the slow one
elapsed = time.perf_counter() - t0
slowest = sharding.max_over_ranks(elapsed)
total = sharding.sum_over_ranks(local_samples)
assert slowest >= 0.1 - 1e-3
# configs[4]: 1024 streams of two kinds (two fixtures / setup headers), every rank decodes only its shard and the
# per-stream results (sample count, PCM checksum) are merged with one all-reduce
plan = sharding.plan_stream_shard(1024, world, rank, n_kinds=2)
assert [len(p) for p in plan] == [256, 256] and all(s %% 2 == k for k, ids in enumerate(plan) for s in ids)
fake = lambda s: (288094 if s %% 2 == 0 else 548160, (0x9E3779B97F4A7C15 * (s %% 2 + 1)) & ((1 << 64) - 1))
local = {s: fake(s) for ids in plan for s in ids}
samples, sums = sharding.merge_stream_results(1024, local)
assert samples == [fake(s)[0] for s in range(1024)] and sums == [fake(s)[1] for s in range(1024)]
job_checksum = sharding.combine_stream_checksums(sums)
# per-rank diagnostics of bench.py's whole-job record: every rank sees every rank's row
rows = sharding.gather_floats([10.0 + rank, 0.5 * (rank + 1)])
assert rows == [[10.0, 0.5], [11.0, 1.0]]
if rank == 0:
    print(json.dumps({"lo": lo, "hi": hi, "slowest": slowest, "total": total, "job_checksum": job_checksum,
                      "job_samples": sum(samples)}))
sharding.finalize()
'''


def test_shard_range_partitions_exactly():
    sys.path.insert(0, ROOT)
    from vorbispizza_amd import sharding
    for n in (0, 1, 7, 8, 1024, 1027):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = sharding.shard_range(n, world, r)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert sharding.shard_range(1024, 8, 3) == (384, 512)  # 128 streams per GPU (BASELINE configs[4])
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


@pytest.mark.timeout(180)
def test_two_rank_gloo_run(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29511", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=170)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert (res["lo"], res["hi"]) == (0, 514)
    assert res["total"] == sum(1000 + s for s in range(1027))
    assert res["slowest"] >= 0.099
    # the merged job equals what one process computes over all 1024 streams
    from vorbispizza_amd import sharding
    fake = lambda s: (288094 if s % 2 == 0 else 548160, (0x9E3779B97F4A7C15 * (s % 2 + 1)) & ((1 << 64) - 1))
    assert res["job_samples"] == sum(fake(s)[0] for s in range(1024))
    assert res["job_checksum"] == sharding.combine_stream_checksums([fake(s)[1] for s in range(1024)])
    solo_samples, solo_sums = sharding.merge_stream_results(1024, {s: fake(s) for s in range(1024)})
    assert sharding.combine_stream_checksums(solo_sums) == res["job_checksum"]


def test_stream_plan_covers_every_stream_once_and_keeps_the_mix():
    sys.path.insert(0, ROOT)
    from vorbispizza_amd import sharding
    for world in (1, 2, 4, 8, 3):
        seen = []
        for r in range(world):
            plan = sharding.plan_stream_shard(1024, world, r, n_kinds=2)
            seen += [s for ids in plan for s in ids]
            if 1024 % (2 * world) == 0:
                assert len(plan[0]) == len(plan[1]) == 512 // world
        assert sorted(seen) == list(range(1024))
    assert sharding.combine_stream_checksums({0: 5, 3: 7}) == 5 + 4 * 7


def test_merge_setups_shares_equal_floors_and_shifts_the_mappings():
    from vorbispizza_amd import sharding
    a = ([([0, 128, 64], 2), ([0, 1024, 512, 256], 2)],
         [{"coupling": [(0, 1)], "channel_floor": [0, 0]}, {"coupling": [(0, 1)], "channel_floor": [1, 1]}])
    b = ([([0, 128, 32], 1), ([0, 1024, 512, 256], 2)],
         [{"coupling": [], "channel_floor": [0, 0]}, {"coupling": [(1, 0)], "channel_floor": [1, 0]}])
    floors, mappings, bases = sharding.merge_setups([a, b])
    assert bases == [0, 2]
    assert floors == [([0, 128, 64], 2), ([0, 1024, 512, 256], 2), ([0, 128, 32], 1)]  # the equal floor is shared
    assert [m["channel_floor"] for m in mappings] == [[0, 0], [1, 1], [2, 2], [1, 2]]
    assert [m["coupling"] for m in mappings] == [[(0, 1)], [(0, 1)], [], [(1, 0)]]
    # a single setup comes back as it was
    f1, m1, b1 = sharding.merge_setups([a])
    assert f1 == a[0] and b1 == [0] and [m["channel_floor"] for m in m1] == [[0, 0], [1, 1]]
