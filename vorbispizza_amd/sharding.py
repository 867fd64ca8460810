"""Multi-GPU sharding of independent streams / channel-block batches.

Vorbis streams are independent, so the path shards embarrassingly: one process per GPU, a
contiguous range of streams (or of a batch) per rank, NO collective on the data path.  The only
communication is the barrier and the max-over-ranks of the elapsed time that bench.py needs; it goes
through torch.distributed (backend "nccl" = RCCL on the GPUs, "gloo" in the CPU tests).
"""
import os


def shard_range(n_items, world_size, rank):
    """Contiguous partition: rank r owns [start, stop); item s belongs to rank s // ceil(n/world)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    per = -(-n_items // world_size) if n_items > 0 else 0
    start = min(n_items, rank * per)
    return start, min(n_items, start + per)


def owner_of(item, n_items, world_size):
    per = -(-n_items // world_size) if n_items > 0 else 1
    return min(world_size - 1, item // max(per, 1))


def plan_stream_shard(n_streams, world_size, rank, n_kinds=1):
    """The streams a rank decodes: its contiguous range of GLOBAL stream ids, split by kind (global stream s is of
    kind s % n_kinds -- e.g. which setup header / fixture it uses -- so that every rank gets the same mix and one
    decoder group per kind).  Returns [ids of kind 0, ids of kind 1, ...]."""
    lo, hi = shard_range(n_streams, world_size, rank)
    return [[s for s in range(lo, hi) if s % n_kinds == k] for k in range(n_kinds)]


def merge_setups(setups):
    """One decoder configuration for streams that come with DIFFERENT setup headers (same channel count and block
    sizes): the union of their floors (equal ones shared) and of their mappings.  setups: [(floors, mappings), ...] in the
    form `Decoder` takes.  Returns (floors, mappings, mapping_base): a packet of setup k with mapping index m carries
    mapping_base[k] + m in the merged batch.  A synth call then covers all the streams at once -- one launch, longer runs
    per wavefront -- instead of one call per setup (the reference has one StreamDecoder per stream, StreamDecoder.cs:45-49;
    how streams are batched is this back end's business)."""
    floors, mappings, bases = [], [], []

    def key(fl):
        if isinstance(fl, dict):
            return ("f0",) + tuple(sorted((k, tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in fl.items()))
        return ("f1", tuple(int(x) for x in fl[0]), int(fl[1]))

    index = {}
    for fls, mps in setups:
        remap = []
        for fl in fls:
            k = key(fl)
            if k not in index:
                index[k] = len(floors)
                floors.append(fl)
            remap.append(index[k])
        bases.append(len(mappings))
        for m in mps:
            merged = {"coupling": [tuple(p) for p in m["coupling"]],
                      "channel_floor": [remap[f] for f in m["channel_floor"]]}
            for k in ("residue_begin", "residue_end"):  # (ABI v4: the residue's support travels with its mapping)
                if k in m:
                    merged[k] = tuple(m[k])
            mappings.append(merged)
    if len(mappings) > 256:
        raise ValueError("merge_setups: more than 256 mappings do not fit a packet's mapping index")
    return floors, mappings, bases


_MASK64 = (1 << 64) - 1


def combine_stream_checksums(checksums):
    """Checksum of per-stream checksums, independent of which rank decoded which stream: {global id: c} or a
    sequence indexed by global id -> sum of (id + 1) * c mod 2^64."""
    items = checksums.items() if hasattr(checksums, "items") else enumerate(checksums)
    total = 0
    for sid, c in items:
        total = (total + (int(sid) + 1) * (int(c) & _MASK64)) & _MASK64
    return total


def merge_stream_results(n_streams, local, device="cpu"):
    """local: {global stream id: (samples, checksum)} of the streams THIS rank decoded.  Returns two lists indexed
    by global id -- samples per channel and PCM checksum of every stream of the job -- after one SUM all-reduce of
    two int64 vectors (every stream has exactly one owner; the PCM itself never crosses ranks)."""
    import torch
    import torch.distributed as dist
    t = torch.zeros((2, n_streams), dtype=torch.int64)
    for sid, (samples, checksum) in local.items():
        c = int(checksum) & _MASK64
        t[0, sid] = int(samples)
        t[1, sid] = c - (1 << 64) if c >= (1 << 63) else c   # two's complement: int64 carries the 64 bits
    if dist.is_available() and dist.is_initialized():
        t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t = t.cpu()
    return [int(v) for v in t[0]], [int(v) & _MASK64 for v in t[1]]


def env_world():
    """(world_size, rank, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend):
    """Initialise torch.distributed when launched with WORLD_SIZE > 1; returns (world, rank, local_rank)."""
    world, rank, local_rank = env_world()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if not dist.is_initialized():
            dist.init_process_group(backend, rank=rank, world_size=world)
    return world, rank, local_rank


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX-reduce a python float over all ranks (identity when not distributed)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(values, device="cpu"):
    """Every rank's list of python floats, as a list indexed by rank (one SUM all-reduce of a [world, len] matrix in
    which a rank fills only its own row; [values] when not distributed).  Diagnostics only -- per-rank timings."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [[float(v) for v in values]]
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.zeros((world, len(values)), dtype=torch.float64)
    t[rank] = torch.tensor([float(v) for v in values], dtype=torch.float64)
    t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [[float(v) for v in row] for row in t.cpu()]


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
