"""Containers whose SETUP header was damaged after encoding (page checksums made valid again): shared by
tests/test_hostile_setup_gpu.py and tools/make_hostile_setup_seeds.py.

A setup packet is ~4 KB of codebooks, floors, residues, mappings and modes (StreamDecoder.cs:262-321); a random change of one
to three of its bytes mostly breaks a codebook (the front end refuses the stream, like the reference's InvalidDataException),
and in about a quarter of the cases yields a stream that still OPENS: other floor X lists and multipliers, other residue
ranges, partition sizes and books, other coupling pairs, submaps and modes than any encoder wrote."""
import json
import os
import struct

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SEEDS_FILE = os.path.join(GOLDEN, "hostile_setup_seeds.json")
FIXTURES = ("1test.ogg", "2test.ogg", "3test.ogg", "issue6test.ogg")
WRITER_STREAMS = ("stereo_coupled_res2", "three_channels_two_submaps", "stereo_floor0", "six_channels_51", "mono_floor1_res1")


def pages(raw):
    out, pos = [], 0
    while pos + 27 <= len(raw) and raw[pos:pos + 4] == b"OggS":
        nseg = raw[pos + 26]
        lac = list(raw[pos + 27: pos + 27 + nseg])
        out.append((pos, 27 + nseg, lac))
        pos += 27 + nseg + sum(lac)
    return out


def packet_spans(raw, want):
    """byte ranges [(start, end)] of packet `want` of the file's (single) logical stream, page by page"""
    spans, idx = [], 0
    for pos, hdr, lac in pages(raw):
        at = pos + hdr
        for v in lac:
            if idx == want and v > 0:
                spans.append((at, at + v))
            at += v
            if v < 255:
                idx += 1
                if idx > want:
                    return spans
    return spans


def mutate_setup(raw, seed):
    """one to three bytes of the setup packet (the third header packet, behind its "\\x05vorbis") changed -- a bit flipped,
    a byte replaced, a byte moved by a few counts --, the checksums of the pages it touches recomputed"""
    import vorbis_writer as vw
    rng = np.random.default_rng(seed)
    spans = packet_spans(raw, 2)
    total = sum(e - s for s, e in spans)
    data = bytearray(raw)
    for _ in range(int(rng.integers(1, 4))):
        k = int(rng.integers(7, total))
        at = None
        for s, e in spans:
            if k < e - s:
                at = s + k
                break
            k -= e - s
        kind = int(rng.integers(0, 3))
        if kind == 0:
            data[at] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            data[at] = int(rng.integers(0, 256))
        else:
            data[at] = (data[at] + int(rng.integers(1, 4)) * (1 if rng.integers(0, 2) else -1)) & 255
    for pos, hdr, lac in pages(raw):
        body = sum(lac)
        if any(s < pos + hdr + body and e > pos + hdr for s, e in spans):
            page = bytes(data[pos:pos + 22]) + b"\0\0\0\0" + bytes(data[pos + 26:pos + hdr + body])
            data[pos + 22:pos + 26] = struct.pack("<I", vw._crc(page))
    return bytes(data)


def sources():
    """name -> clean container bytes: the reference's four fixtures and five streams of the spec-based writer"""
    import synthetic_streams as ss
    out = {}
    for name in FIXTURES:
        out[name] = open(os.path.join(GOLDEN, name), "rb").read()
    for name in WRITER_STREAMS:
        stream, rng = ss.ALL[name]()
        ogg, _ = stream.build(rng, 24)
        out[name] = bytes(ogg)
    return out


def committed_cases():
    """[(source name, seed, damaged container)] of the committed campaign (tests/golden/hostile_setup_seeds.json)"""
    seeds = json.load(open(SEEDS_FILE))["seeds"]
    src = sources()
    return [(name, seed, mutate_setup(src[name], seed)) for name in sorted(seeds) for seed in seeds[name]]


# ---- setups at and beyond the edges of what the headers can say, built with the specification-based writer
def _floor1_with_posts(rng, books, half, n_posts, multiplier, xs=None):
    """a type-1 floor with exactly n_posts posts (2 + the partitions' dimensions); xs: its X values, else random distinct ones"""
    import vorbis_writer as vw
    base = len(books)
    books.append(vw.random_codebook(rng, 1, 8, 0))
    need = n_posts - 2
    dims = []
    while need > 0:
        d = min(8, need)
        dims.append(d)
        need -= d
    class_dims = sorted(set(dims))
    part_class = [class_dims.index(d) for d in dims]
    rangebits = vw.ilog(half - 1)
    if xs is None:
        xs = [int(v) for v in rng.choice(np.arange(1, 1 << rangebits), size=n_posts - 2, replace=False)]
    return vw.Floor1(part_class, class_dims, [0] * len(class_dims), [0] * len(class_dims), [[base]] * len(class_dims),
                     multiplier, rangebits, xs)


def _plain_residue(rng, books, rtype, begin, end, partition_size, classifications=2):
    import vorbis_writer as vw
    base = len(books)
    books.append(vw.random_codebook(rng, 2, 9, 1))
    books.append(vw.random_codebook(rng, 1, classifications, 0))
    cascade = [0] + [1] * (classifications - 1)
    rbooks = [[None] * 8] + [[base] + [None] * 7 for _ in range(classifications - 1)]
    return vw.Residue(rtype, begin, end, partition_size, base + 1, cascade, rbooks)


def crafted():
    """name -> (container bytes, what to expect: "pcm" or "refused")"""
    import vorbis_writer as vw
    out = {}

    def stream(name, channels, logs, books, floors, residues, maps, modes, packets=20, expect="pcm", seed=0):
        rng = np.random.default_rng(1000 + seed)
        st = vw.Stream(channels, 44100, logs[0], logs[1], books, floors, residues, maps, modes)
        ogg, _ = st.build(rng, packets)
        out[name] = (bytes(ogg), expect)

    # 64 posts (all `Posts = new int[64]` holds, Floor1.cs:17), multiplier 4, every X list value the range bits allow
    rng = np.random.default_rng(11)
    books = []
    fl = [_floor1_with_posts(rng, books, 128, 64, 4), _floor1_with_posts(rng, books, 1024, 64, 4)]
    rs = [_plain_residue(rng, books, 2, 0, 256, 8), _plain_residue(rng, books, 2, 0, 2048, 32)]
    maps = [vw.Mapping(2, [(0, 1)], [0, 0], [0], [0]), vw.Mapping(2, [(1, 0)], [0, 0], [1], [1])]
    stream("floor1_64_posts_multiplier_4", 2, (8, 11), books, fl, rs, maps, [(0, 0), (1, 1)], seed=1)
    # 65 posts: the specification's maximum, one more than the reference's array -- its Unpack would throw on every packet
    rng = np.random.default_rng(12)
    books = []
    fl = [_floor1_with_posts(rng, books, 128, 19, 2), _floor1_with_posts(rng, books, 1024, 65, 1)]
    rs = [_plain_residue(rng, books, 1, 0, 128, 8), _plain_residue(rng, books, 1, 0, 1024, 32)]
    maps = [vw.Mapping(2, [], [0, 0], [0], [0]), vw.Mapping(2, [], [0, 0], [1], [1])]
    stream("floor1_65_posts", 2, (8, 11), books, fl, rs, maps, [(0, 0), (1, 1)], expect="refused", seed=2)
    # X lists that crowd one end: 0, 1, 2, 3 ... and the block's last bins (segments of one bin, a flat tail of none)
    rng = np.random.default_rng(13)
    books = []
    fl = [_floor1_with_posts(rng, books, 128, 12, 1, xs=list(range(1, 6)) + list(range(123, 128))),
          _floor1_with_posts(rng, books, 1024, 30, 3, xs=list(range(1, 15)) + list(range(1010, 1024)))]
    rs = [_plain_residue(rng, books, 2, 0, 256, 8), _plain_residue(rng, books, 2, 0, 2048, 32)]
    maps = [vw.Mapping(2, [(0, 1)], [0, 0], [0], [0]), vw.Mapping(2, [(0, 1)], [0, 0], [1], [1])]
    stream("floor1_x_lists_at_the_ends", 2, (8, 11), books, fl, rs, maps, [(0, 0), (1, 1)], seed=3)
    # residue ranges the block does not have: begin beyond the block, end of 2^24 - 1, begin > end, partitions of two bins
    rng = np.random.default_rng(14)
    books = []
    fl = [_floor1_with_posts(rng, books, 128, 10, 2), _floor1_with_posts(rng, books, 1024, 20, 2)]
    rs = [_plain_residue(rng, books, 1, 4000, (1 << 24) - 1, 8), _plain_residue(rng, books, 1, 900, 40, 2),
          _plain_residue(rng, books, 0, 0, (1 << 24) - 1, 4), _plain_residue(rng, books, 2, 6, 2046, 2)]
    maps = [vw.Mapping(2, [], [0, 1], [0, 0], [0, 2]), vw.Mapping(2, [(1, 0)], [0, 1], [1, 1], [1, 3])]
    stream("residue_ranges_beyond_the_block", 2, (8, 11), books, fl, rs, maps, [(0, 0), (1, 1)], seed=4)
    # 24 floors, 12 mappings, 12 modes: every mode its own mapping, every mapping its own pair of floors
    rng = np.random.default_rng(15)
    books = []
    fl, rs, maps, modes = [], [], [], []
    for k in range(12):
        long_ = k % 2
        half = 1024 if long_ else 128
        fl += [_floor1_with_posts(rng, books, half, 8 + k, 1 + k % 4), _floor1_with_posts(rng, books, half, 20 - k, 1 + (k + 1) % 4)]
        rs.append(_plain_residue(rng, books, 2 if k % 3 else 1, 0, half * (2 if k % 3 else 1), 16))
        maps.append(vw.Mapping(2, [(0, 1)] if k % 3 else [], [0, 1], [2 * k, 2 * k + 1], [k, k]))
        modes.append((long_, k))
    stream("twelve_modes_twenty_four_floors", 2, (8, 11), books, fl, rs, maps, modes, packets=60, seed=5)
    # 8 channels, 8 coupling steps that chain through every channel and come back (group mode's levels), two submaps
    rng = np.random.default_rng(16)
    books = []
    fl = [_floor1_with_posts(rng, books, 128, 9, 2), _floor1_with_posts(rng, books, 1024, 21, 2),
          _floor1_with_posts(rng, books, 128, 5, 1), _floor1_with_posts(rng, books, 1024, 33, 4)]
    rs = [_plain_residue(rng, books, 2, 0, 128 * 4, 16), _plain_residue(rng, books, 2, 0, 1024 * 4, 32)]
    chain = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 0)]
    mux = [0, 1, 0, 1, 0, 1, 0, 1]
    maps = [vw.Mapping(8, chain, mux, [0, 2], [0, 0]), vw.Mapping(8, chain, mux, [1, 3], [1, 1])]
    stream("eight_channels_coupled_in_a_ring", 8, (8, 11), books, fl, rs, maps, [(0, 0), (1, 1)], packets=16, seed=6)
    # 40 channels (beyond group mode), 16 submaps, 39 coupling steps in a chain, small blocks
    rng = np.random.default_rng(17)
    books = []
    fl, rs = [], []
    for s in range(16):
        fl += [_floor1_with_posts(rng, books, 32, 4 + s % 5, 1 + s % 4)]
        rs.append(_plain_residue(rng, books, s % 3, 0, 32 * (3 if s % 3 == 2 else 1), 4))
    mux = [c % 16 for c in range(40)]
    for s in range(16):  # (a type-2 residue's vector length is the submap's channel count times the half block)
        members = sum(1 for m in mux if m == s)
        if rs[s].type == 2:
            rs[s].end = 32 * members
    chain = [(c, c + 1) for c in range(39)]
    maps = [vw.Mapping(40, chain, mux, list(range(16)), list(range(16)))]
    stream("forty_channels_sixteen_submaps", 40, (6, 6), books, fl, rs, maps, [(0, 0)], packets=12, seed=7)
    return out
