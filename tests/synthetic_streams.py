"""Synthetic Ogg/Vorbis streams (tests/vorbis_writer.py) that exercise what the reference's four fixtures
do not: Floor0, residue type 0, several submaps, ordered / sparse codebooks, lookup type 2, block sizes
other than 256/2048, three modes, and packets continued over page boundaries."""
import numpy as np

import helpers
import vorbis_writer as vw


def _scalar_book(rng, entries, **kw):
    return vw.random_codebook(rng, 1, entries, 0, **kw)


def _floor1(rng, books, half, partitions, multiplier):
    """Appends its books to `books`; two classes: class 0 without subclasses, class 1 with 2 subclass bits."""
    base = len(books)
    books.append(_scalar_book(rng, 4))                    # masterbook of class 1
    books.append(_scalar_book(rng, 8))                    # subclass books
    books.append(_scalar_book(rng, 12, sparse=True))
    books.append(_scalar_book(rng, 6, ordered=True))
    class_dims = [2, 3]
    class_sub = [0, 2]
    class_master = [0, base]
    sub_books = [[base + 1], [base + 2, -1, base + 3, base + 1]]
    part_class = [int(rng.integers(2)) for _ in range(partitions)]
    n_x = sum(class_dims[c] for c in part_class)
    rangebits = vw.ilog(half - 1)
    xs = [int(v) for v in rng.choice(np.arange(1, half), size=n_x, replace=False)]
    return vw.Floor1(part_class, class_dims, class_sub, class_master, sub_books, multiplier, rangebits, xs)


def _residue(rng, books, rtype, half, partition_size, classifications, class_dims, vq):
    """vq: list of (dims, lookup_type, entries) books shared by the classes."""
    base = len(books)
    for dims, lt, entries in vq:
        books.append(vw.random_codebook(rng, dims, entries, lt, sparse=bool(rng.integers(2))))
    books.append(_scalar_book(rng, classifications ** class_dims))
    classbook = len(books) - 1
    cascade, rbooks = [], []
    for c in range(classifications):
        casc = int(rng.integers(0, 8)) if c else 0      # class 0: nothing coded
        if c == classifications - 1:
            casc |= 0x21                                # a high stage too: cascade needs the 5 high bits
        cascade.append(casc)
        rbooks.append([base + int(rng.integers(len(vq))) if casc & (1 << st) else None for st in range(8)])
    begin = int(rng.integers(0, 3)) * partition_size
    end = half - int(rng.integers(0, 2)) * partition_size
    return vw.Residue(rtype, begin, end, partition_size, classbook, cascade, rbooks)


def _lsp_book(rng, dims):
    # ascending-ish LSP coefficients in (0, pi): min > 0, small positive steps, sequence_p
    entries = 16
    return vw.Codebook(dims, [4] * entries, 1,
                       minv=vw.float32_pack(5, 788 - 5), delta=vw.float32_pack(1, 788 - 5), value_bits=2,
                       seq_p=True, mults=[int(v) for v in rng.integers(0, 4, size=vw.lookup1_values(entries, dims))])


def mono_floor1_res1(seed=1):
    """mono, block sizes 64/512, residue 1, three modes."""
    rng = np.random.default_rng(seed)
    books = []
    f_short = _floor1(rng, books, 32, 3, 2)
    f_long = _floor1(rng, books, 256, 6, 1)
    r_short = _residue(rng, books, 1, 32, 8, 3, 2, [(2, 1, 9), (4, 1, 16), (2, 2, 7)])
    r_long = _residue(rng, books, 1, 256, 16, 4, 2, [(2, 1, 25), (4, 2, 10), (8, 1, 256)])
    maps = [vw.Mapping(1, [], [0], [0], [0]), vw.Mapping(1, [], [0], [1], [1])]
    modes = [(0, 0), (1, 1), (1, 1)]
    return vw.Stream(1, 8000, 6, 9, books, [f_short, f_long], [r_short, r_long], maps, modes), rng


def stereo_coupled_res2(seed=2):
    """stereo, one coupling step, residue 2, 256/2048 (the fast GPU path)."""
    rng = np.random.default_rng(seed)
    books = []
    f_short = _floor1(rng, books, 128, 4, 2)
    f_long = _floor1(rng, books, 1024, 8, 4)
    r_short = _residue(rng, books, 2, 256, 16, 3, 2, [(2, 1, 9), (4, 1, 81)])
    r_long = _residue(rng, books, 2, 2048, 32, 4, 3, [(2, 1, 49), (4, 2, 12), (8, 1, 300)])
    maps = [vw.Mapping(2, [(0, 1)], [0, 0], [0], [0]), vw.Mapping(2, [(0, 1)], [0, 0], [1], [1])]
    return vw.Stream(2, 44100, 8, 11, books, [f_short, f_long], [r_short, r_long], maps, [(0, 0), (1, 1)]), rng


def three_channels_two_submaps(seed=3):
    """3 channels in 2 submaps (mux 0,0,1): residue 0 on the pair, residue 1 on the third; coupling (0,1)
    and (2,0); block sizes 128/1024."""
    rng = np.random.default_rng(seed)
    books = []
    f_a = _floor1(rng, books, 64, 3, 3)
    f_b = _floor1(rng, books, 512, 7, 2)
    f_c = _floor1(rng, books, 512, 5, 1)
    r0_short = _residue(rng, books, 0, 64, 8, 3, 2, [(2, 1, 9), (4, 1, 16)])
    r1_short = _residue(rng, books, 1, 64, 8, 2, 1, [(2, 2, 6)])
    r0_long = _residue(rng, books, 0, 512, 16, 3, 2, [(4, 1, 81), (8, 1, 256)])
    r1_long = _residue(rng, books, 1, 512, 32, 3, 2, [(2, 1, 25), (4, 2, 9)])
    coupling = [(0, 1), (2, 0)]
    maps = [vw.Mapping(3, coupling, [0, 0, 1], [0, 0], [0, 1]), vw.Mapping(3, coupling, [0, 0, 1], [1, 2], [2, 3])]
    return (vw.Stream(3, 22050, 7, 10, books, [f_a, f_b, f_c], [r0_short, r1_short, r0_long, r1_long], maps,
                      [(0, 0), (1, 1)]), rng)


def stereo_floor0(seed=4):
    """stereo, Floor0 on both block sizes (two LSP books to choose from), residue 1, equal 512/512 blocks."""
    rng = np.random.default_rng(seed)
    books = []
    books.append(_lsp_book(rng, 2))
    books.append(_lsp_book(rng, 4))
    f0 = vw.Floor0(8, 16000, 128, 6, 40, [0, 1], max_amp_raw=3)
    f1 = vw.Floor0(12, 16000, 200, 5, 30, [1, 0], max_amp_raw=2)
    r = _residue(rng, books, 1, 256, 16, 3, 2, [(2, 1, 25), (4, 1, 81)])
    maps = [vw.Mapping(2, [], [0, 0], [0], [0]), vw.Mapping(2, [(0, 1)], [0, 0], [1], [0])]
    return vw.Stream(2, 16000, 9, 9, books, [f0, f1], [r], maps, [(0, 0), (1, 1)]), rng


def six_channels_51(seed=5):
    """5.1 as libvorbis lays it out: 6 channels in one submap, coupling (0,2) and (3,4) -- front and rear pairs, the
    centre between them --, residue 2, 256/2048: group mode with a real Residue2 vector and non-adjacent pairs."""
    rng = np.random.default_rng(seed)
    books = []
    f_short = _floor1(rng, books, 128, 4, 2)
    f_long = _floor1(rng, books, 1024, 8, 3)
    r_short = _residue(rng, books, 2, 6 * 128, 16, 3, 2, [(2, 1, 9), (4, 1, 81)])
    r_long = _residue(rng, books, 2, 6 * 1024, 32, 4, 3, [(2, 1, 49), (4, 2, 12), (8, 1, 300)])
    coupling = [(0, 2), (3, 4)]
    maps = [vw.Mapping(6, coupling, [0] * 6, [0], [0]), vw.Mapping(6, coupling, [0] * 6, [1], [1])]
    return vw.Stream(6, 48000, 8, 11, books, [f_short, f_long], [r_short, r_long], maps, [(0, 0), (1, 1)]), rng


def three_channels_chained(seed=7):
    """3 channels, two CHAINED coupling steps (0,1) then (2,0) -- the inverse must run them in reverse order --, residue 1,
    block sizes 512/1024 (the general-size kernels)."""
    rng = np.random.default_rng(seed)
    books = []
    f_short = _floor1(rng, books, 256, 5, 2)
    f_long = _floor1(rng, books, 512, 6, 1)
    r_short = _residue(rng, books, 1, 256, 16, 3, 2, [(2, 1, 9), (4, 1, 16)])
    r_long = _residue(rng, books, 1, 512, 32, 3, 2, [(2, 1, 25), (4, 2, 9)])
    coupling = [(0, 1), (2, 0)]
    maps = [vw.Mapping(3, coupling, [0, 0, 0], [0], [0]), vw.Mapping(3, coupling, [0, 0, 0], [1], [1])]
    return vw.Stream(3, 22050, 9, 10, books, [f_short, f_long], [r_short, r_long], maps, [(0, 0), (1, 1)]), rng


def _many_channels(seed, channels, coupling):
    rng = np.random.default_rng(seed)
    books = []
    f_short = _floor1(rng, books, 128, 4, 2)
    f_long = _floor1(rng, books, 1024, 7, 3)
    r_short = _residue(rng, books, 2, channels * 128, 16, 3, 2, [(2, 1, 9), (4, 1, 81)])
    r_long = _residue(rng, books, 2, channels * 1024, 32, 4, 3, [(2, 1, 49), (4, 2, 12), (8, 1, 300)])
    maps = [vw.Mapping(channels, coupling, [0] * channels, [0], [0]), vw.Mapping(channels, coupling, [0] * channels, [1], [1])]
    return vw.Stream(channels, 48000, 8, 11, books, [f_short, f_long], [r_short, r_long], maps, [(0, 0), (1, 1)]), rng


def five_channels(seed=8):
    """5 channels (an odd count: Residue2 vectors of 5 floats per bin), coupling (0,1) and (3,4), 256/2048."""
    return _many_channels(seed, 5, [(0, 1), (3, 4)])


def ten_channels(seed=9):
    """10 channels -- more than a workgroup holds: the separate coupling pass --, coupling (0,1), (2,3), (8,9), 256/2048."""
    return _many_channels(seed, 10, [(0, 1), (2, 3), (8, 9)])


def four_channels_quad(seed=6):
    """4 channels, coupling (0,1) and (2,3), residue 2, 256/2048."""
    rng = np.random.default_rng(seed)
    books = []
    f_short = _floor1(rng, books, 128, 3, 1)
    f_long = _floor1(rng, books, 1024, 7, 2)
    r_short = _residue(rng, books, 2, 4 * 128, 8, 3, 2, [(2, 1, 9), (4, 1, 16)])
    r_long = _residue(rng, books, 2, 4 * 1024, 32, 4, 2, [(2, 1, 25), (4, 2, 10), (8, 1, 256)])
    coupling = [(0, 1), (2, 3)]
    maps = [vw.Mapping(4, coupling, [0] * 4, [0], [0]), vw.Mapping(4, coupling, [0] * 4, [1], [1])]
    return vw.Stream(4, 44100, 8, 11, books, [f_short, f_long], [r_short, r_long], maps, [(0, 0), (1, 1)]), rng


ALL = {"mono_floor1_res1": mono_floor1_res1, "stereo_coupled_res2": stereo_coupled_res2,
       "three_channels_two_submaps": three_channels_two_submaps, "stereo_floor0": stereo_floor0,
       "six_channels_51": six_channels_51, "four_channels_quad": four_channels_quad, "three_channels_chained": three_channels_chained,
       "five_channels": five_channels, "ten_channels": ten_channels}


def random_stream(seed):
    """A random setup: 1-3 channels, any block-size pair 64..4096, one or two submaps, floor 1 (or, one time in
    four, floor 0 on every submap), residue types 0 / 1 / 2 at random, 0-2 coupling steps, one or two long modes."""
    rng = np.random.default_rng(seed)
    channels = int(rng.integers(1, 4))
    logs = sorted(int(v) for v in rng.integers(6, 13, size=2))      # block sizes 64 .. 4096
    h0, h1 = (1 << logs[0]) // 2, (1 << logs[1]) // 2
    books = []
    submaps = 2 if (channels > 1 and rng.random() < 0.4) else 1
    mux = [0] * channels if submaps == 1 else [int(v) for v in rng.integers(0, 2, size=channels)]
    if submaps == 2 and len(set(mux)) == 1:
        mux[-1] = 1 - mux[0]
    floors, residues, maps = [], [], []
    use_floor0 = rng.random() < 0.25
    if use_floor0:
        lsp = len(books)
        books += [_lsp_book(rng, 2), _lsp_book(rng, 4)]
    for half in (h0, h1):
        sub_floor, sub_res = [], []
        for s in range(submaps):
            if use_floor0:
                # bark_map_size <= the smaller half block: the reference indexes its w map out of range otherwise
                # 16 amplitude bits and an amplitude capped where the curve reaches +20 dB: random LSP roots make
                # far peakier filters than an encoder would (exp(0.115 * (amp / sqrt(p + q) - ofs)))
                fl = vw.Floor0(int(rng.choice([4, 8, 12])), 16000, int(rng.integers(8, h0 + 1)), 16,
                               int(rng.integers(45, 64)), [lsp, lsp + 1][: int(rng.integers(1, 3))])
                fl.amp_policy = (lambda c, fl=fl: helpers.floor0_safe_amp(c, fl.bark_map_size, fl.amp_ofs) * 1.3)
                floors.append(fl)
            else:
                floors.append(_floor1(rng, books, half, int(rng.integers(2, 6)) if half >= 64 else 2, int(rng.integers(1, 5))))
            sub_floor.append(len(floors) - 1)
            members = sum(1 for m in mux if m == s)
            rtype = int(rng.integers(0, 3))
            size = half * members if rtype == 2 else half
            psize = int(rng.choice([8, 16, 32])) if size >= 128 else 8
            residues.append(_residue(rng, books, rtype, size, psize, int(rng.integers(2, 5)), int(rng.integers(1, 3)),
                                     [(2, 1, 9), (4, 1, 16), (2, 2, 7), (8, 1, 256)][: int(rng.integers(2, 5))]))
            sub_res.append(len(residues) - 1)
        coupling = []
        if channels > 1:
            for _ in range(int(rng.integers(0, 3))):
                m, a = (int(v) for v in rng.choice(channels, size=2, replace=False))
                coupling.append((m, a))
        maps.append(vw.Mapping(channels, coupling, mux, sub_floor, sub_res))
    modes = [(0, 0), (1, 1)] + ([(1, 1)] if rng.random() < 0.3 else [])
    return vw.Stream(channels, int(rng.choice([8000, 22050, 44100])), logs[0], logs[1], books, floors, residues, maps,
                     modes), rng
