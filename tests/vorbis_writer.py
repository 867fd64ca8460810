"""A small Ogg/Vorbis bitstream WRITER for tests, written from the Vorbis I specification (not from the
reference's code): it emits setup headers with arbitrary codebooks / floors / residues / mappings / modes and
audio packets whose every symbol is chosen at random, and it keeps its own model of what a decoder must
recover from them (raw floor posts, Floor0 amplitude + coefficients, residue vectors).  The C++ front end
(vorbispizza_amd/host/vorbis_front.cpp), which follows the reference's C#, is checked against that model --
two independent implementations of the entropy stage.

Where the reference deviates from the specification the model follows the REFERENCE and says so:
  * residue type 0 sums a VQ entry into ONE bin (Residue0.cs:211-230);
  * Floor0.Unpack reads the book number and the vectors even when the amplitude is 0 (Floor0.cs:122-146),
    so the writer never emits a zero amplitude for a type-0 floor.
"""
import struct

import numpy as np


def ilog(x):
    n = 0
    while x > 0:
        n += 1
        x >>= 1
    return n


class BitWriter:
    """LSB-first bit packing (Vorbis I section 2)."""

    def __init__(self):
        self.acc = 0
        self.n = 0

    def write(self, value, bits):
        assert 0 <= value < (1 << bits) or bits == 0, (value, bits)
        self.acc |= value << self.n
        self.n += bits

    def bytes(self):
        nbytes = (self.n + 7) // 8
        return self.acc.to_bytes(nbytes, "little")


# ------------------------------------------------------------------------------------------------ Ogg
_CRC = []
for _i in range(256):
    _r = _i << 24
    for _ in range(8):
        _r = ((_r << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if _r & 0x80000000 else (_r << 1) & 0xFFFFFFFF
    _CRC.append(_r)


def _crc(data):
    c = 0
    for b in data:
        c = ((c << 8) & 0xFFFFFFFF) ^ _CRC[((c >> 24) & 0xFF) ^ b]
    return c


def ogg_mux(packets, granules, serial=0x1234, packets_per_page=3, max_segments=255):
    """packets[i] bytes, granules[i] the granule position after packet i (stamped on the page that
    completes it; a page that completes nothing carries -1).  Packet 0 sits alone on the first page, the
    other two headers share the following page(s), audio packets go `packets_per_page` to a page; a page
    holds at most `max_segments` lacing values, so longer packets continue on the next page."""
    breaks = {0, 2}
    i = 2
    while i < len(packets):
        i += packets_per_page
        breaks.add(min(i, len(packets) - 1))
    raw = []  # (continued, granule, segs, body)
    segs, body, last_done, cont = [], b"", None, False

    def flush(next_cont):
        nonlocal segs, body, last_done, cont
        raw.append((cont, granules[last_done] if last_done is not None else -1, segs, body))
        segs, body, last_done, cont = [], b"", None, next_cont

    for idx, data in enumerate(packets):
        lac = [255] * (len(data) // 255) + [len(data) % 255]
        pos = 0
        for k, v in enumerate(lac):
            if len(segs) == max_segments:
                flush(k > 0)
            segs.append(v)
            body += data[pos:pos + v]
            pos += v
        last_done = idx
        if idx in breaks:
            flush(False)
    if segs:
        flush(False)
    pages = []
    for seq, (c, gran, sg, bd) in enumerate(raw):
        flags = (1 if c else 0) | (2 if seq == 0 else 0) | (4 if seq == len(raw) - 1 else 0)
        page = b"OggS" + struct.pack("<BBqIIIB", 0, flags, gran, serial, seq, 0, len(sg)) + bytes(sg) + bd
        page = page[:22] + struct.pack("<I", _crc(page)) + page[26:]
        pages.append(page)
    return b"".join(pages)


# ------------------------------------------------------------------------------------------- codebooks
def float32_pack(mantissa, exponent, negative=False):
    """Vorbis float32: value = (-1)^s * mantissa * 2^(exponent - 788); returns (bits, float32 value)."""
    assert 0 <= mantissa < (1 << 21) and 0 <= exponent < 1024
    bits = (0x80000000 if negative else 0) | (exponent << 21) | mantissa
    val = np.float32(np.ldexp(float(-mantissa if negative else mantissa), exponent - 788))
    return bits, val


def lookup1_values(entries, dims):
    r = 0
    while (r + 1) ** dims <= entries:
        r += 1
    return r


class Codebook:
    def __init__(self, dims, lengths, lookup_type=0, minv=None, delta=None, value_bits=4, seq_p=False,
                 mults=None, ordered=False, sparse=False):
        self.dims, self.lengths, self.entries = dims, list(lengths), len(lengths)
        self.lookup_type, self.minv, self.delta = lookup_type, minv, delta
        self.value_bits, self.seq_p, self.mults = value_bits, seq_p, mults
        self.ordered, self.sparse = ordered, sparse
        self.codes = self._assign()
        self.used = [i for i, l in enumerate(self.lengths) if l > 0]

    def _assign(self):
        """Entry i takes the lowest-valued unused codeword of its length (spec 3.2.1)."""
        root = {}
        codes = {}

        def place(node, depth, prefix):
            if node.get("leaf"):
                return None
            if depth == 0:
                if node.get(0) or node.get(1):
                    return None
                node["leaf"] = True
                return prefix
            for b in (0, 1):
                child = node.setdefault(b, {})
                got = place(child, depth - 1, prefix + [b])
                if got is not None:
                    return got
                if not child:
                    del node[b]
            return None

        for i, l in enumerate(self.lengths):
            if l <= 0:
                continue
            c = place(root, l, [])
            assert c is not None, "over-specified codebook"
            codes[i] = c
        return codes

    def write_header(self, bw):
        bw.write(0x564342, 24)
        bw.write(self.dims, 16)
        bw.write(self.entries, 24)
        if self.ordered:
            assert all(l > 0 for l in self.lengths) and self.lengths == sorted(self.lengths)
            bw.write(1, 1)
            cur = self.lengths[0]
            bw.write(cur - 1, 5)
            i = 0
            while i < self.entries:
                cnt = sum(1 for l in self.lengths if l == cur)
                bw.write(cnt, ilog(self.entries - i))
                i += cnt
                cur += 1
        else:
            bw.write(0, 1)
            bw.write(1 if self.sparse else 0, 1)
            for l in self.lengths:
                if self.sparse:
                    bw.write(1 if l > 0 else 0, 1)
                    if l > 0:
                        bw.write(l - 1, 5)
                else:
                    assert l > 0
                    bw.write(l - 1, 5)
        bw.write(self.lookup_type, 4)
        if self.lookup_type:
            bw.write(self.minv[0], 32)
            bw.write(self.delta[0], 32)
            bw.write(self.value_bits - 1, 4)
            bw.write(1 if self.seq_p else 0, 1)
            for m in self.mults:
                bw.write(int(m), self.value_bits)

    def n_mults(self):
        return lookup1_values(self.entries, self.dims) if self.lookup_type == 1 else self.entries * self.dims

    def vector(self, e):
        """VQ vector of entry e (spec 3.2.1), float32 arithmetic in the order value*delta + min + last."""
        out = np.zeros(self.dims, dtype=np.float32)
        last = np.float32(0)
        if self.lookup_type == 1:
            lv = lookup1_values(self.entries, self.dims)
            div = 1
            for i in range(self.dims):
                moff = (e // div) % lv
                v = np.float32(np.float32(np.float32(self.mults[moff]) * self.delta[1]) + self.minv[1]) + last
                out[i] = v
                if self.seq_p:
                    last = np.float32(v)
                div *= lv
        else:
            for i in range(self.dims):
                v = np.float32(np.float32(np.float32(self.mults[e * self.dims + i]) * self.delta[1]) + self.minv[1]) + last
                out[i] = v
                if self.seq_p:
                    last = np.float32(v)
        return out

    def write_entry(self, bw, e):
        for b in self.codes[e]:
            bw.write(b, 1)


def random_codebook(rng, dims, entries, lookup_type, max_len=20, ordered=False, sparse=False):
    """A complete (Kraft sum == 1) random prefix code over `entries` symbols (all used unless sparse)."""
    used = entries if not sparse else max(2, entries - entries // 4)
    # build lengths by splitting leaves at random
    leaves = [1, 1]
    while len(leaves) < used:
        i = int(rng.integers(len(leaves)))
        if max_len and leaves[i] >= max_len:
            i = int(np.argmin(leaves))
        l = leaves.pop(i)
        leaves += [l + 1, l + 1]
    lengths = sorted(leaves) if ordered else list(rng.permutation(leaves))
    if sparse:
        full = [0] * entries
        for pos, l in zip(sorted(rng.choice(entries, size=used, replace=False)), lengths):
            full[pos] = int(l)
        lengths = full
    kw = {}
    if lookup_type:
        value_bits = int(rng.integers(2, 6))
        kw = dict(minv=float32_pack(int(rng.integers(1, 40)), 788 - int(rng.integers(0, 3)), negative=True),
                  delta=float32_pack(int(rng.integers(1, 8)), 788 - int(rng.integers(0, 2))),
                  value_bits=value_bits, seq_p=bool(rng.integers(2)) and lookup_type == 1)
        n = lookup1_values(entries, dims) if lookup_type == 1 else entries * dims
        kw["mults"] = [int(v) for v in rng.integers(0, 1 << value_bits, size=n)]
    return Codebook(dims, [int(l) for l in lengths], lookup_type, ordered=ordered, sparse=sparse, **kw)


# ------------------------------------------------------------------------------------------ setup parts
class Floor1:
    RANGES = [256, 128, 86, 64]

    def __init__(self, partition_class, class_dims, class_subclasses, class_masterbook, subclass_books,
                 multiplier, rangebits, xs):
        self.partition_class, self.class_dims = partition_class, class_dims
        self.class_subclasses, self.class_masterbook, self.subclass_books = class_subclasses, class_masterbook, subclass_books
        self.multiplier, self.rangebits, self.xs = multiplier, rangebits, xs
        self.x_list = [0, 1 << rangebits] + list(xs)

    type = 1

    def write_header(self, bw):
        bw.write(len(self.partition_class), 5)
        for c in self.partition_class:
            bw.write(c, 4)
        for c in range(max(self.partition_class) + 1):  # the header carries classes 0 .. max used (spec 7.2.2)
            bw.write(self.class_dims[c] - 1, 3)
            bw.write(self.class_subclasses[c], 2)
            if self.class_subclasses[c]:
                bw.write(self.class_masterbook[c], 8)
            for b in self.subclass_books[c]:
                bw.write(b + 1, 8)
        bw.write(self.multiplier - 1, 2)
        bw.write(self.rangebits, 4)
        for x in self.xs:
            bw.write(x, self.rangebits)

    def write_packet(self, bw, books, rng, silent):
        """Returns (post_count, raw posts list)."""
        if silent:
            bw.write(0, 1)
            return 0, []
        bw.write(1, 1)
        rng_range = self.RANGES[self.multiplier - 1]
        ybits = ilog(rng_range - 1)
        posts = [int(rng.integers(rng_range // 4, rng_range // 2)), int(rng.integers(rng_range // 8, rng_range // 3))]
        bw.write(posts[0], ybits)
        bw.write(posts[1], ybits)
        for cls in self.partition_class:
            cdim, cbits = self.class_dims[cls], self.class_subclasses[cls]
            csub = (1 << cbits) - 1
            cval = 0
            if cbits:
                mb = books[self.class_masterbook[cls]]
                cval = int(rng.choice(mb.used))
                mb.write_entry(bw, cval)
            for _ in range(cdim):
                book = self.subclass_books[cls][cval & csub]
                cval >>= cbits
                if book >= 0:
                    # small residuals keep the unwrapped curve inside the dB table
                    cand = [e for e in books[book].used if e < 6] or books[book].used
                    e = int(rng.choice(cand))
                    books[book].write_entry(bw, e)
                    posts.append(e)
                else:
                    posts.append(0)
        return len(posts), posts


class Floor0:
    type = 0

    def __init__(self, order, rate, bark_map_size, amp_bits, amp_ofs, book_list, max_amp_raw=None):
        self.order, self.rate, self.bark_map_size = order, rate, bark_map_size
        self.amp_bits, self.amp_ofs, self.book_list = amp_bits, amp_ofs, book_list
        # random LSP roots make a far peakier filter than an encoder would; a small amplitude keeps the
        # curve exp(0.115 * (amp / sqrt(p + q) - amp_ofs)) inside float range
        self.max_amp_raw = max_amp_raw or (1 << amp_bits) - 1
        self.amp_policy = None  # optional: coeff -> largest amplitude (float) that keeps the curve in range

    def write_header(self, bw):
        bw.write(self.order, 8)
        bw.write(self.rate, 16)
        bw.write(self.bark_map_size, 16)
        bw.write(self.amp_bits, 6)
        bw.write(self.amp_ofs, 8)
        bw.write(len(self.book_list) - 1, 4)
        for b in self.book_list:
            bw.write(b, 8)

    def write_packet(self, bw, books, rng):
        """Returns (amp float32, coeff float32[order]) -- never a zero amplitude (see module docstring).  The book
        entries are drawn first so that `amp_policy(coeff) -> largest sensible amplitude` (if set) can keep the curve
        in range; the bitstream order is amplitude, book number, entries."""
        bi = int(rng.integers(len(self.book_list)))
        book = books[self.book_list[bi]]
        assert self.order % book.dims == 0  # (else the reference's `last` differs from the spec's; keep them equal)
        entries, coeff = [], []
        last = np.float32(0)
        while len(coeff) < self.order:
            e = int(rng.choice(book.used))
            entries.append(e)
            chunk = [np.float32(v + last) for v in book.vector(e)]
            last = chunk[-1]
            coeff += chunk
        coeff = np.array(coeff[: self.order], dtype=np.float32)
        top = self.max_amp_raw
        if self.amp_policy is not None:
            limit = self.amp_policy(coeff) * ((1 << self.amp_bits) - 1) / float(self.amp_ofs)
            top = max(1, min(top, int(limit)))
        amp_raw = int(rng.integers(max(1, top // 2), top + 1))
        bw.write(amp_raw, self.amp_bits)
        amp = np.float32(amp_raw * self.amp_ofs / float((1 << self.amp_bits) - 1))
        bw.write(bi, ilog(len(self.book_list)))
        for e in entries:
            book.write_entry(bw, e)
        return amp, coeff


class Residue:
    def __init__(self, rtype, begin, end, partition_size, classbook, cascade, books):
        """books[class][stage] = codebook index or None"""
        self.type, self.begin, self.end, self.partition_size = rtype, begin, end, partition_size
        self.classbook, self.cascade, self.books = classbook, cascade, books
        self.classifications = len(cascade)

    def write_header(self, bw):
        bw.write(self.begin, 24)
        bw.write(self.end, 24)
        bw.write(self.partition_size - 1, 24)
        bw.write(self.classifications - 1, 6)
        bw.write(self.classbook, 8)
        for c in self.cascade:
            bw.write(c & 7, 3)
            if c >> 3:
                bw.write(1, 1)
                bw.write(c >> 3, 5)
            else:
                bw.write(0, 1)
        for c, casc in enumerate(self.cascade):
            for st in range(8):
                if casc & (1 << st):
                    bw.write(self.books[c][st], 8)

    def write_packet(self, bw, books, rng, do_not_decode, half, out=None):
        """Spec 8.6.2 with the reference's layouts.  Accumulates into (and returns) vectors [ch][half] f32."""
        n_ch = len(do_not_decode)
        if out is None:
            out = np.zeros((n_ch, half), dtype=np.float32)
        begin, end = min(self.begin, half), min(self.end, half)
        n = end - begin
        if n <= 0:
            return out
        parts = n // self.partition_size
        cb = books[self.classbook]
        D = cb.dims
        max_stage = max(ilog(c) for c in self.cascade)
        classes = [[0] * (parts + D) for _ in range(n_ch)]
        for stage in range(max_stage):
            p = 0
            while p < parts:
                if stage == 0:
                    for ch in range(n_ch):
                        if do_not_decode[ch]:
                            continue
                        cls = [int(rng.integers(self.classifications)) for _ in range(D)]
                        word = 0
                        for c in cls:
                            word = word * self.classifications + c
                        assert cb.lengths[word] > 0
                        cb.write_entry(bw, word)
                        classes[ch][p:p + D] = cls
                for i in range(D):
                    if p >= parts:
                        break
                    offset = begin + p * self.partition_size
                    for ch in range(n_ch):
                        if do_not_decode[ch]:
                            continue
                        cls = classes[ch][p]
                        if not (self.cascade[cls] & (1 << stage)):
                            continue
                        book = books[self.books[cls][stage]]
                        if self.type == 0:
                            steps = self.partition_size // book.dims
                            for s in range(steps):
                                e = int(rng.choice(book.used))
                                book.write_entry(bw, e)
                                r = np.float32(0)
                                for v in book.vector(e):  # reference quirk q9: the entry is SUMMED into one bin
                                    r = np.float32(r + v)
                                out[ch, offset + s] = np.float32(out[ch, offset + s] + r)
                        else:
                            k = 0
                            while k < self.partition_size:
                                e = int(rng.choice(book.used))
                                book.write_entry(bw, e)
                                vec = book.vector(e)
                                out[ch, offset + k: offset + k + book.dims] += vec
                                k += book.dims
                    p += 1
        return out


class Mapping:
    def __init__(self, channels, coupling, mux, submap_floor, submap_residue):
        self.channels, self.coupling, self.mux = channels, coupling, mux
        self.submap_floor, self.submap_residue = submap_floor, submap_residue

    def write_header(self, bw):
        bw.write(0, 16)
        submaps = len(self.submap_floor)
        if submaps > 1:
            bw.write(1, 1)
            bw.write(submaps - 1, 4)
        else:
            bw.write(0, 1)
        if self.coupling:
            bw.write(1, 1)
            bw.write(len(self.coupling) - 1, 8)
            for m, a in self.coupling:
                bw.write(m, ilog(self.channels - 1))
                bw.write(a, ilog(self.channels - 1))
        else:
            bw.write(0, 1)
        bw.write(0, 2)
        if submaps > 1:
            for c in range(self.channels):
                bw.write(self.mux[c], 4)
        for s in range(submaps):
            bw.write(0, 8)
            bw.write(self.submap_floor[s], 8)
            bw.write(self.submap_residue[s], 8)


class Stream:
    """A whole logical stream: setup + random audio packets + the decode model's expectations."""

    def __init__(self, channels, rate, bs0_log, bs1_log, books, floors, residues, mappings, modes):
        self.channels, self.rate, self.bs0, self.bs1 = channels, rate, 1 << bs0_log, 1 << bs1_log
        self.bs_logs = (bs0_log, bs1_log)
        self.books, self.floors, self.residues, self.mappings, self.modes = books, floors, residues, mappings, modes

    def headers(self):
        ident = BitWriter()
        for c in b"\x01vorbis":
            ident.write(c, 8)
        ident.write(0, 32)
        ident.write(self.channels, 8)
        ident.write(self.rate, 32)
        for _ in range(3):
            ident.write(0, 32)
        ident.write(self.bs_logs[0], 4)
        ident.write(self.bs_logs[1], 4)
        ident.write(1, 1)
        comment = BitWriter()
        for c in b"\x03vorbis":
            comment.write(c, 8)
        vendor = b"vorbis_writer.py"
        comment.write(len(vendor), 32)
        for c in vendor:
            comment.write(c, 8)
        comment.write(0, 32)
        comment.write(1, 1)
        setup = BitWriter()
        for c in b"\x05vorbis":
            setup.write(c, 8)
        setup.write(len(self.books) - 1, 8)
        for b in self.books:
            b.write_header(setup)
        setup.write(0, 6)
        setup.write(0, 16)
        setup.write(len(self.floors) - 1, 6)
        for f in self.floors:
            setup.write(f.type, 16)
            f.write_header(setup)
        setup.write(len(self.residues) - 1, 6)
        for r in self.residues:
            setup.write(r.type, 16)
            r.write_header(setup)
        setup.write(len(self.mappings) - 1, 6)
        for m in self.mappings:
            m.write_header(setup)
        setup.write(len(self.modes) - 1, 6)
        for blockflag, mapping in self.modes:
            setup.write(blockflag, 1)
            setup.write(0, 16)
            setup.write(0, 16)
            setup.write(mapping, 8)
        setup.write(1, 1)
        return [ident.bytes(), comment.bytes(), setup.bytes()]

    def audio_packet(self, rng, mode_idx, prev_flag, next_flag, silent_prob=0.15):
        """Returns (bytes, expectation dict)."""
        bw = BitWriter()
        bw.write(0, 1)
        bw.write(mode_idx, ilog(len(self.modes) - 1))
        blockflag, mapping_idx = self.modes[mode_idx]
        if blockflag:
            bw.write(prev_flag, 1)
            bw.write(next_flag, 1)
        n = self.bs1 if blockflag else self.bs0
        half = n // 2
        mp = self.mappings[mapping_idx]
        C = self.channels
        posts = np.zeros((C, 64), dtype=np.int16)
        counts = np.zeros(C, dtype=np.uint8)
        f0_amp = np.zeros(C, dtype=np.float32)
        f0_coeff = [None] * C
        no_energy = [False] * C
        for ch in range(C):
            fl = self.floors[mp.submap_floor[mp.mux[ch]]]
            if fl.type == 0:
                amp, coeff = fl.write_packet(bw, self.books, rng)
                f0_amp[ch], f0_coeff[ch] = amp, coeff
                counts[ch] = 1
            else:
                pc, raw = fl.write_packet(bw, self.books, rng, silent=rng.random() < silent_prob)
                counts[ch] = pc
                posts[ch, :min(pc, posts.shape[1])] = raw[:posts.shape[1]]  # (a 65-post floor: what the expectation array holds)
                no_energy[ch] = pc == 0
        for m, a in mp.coupling:
            if not (no_energy[m] and no_energy[a]):
                no_energy[m] = no_energy[a] = False
        residue = np.zeros((C, half), dtype=np.float32)
        # the reference decodes every submap into ONE scratch buffer that it never clears between submaps
        # (Mapping.cs:132-160): residue 0/1 accumulate on top of what an earlier submap left in slot k, and a
        # do-not-decode channel copies that left-over out.  Model it.
        scratch = np.zeros((C, half), dtype=np.float32)
        for s in range(len(mp.submap_floor)):
            members = [c for c in range(C) if mp.mux[c] == s]
            dnd = [no_energy[c] for c in members]
            res = self.residues[mp.submap_residue[s]]
            if res.type == 2:
                if all(dnd):
                    scratch[:len(members)] = 0
                else:
                    inter = res.write_packet(bw, self.books, rng, [False], half * len(members))[0]
                    for k in range(len(members)):
                        scratch[k] = inter[k::len(members)]
            else:
                res.write_packet(bw, self.books, rng, dnd, half, out=scratch[:len(members)])
            for k, c in enumerate(members):
                residue[c] = scratch[k]
        exp = {"blockflag": blockflag, "prev": prev_flag, "next": next_flag, "mapping": mapping_idx, "posts": posts,
               "post_count": counts, "residue": residue, "f0_amp": f0_amp, "f0_coeff": f0_coeff}
        return bw.bytes(), exp

    def build(self, rng, n_packets, packets_per_page=3, max_segments=255):
        """Returns (ogg bytes, [expectations])."""
        long_modes = [i for i, (bf, _) in enumerate(self.modes) if bf]
        short_modes = [i for i, (bf, _) in enumerate(self.modes) if not bf]
        seq = []
        for _ in range(n_packets):
            pool = long_modes if (long_modes and (not short_modes or rng.random() < 0.6)) else short_modes
            seq.append(int(rng.choice(pool)))
        pkts, exps, grans = self.headers(), [], [0, 0, 0]
        total = 0
        for i, mi in enumerate(seq):
            bf = self.modes[mi][0]
            prev = self.modes[seq[i - 1]][0] if i else 1
            nxt = self.modes[seq[i + 1]][0] if i + 1 < n_packets else 1
            data, exp = self.audio_packet(rng, mi, prev, nxt)
            pkts.append(data)
            exps.append(exp)
            # granule: samples finished after this packet (spec 4.3.8): none for the first packet
            if i:
                pb = self.modes[seq[i - 1]][0]
                total += ((self.bs1 if pb else self.bs0) + (self.bs1 if bf else self.bs0)) // 4
            grans.append(total)
        return ogg_mux(pkts, grans, packets_per_page=packets_per_page, max_segments=max_segments), exps
