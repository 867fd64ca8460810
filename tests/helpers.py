"""Shared test helpers: workload generators and the oracle-side driver that
plays the role vpz_decoder_synth plays on the GPU."""
import ctypes as C

import numpy as np

PKT_BLOCK_FLAG, PKT_PREV_FLAG, PKT_NEXT_FLAG, PKT_EOS = 0x01, 0x02, 0x04, 0x08
PKT_NOT_DECODED, PKT_INTERLEAVED, PKT_NO_FLOOR = 0x10, 0x20, 0x40

# X list of a libvorbis 44.1 kHz long-block floor1 (29 posts, the shape 3test.ogg's long floor has)
LONG_XLIST = [0, 1024, 93, 23, 372, 6, 46, 186, 750, 14, 33, 65, 130, 260, 556, 3, 10, 18, 28, 39, 55,
              79, 111, 158, 220, 312, 464, 650, 850]
SHORT_XLIST = [0, 128, 12, 46, 4, 8, 16, 23, 33, 70, 2, 6, 10, 14, 19, 28, 39, 58, 90]


def markov_block_flags(frames, seed, p_ls=0.1, p_sl=0.3, start_long=True):
    """Config 3 block-flag chain: p(long->short)=0.1, p(short->long)=0.3; returns the packet flag
    bytes with prev/next window flags made mutually consistent (StreamDecoder.cs:654,778)."""
    rng = np.random.default_rng(seed)
    u = rng.random(frames)
    bf = np.zeros(frames, dtype=np.uint8)
    cur = 1 if start_long else 0
    for i in range(frames):
        bf[i] = cur
        cur = (0 if u[i] < p_ls else 1) if cur else (1 if u[i] < p_sl else 0)
    prev = np.concatenate([[1], bf[:-1]])
    nxt = np.concatenate([bf[1:], [1]])
    return (bf * PKT_BLOCK_FLAG | prev * PKT_PREV_FLAG * bf | nxt * PKT_NEXT_FLAG * bf).astype(np.uint8)


def gaussian_spectra(shape, seed, sigma=2.0 ** -8):
    return (np.random.default_rng(seed).standard_normal(shape) * sigma).astype(np.float32)


PKT_RESYNC = 0x80


class OracleStream:
    """One reference StreamDecoder, restated: the oracle's stream state plus what Read()'s `while (idx == 0)` loop does
    around ReadNextPacket (StreamDecoder.cs:418-498).  Packets are dicts with keys flags, granule (default -1), mapping
    (default 0), residue (np [channels*half] as laid out for the ABI), posts (np [channels, <=64]), post_count
    (np [channels]), for type-0 floors f0_amp / f0_coeff."""

    def __init__(self, orc, channels, size0, size1, floors=(), mappings=(), clip=False, interleave=False, state=None):
        self.orc, self.L = orc, orc.lib()
        self.channels, self.size0, self.size1 = channels, size0, size1
        self.floors, self.mappings, self.clip, self.interleave = floors, mappings, clip, interleave
        self.st = state if state is not None else self.L.orc_stream_create(channels, size0, size1)
        self.ofl = [orc.floor1_init(*f) if not isinstance(f, dict) else None for f in floors]
        self.eos_seen = False
        self.mismatches = 0

    def close(self):
        if self.st is not None:
            self.L.orc_stream_destroy(self.st)
            self.st = None

    def take(self):
        """Hands out everything readable, like repeated Read calls; returns [channels, n] or None."""
        L, st, channels = self.L, self.st, self.channels
        n = L.orc_stream_available(st)
        if n <= 0:
            return None
        if self.interleave:
            buf = np.zeros((n, channels), dtype=np.float32)
            L.orc_stream_store(st, buf.ctypes.data_as(C.POINTER(C.c_float)), 0, n, 0, 1, int(self.clip))
            return buf.T.copy()
        buf = np.zeros((channels, n), dtype=np.float32)
        L.orc_stream_store(st, buf.ctypes.data_as(C.POINTER(C.c_float)), 0, n, n, 0, int(self.clip))
        return buf

    def read_next_packet(self, pk):
        """DecodeNextPacket + ReadNextPacket for one packet (StreamDecoder.cs:640-762); returns ReadNextPacket's result
        (1 accepted, 0 no packet, -1 where OverlapBuffers throws)."""
        orc, L, st = self.orc, self.L, self.st
        channels, size0, size1, floors, mappings = self.channels, self.size0, self.size1, self.floors, self.mappings
        flags = pk["flags"]
        eos = 1 if flags & PKT_EOS else 0
        if flags & PKT_RESYNC:
            L.orc_stream_mark_resync(st)  # :718-722, before the packet's first bit is looked at
        if flags & PKT_NOT_DECODED:
            L.orc_stream_read_next_packet(st, 0, None, -1, eos)
            return 0
        bf = 1 if flags & PKT_BLOCK_FLAG else 0
        n = size1 if bf else size0
        half = n // 2
        info = orc.packet_info(size0, size1, bf, bool(flags & PKT_PREV_FLAG), bool(flags & PKT_NEXT_FLAG))
        res = np.asarray(pk["residue"], dtype=np.float32)
        if flags & PKT_INTERLEAVED:
            res = res.reshape(half, channels).T.copy()
        else:
            res = res.reshape(channels, half)
        if flags & PKT_NO_FLOOR:
            pcm = np.stack([orc.mdct_reverse(res[c][None, :], n)[0] for c in range(channels)])
        elif any(isinstance(f, dict) for f in floors):
            # mixed floor types: Mapping.cs:166-195 step by step (type-0 floors: Floor0.cs:164-225)
            m = mappings[pk.get("mapping", 0)]
            res = res.copy()
            for mag, ang in reversed(m.get("coupling", [])):
                res[mag], res[ang] = orc.apply_coupling(res[mag], res[ang])
            pcm = np.zeros((channels, n), dtype=np.float32)
            for c in range(channels):
                fl = floors[m.get("channel_floor", [0] * channels)[c]]
                if isinstance(fl, dict):
                    amp = float(pk["f0_amp"][c])
                    if amp == 0:
                        continue  # ExecuteChannel false (Floor0.cs:22)
                    spec = orc.floor0_apply(fl["order"], fl["rate"], fl["bark_map_size"], fl["amp_bits"], fl["amp_ofs"],
                                            pk["f0_coeff"][c][:fl["order"]], amp, n, res[c])
                else:
                    if pk["post_count"][c] == 0:
                        continue
                    spec = orc.floor1_apply(orc.floor1_init(*fl), pk["posts"][c], int(pk["post_count"][c]), n, res[c])
                pcm[c] = orc.mdct_reverse(spec[None, :], n)[0]
        else:
            m = mappings[pk.get("mapping", 0)]
            pcm = orc.mapping_synth(channels, n, res, self.ofl, m.get("channel_floor", [0] * channels),
                                    pk["posts"], pk["post_count"], m.get("coupling", []))
        p = L.orc_stream_next_buffer(st)
        view = np.ctypeslib.as_array(p, shape=(channels, size1))
        view[:] = 0
        view[:, :n] = pcm
        return L.orc_stream_read_next_packet(st, 1, C.byref(info), int(pk.get("granule", -1)), eos)

    def feed(self, pk):
        """One iteration of Read()'s loop for a packet: returns the samples it made readable ([channels, n] or None)."""
        if self.eos_seen and self.L.orc_stream_available(self.st) == 0:
            return None  # Read(): nothing more is read after EOS (StreamDecoder.cs:441-447)
        rc = self.read_next_packet(pk)
        eos = bool(pk["flags"] & PKT_EOS)
        if pk["flags"] & PKT_NOT_DECODED:
            if eos:
                self.eos_seen = True
                self.L.orc_stream_drain_eos(self.st)
                return self.take()
            return None
        if eos:
            self.eos_seen = True
        if rc < 0:
            # OverlapBuffers would throw (StreamDecoder.cs:777-778): that Read fails, the packet is
            # consumed, the decoder state stays as it was
            self.mismatches += 1
            return None
        return self.take()

    @property
    def position(self):
        return self.L.orc_stream_position(self.st)

    @property
    def has_clipped(self):
        return bool(self.L.orc_stream_has_clipped(self.st))


def oracle_decode(orc, channels, size0, size1, packets, floors=(), mappings=(), clip=False,
                  interleave=False, state=None, keep_state=False):
    """Runs one stream through the oracle (see OracleStream for the packet dicts).  Returns PCM [channels, T] or
    [T, channels], the position and the clip flag."""
    # `state`: continue on an oracle stream a previous call kept (keep_state=True returns it as a 4th value,
    # e.g. to put an orc_stream_reset between two calls)
    s = OracleStream(orc, channels, size0, size1, floors, mappings, clip, interleave, state=state)
    chunks = []
    for pk in packets:
        got = s.feed(pk)
        if got is not None:
            chunks.append(got)
    pos, clipped, st = s.position, s.has_clipped, s.st
    if not keep_state:
        s.close()
    pcm = np.concatenate(chunks, axis=1) if chunks else np.zeros((channels, 0), dtype=np.float32)
    oracle_decode.last_mismatches = s.mismatches
    if keep_state:
        return (pcm.T.copy() if interleave else pcm), pos, clipped, st
    return (pcm.T.copy() if interleave else pcm), pos, clipped


def random_posts(rng, xlist, multiplier, n_ch, silent_prob=0.0):
    """Raw floor1 posts as `Floor1.Unpack` leaves them: two absolute values then residuals."""
    rng_range = {1: 256, 2: 128, 3: 86, 4: 64}[multiplier]
    posts = np.zeros((n_ch, 64), dtype=np.int16)
    counts = np.zeros(n_ch, dtype=np.uint8)
    for c in range(n_ch):
        if rng.random() < silent_prob:
            continue
        counts[c] = len(xlist)
        posts[c, 0] = rng.integers(rng_range // 4, rng_range // 2)
        posts[c, 1] = rng.integers(rng_range // 8, rng_range // 3)
        vals = rng.integers(0, 12, size=len(xlist) - 2)
        vals[rng.random(len(vals)) < 0.35] = 0
        posts[c, 2:len(xlist)] = vals
    return posts, counts


def packets_for_oracle(ogg, pk, residue, posts, counts):
    """Converts the front end's batch arrays (vorbispizza_amd.front.OggVorbisFile.decode_packets) into
    the per-packet dicts oracle_decode takes."""
    C_ = ogg.channels
    out = []
    for i in range(len(pk)):
        flags = int(pk["flags"][i])
        d = {"flags": flags, "granule": int(pk["granule"][i]), "mapping": int(pk["mapping"][i])}
        if not flags & PKT_NOT_DECODED:
            half = (ogg.block_size1 if flags & PKT_BLOCK_FLAG else ogg.block_size0) // 2
            off = int(pk["residue_offset"][i])
            d["residue"] = residue[off: off + C_ * half]
            d["posts"] = posts[i * C_:(i + 1) * C_]
            d["post_count"] = counts[i * C_:(i + 1) * C_]
            if getattr(ogg, "floor0_data", None) is not None:
                d["f0_amp"] = ogg.floor0_data[0][i * C_:(i + 1) * C_]
                d["f0_coeff"] = ogg.floor0_data[1][i * C_:(i + 1) * C_]
        out.append(d)
    return out


def floor0_safe_amp(coeff, bark_map_size, amp_ofs):
    """Largest amp for which Floor0's curve stays <= 0 dB: amp_ofs * min_k sqrt(p(k) + q(k))."""
    c = 2.0 * np.cos(coeff.astype(np.float64))
    w = 2.0 * np.cos(np.pi / bark_map_size * np.arange(bark_map_size))
    p = np.full_like(w, 0.5)
    q = np.full_like(w, 0.5)
    order = len(c)
    j = 1
    while j < order:
        q *= w - c[j - 1]
        p *= w - c[j]
        j += 2
    if j == order:
        q *= w - c[j - 1]
        p *= p * (4.0 - w * w)
        q *= q
    else:
        p *= p * (2.0 - w)
        q *= q * (2.0 + w)
    return float(amp_ofs * np.sqrt(np.maximum(p + q, 1e-30)).min())
