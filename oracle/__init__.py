"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (vorbispizza_amd) never does.  PARITY: "parity unpinned" by reference golden
vectors (none exist for this path); see oracle/vorbis_synth_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile oracle/vorbis_synth_oracle.c with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "vorbis_synth_oracle.c"))
    ):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _LIB_PATH


class PacketInfo(C.Structure):
    """PacketInfo.cs:3-15"""
    _fields_ = [("Length", C.c_int32), ("LeftUseSize1", C.c_int32), ("LeftStart", C.c_int32),
                ("LeftEnd", C.c_int32), ("RightStart", C.c_int32), ("RightEnd", C.c_int32)]

    @property
    def SampleCount(self):
        return self.RightStart - self.LeftStart


class Floor0(C.Structure):
    _fields_ = [("order", C.c_int), ("rate", C.c_int), ("bark_map_size", C.c_int), ("amp_bits", C.c_int),
                ("amp_ofs", C.c_int)]


class Floor1(C.Structure):
    _fields_ = [("count", C.c_int), ("multiplier", C.c_int), ("range", C.c_int),
                ("xlist", C.c_int * 65), ("lneigh", C.c_int * 65), ("hneigh", C.c_int * 65),
                ("sortidx", C.c_int * 65)]


READ_NEXT_PACKET_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)  # one ReadNextPacket of the caller's packet provider


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        f32p, i32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint8)
        L.orc_mdct_reverse_batch.argtypes = [C.c_int, C.c_long, f32p, f32p]
        L.orc_mdct_reverse_batch.restype = None
        L.orc_window_slope.argtypes = [C.c_int, f32p]
        L.orc_window_slope.restype = None
        L.orc_get_packet_info.argtypes = [C.c_int] * 5 + [C.POINTER(PacketInfo)]
        L.orc_get_packet_info.restype = None
        L.orc_apply_coupling.argtypes = [f32p, f32p, C.c_int, C.c_int]
        L.orc_apply_coupling.restype = None
        L.orc_residue2_deinterleave.argtypes = [f32p, C.c_int, C.c_int, f32p, C.c_int]
        L.orc_residue2_deinterleave.restype = None
        L.orc_floor1_init.argtypes = [C.POINTER(Floor1), i32p, C.c_int, C.c_int]
        L.orc_floor1_init.restype = C.c_int
        L.orc_floor1_unwrap_posts.argtypes = [C.POINTER(Floor1), i32p, C.c_int, u8p]
        L.orc_floor1_unwrap_posts.restype = None
        L.orc_floor1_apply.argtypes = [C.POINTER(Floor1), i32p, C.c_int, C.c_int, f32p]
        L.orc_floor1_apply.restype = None
        L.orc_floor1_render.argtypes = [C.POINTER(Floor1), i32p, u8p, C.c_int, C.c_int, f32p]
        L.orc_floor1_render.restype = None
        L.orc_floor1_render_indices.argtypes = [C.POINTER(Floor1), i32p, u8p, C.c_int, C.c_int, i32p]
        L.orc_floor1_render_indices.restype = None
        L.orc_floor1_inverse_db_table.argtypes = []
        L.orc_floor1_inverse_db_table.restype = f32p
        L.orc_floor0_apply.argtypes = [C.POINTER(Floor0), f32p, C.c_float, C.c_int, f32p]
        L.orc_floor0_apply.restype = C.c_int
        L.orc_floor0_bark_map.argtypes = [C.POINTER(Floor0), C.c_int, i32p]
        L.orc_floor0_bark_map.restype = None
        L.orc_clip_value.argtypes = [C.c_float, i32p]
        L.orc_clip_value.restype = C.c_float
        L.orc_mapping_synth.argtypes = [C.c_int, C.c_int, f32p, C.c_int, C.POINTER(Floor1), i32p,
                                        i32p, i32p, u8p, u8p, C.c_int, C.c_int]
        L.orc_mapping_synth.restype = None
        L.orc_stream_create.argtypes = [C.c_int] * 3
        L.orc_stream_create.restype = C.c_void_p
        L.orc_stream_destroy.argtypes = [C.c_void_p]
        L.orc_stream_destroy.restype = None
        L.orc_stream_reset.argtypes = [C.c_void_p]
        L.orc_stream_reset.restype = None
        L.orc_stream_mark_resync.argtypes = [C.c_void_p]
        L.orc_stream_mark_resync.restype = None
        L.orc_stream_seek_to.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, READ_NEXT_PACKET_FN, C.c_void_p]
        L.orc_stream_seek_to.restype = C.c_int
        L.orc_stream_next_buffer.argtypes = [C.c_void_p]
        L.orc_stream_next_buffer.restype = f32p
        L.orc_stream_read_next_packet.argtypes = [C.c_void_p, C.c_int, C.POINTER(PacketInfo),
                                                  C.c_int64, C.c_int]
        L.orc_stream_read_next_packet.restype = C.c_int
        L.orc_stream_available.argtypes = [C.c_void_p]
        L.orc_stream_available.restype = C.c_int
        L.orc_stream_drain_eos.argtypes = [C.c_void_p]
        L.orc_stream_drain_eos.restype = None
        L.orc_stream_store.argtypes = [C.c_void_p, f32p, C.c_long, C.c_int, C.c_long, C.c_int, C.c_int]
        L.orc_stream_store.restype = None
        L.orc_stream_has_clipped.argtypes = [C.c_void_p]
        L.orc_stream_has_clipped.restype = C.c_int
        L.orc_stream_position.argtypes = [C.c_void_p]
        L.orc_stream_position.restype = C.c_int64
        L.orc_synth_stream_planar.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long, u8p, f32p, f32p,
                                              C.c_long, C.c_int]
        L.orc_synth_stream_planar.restype = C.c_long
        i64p, i16p = C.POINTER(C.c_int64), C.POINTER(C.c_int16)
        L.orc_synth_stream_floored.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long, u8p, u8p, i64p, i64p, f32p,
                                               C.POINTER(Floor1), i32p, u8p, u8p, i32p, i16p, u8p, f32p, C.c_long, C.c_int]
        L.orc_synth_stream_floored.restype = C.c_long
        _lib = L
    return _lib


# ---------------------------------------------------------------- numpy-facing helpers
def mdct_reverse(spectra, n):
    """Mdct.Reverse (Mdct.cs:15-19) per row: spectra [count, n/2] f32 -> [count, n] f32."""
    spectra = np.ascontiguousarray(spectra, dtype=np.float32).reshape(-1, n // 2)
    out = np.empty((spectra.shape[0], n), dtype=np.float32)
    lib().orc_mdct_reverse_batch(n, spectra.shape[0], _fp(spectra), _fp(out))
    return out


def window_slope(half):
    out = np.empty(half, dtype=np.float32)
    lib().orc_window_slope(half, _fp(out))
    return out


def packet_info(size0, size1, block_flag, prev_flag=True, next_flag=True):
    info = PacketInfo()
    lib().orc_get_packet_info(size0, size1, int(block_flag), int(prev_flag), int(next_flag), C.byref(info))
    return info


def apply_coupling(mag, ang, vector_form=True):
    mag = np.array(mag, dtype=np.float32)
    ang = np.array(ang, dtype=np.float32)
    lib().orc_apply_coupling(_fp(mag), _fp(ang), mag.size, int(vector_form))
    return mag, ang


def floor0_apply(order, rate, bark_map_size, amp_bits, amp_ofs, coeff, amp, block_size, residue):
    """Floor0.Apply (Floor0.cs:164-225); returns the multiplied residue, raises where the reference throws."""
    f = Floor0(order, rate, bark_map_size, amp_bits, amp_ofs)
    c = np.array(coeff, dtype=np.float32)
    r = np.array(residue, dtype=np.float32)
    rc = lib().orc_floor0_apply(C.byref(f), _fp(c), float(amp), block_size, _fp(r))
    if rc != 0:
        raise IndexError("Floor0.Apply indexes its w map out of range (bark_map_size > blockSize/2)")
    return r


def floor0_bark_map(order, rate, bark_map_size, n):
    f = Floor0(order, rate, bark_map_size, 0, 0)
    m = np.zeros(n + 1, dtype=np.int32)
    lib().orc_floor0_bark_map(C.byref(f), n, m.ctypes.data_as(C.POINTER(C.c_int)))
    return m


def inverse_db_table():
    p = lib().orc_floor1_inverse_db_table()
    return np.ctypeslib.as_array(p, shape=(256,)).copy()


def floor1_init(xlist, multiplier):
    f = Floor1()
    xl = np.ascontiguousarray(xlist, dtype=np.int32)
    rc = lib().orc_floor1_init(C.byref(f), xl.ctypes.data_as(C.POINTER(C.c_int)), len(xl), multiplier)
    if rc != 0:
        raise ValueError("invalid floor1 configuration")
    return f


def floor1_unwrap(f, posts, post_count):
    p = np.zeros(64, dtype=np.int32)
    p[:len(posts)] = posts
    flags = np.zeros(64, dtype=np.uint8)
    lib().orc_floor1_unwrap_posts(C.byref(f), p.ctypes.data_as(C.POINTER(C.c_int)), post_count,
                                  flags.ctypes.data_as(C.POINTER(C.c_uint8)))
    return p, flags


def floor1_indices(f, posts, post_count, n):
    """UnwrapPosts + the render walk of Apply (Floor1.cs:222-268) as integers only: returns (finalY [64],
    step flags [64], table index per bin [n] -- unclamped, as the reference indexes its table)."""
    final_y, flags = floor1_unwrap(f, posts, post_count)
    out = np.zeros(n, dtype=np.int32)
    i32p, u8p = C.POINTER(C.c_int), C.POINTER(C.c_uint8)
    lib().orc_floor1_render_indices(C.byref(f), final_y.ctypes.data_as(i32p), flags.ctypes.data_as(u8p), post_count, n,
                                    out.ctypes.data_as(i32p))
    return final_y, flags, out


def floor1_apply(f, posts, post_count, block_size, residue):
    p = np.zeros(64, dtype=np.int32)
    p[:len(posts)] = posts
    r = np.array(residue, dtype=np.float32)
    lib().orc_floor1_apply(C.byref(f), p.ctypes.data_as(C.POINTER(C.c_int)), post_count, block_size, _fp(r))
    return r


def mapping_synth(channels, block_size, residue, floors, floor_of_channel, posts, post_count,
                  coupling, vector_form=True):
    """Mapping.cs:166-195.  residue [channels, block_size/2] -> pcm [channels, block_size]."""
    buf = np.zeros((channels, block_size), dtype=np.float32)
    buf[:, :block_size // 2] = residue
    farr = (Floor1 * len(floors))(*floors)
    foc = np.ascontiguousarray(floor_of_channel, dtype=np.int32)
    p = np.zeros((channels, 64), dtype=np.int32)
    posts = np.asarray(posts)
    p[:, :posts.shape[1]] = posts
    pc = np.ascontiguousarray(post_count, dtype=np.int32)
    mag = np.array([c[0] for c in coupling], dtype=np.uint8)
    ang = np.array([c[1] for c in coupling], dtype=np.uint8)
    i32p, u8p = C.POINTER(C.c_int), C.POINTER(C.c_uint8)
    lib().orc_mapping_synth(channels, block_size, _fp(buf), block_size, farr, foc.ctypes.data_as(i32p),
                            p.ctypes.data_as(i32p), pc.ctypes.data_as(i32p),
                            mag.ctypes.data_as(u8p), ang.ctypes.data_as(u8p), len(coupling), int(vector_form))
    return buf


def synth_stream_planar(channels, size0, size1, flags, spectra, clip=False):
    """StreamDecoder decode half over one stream of pre-decoded spectra.

    flags: uint8 [frames] (bit0 block flag, bit1 prev, bit2 next); spectra [frames, channels, size1/2].
    Returns pcm [channels, total].
    """
    flags = np.ascontiguousarray(flags, dtype=np.uint8)
    spectra = np.ascontiguousarray(spectra, dtype=np.float32)
    frames = len(flags)
    cap = frames * (size1 // 2) + 1
    pcm = np.zeros((channels, cap), dtype=np.float32)
    total = lib().orc_synth_stream_planar(channels, size0, size1, frames,
                                          flags.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(spectra),
                                          _fp(pcm), cap, int(clip))
    if total < 0:
        raise RuntimeError("inconsistent window flags (StreamDecoder.cs:777-778 would throw)")
    return pcm[:, :total].copy()


class FlooredStream:
    """The arguments of orc_synth_stream_floored for ONE stream's packets in the batch-array form the C ABI takes
    (packets: numpy record array with flags / mapping / granule / residue_offset; residue float32; posts int16 [records, 64];
    counts uint8 [records]; floors [(x_list, multiplier)]; mappings [{"coupling": [(mag, ang)], "channel_floor": [...]}]).
    run() pushes the packets through the restated Mapping.DecodePacket tail + StreamDecoder in C (Mdct tables warm) and
    returns the samples per channel; .pcm holds them (planar).  bench.py's CPU baseline of the fused workloads calls run()
    from several threads on separate instances (ctypes releases the GIL)."""

    def __init__(self, channels, size0, size1, packets, residue, posts, counts, floors=(), mappings=(), clip=False):
        self.channels, self.size0, self.size1, self.clip = channels, size0, size1, int(clip)
        self.n = len(packets)
        self.flags = np.ascontiguousarray(packets["flags"], dtype=np.uint8)
        self.mapping = np.ascontiguousarray(packets["mapping"], dtype=np.uint8)
        self.granule = np.ascontiguousarray(packets["granule"], dtype=np.int64)
        self.offsets = np.ascontiguousarray(packets["residue_offset"], dtype=np.int64)
        self.residue = np.ascontiguousarray(residue, dtype=np.float32)
        recs = max(1, self.n * channels)
        self.posts = np.zeros((recs, 64), dtype=np.int16) if posts is None else np.ascontiguousarray(posts, dtype=np.int16)
        self.counts = np.zeros(recs, dtype=np.uint8) if counts is None else np.ascontiguousarray(counts, dtype=np.uint8)
        fl = [floor1_init(x, m) for x, m in floors] or [floor1_init([0, 128], 1)]
        self.floors = (Floor1 * len(fl))(*fl)
        nm = max(1, len(mappings))
        self.foc = np.zeros(nm * channels, dtype=np.int32)
        mag, ang, off = [], [], [0]
        for i, m in enumerate(mappings):
            self.foc[i * channels:(i + 1) * channels] = m.get("channel_floor", [0] * channels)
            for a, b in m.get("coupling", []):
                mag.append(a)
                ang.append(b)
            off.append(len(mag))
        while len(off) < nm + 1:
            off.append(len(mag))
        self.mag = np.array(mag or [0], dtype=np.uint8)
        self.ang = np.array(ang or [0], dtype=np.uint8)
        self.off = np.array(off, dtype=np.int32)
        self.cap = self.n * (size1 // 2) + size1
        self.pcm = np.zeros((channels, self.cap), dtype=np.float32)
        self.total = 0

    def run(self):
        u8p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int)
        i64p, i16p = C.POINTER(C.c_int64), C.POINTER(C.c_int16)
        self.total = lib().orc_synth_stream_floored(
            self.channels, self.size0, self.size1, self.n, self.flags.ctypes.data_as(u8p), self.mapping.ctypes.data_as(u8p),
            self.granule.ctypes.data_as(i64p), self.offsets.ctypes.data_as(i64p), _fp(self.residue), self.floors,
            self.foc.ctypes.data_as(i32p), self.mag.ctypes.data_as(u8p), self.ang.ctypes.data_as(u8p),
            self.off.ctypes.data_as(i32p), self.posts.ctypes.data_as(i16p), self.counts.ctypes.data_as(u8p), _fp(self.pcm),
            self.cap, self.clip)
        return self.total
