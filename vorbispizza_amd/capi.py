"""ctypes binding of include/vorbispizza_synth.h -- the same entry points a C# host P/Invokes.

There is no CPU fallback: if the shared library is missing the import raises, and every compute
call raises SynthError when no MI355X is usable.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VPZ_LIB_DIR: another build of the two libraries, for A/B runs of two builds on one GPU box -- tools/ab_builds.sh)
LIB_PATH = os.path.join(os.environ.get("VPZ_LIB_DIR") or os.path.join(_HERE, "lib"), "libvorbispizza_synth.so")

OK = 0
E_INVALID_ARG, E_UNSUPPORTED, E_HIP, E_NOMEM, E_WINDOW_MISMATCH, E_NO_DEVICE, E_CAPACITY = -1, -2, -3, -4, -5, -6, -7
MEM_HOST, MEM_DEVICE = 0, 1
RESIDUE_F32, RESIDUE_I16 = 0, 1  # vpz_decoder_set_residue_format (ABI v5)
IMDCT_FAST, IMDCT_EXACT = 0, 1
OUT_INTERLEAVED, OUT_PLANAR, OUT_INTERLEAVED_S16, OUT_PLANAR_S16 = 0, 1, 2, 3
PKT_BLOCK_FLAG, PKT_PREV_FLAG, PKT_NEXT_FLAG, PKT_EOS = 0x01, 0x02, 0x04, 0x08
PKT_NOT_DECODED, PKT_INTERLEAVED, PKT_NO_FLOOR, PKT_RESYNC = 0x10, 0x20, 0x40, 0x80
MAX_FLOOR1_POSTS, POSTS_STRIDE, MAX_CHANNELS, MAX_COUPLING = 65, 64, 255, 256


class Floor1Config(C.Structure):
    _fields_ = [("x_count", C.c_int32), ("multiplier", C.c_int32), ("x_list", C.c_int32 * MAX_FLOOR1_POSTS)]


class Floor0Config(C.Structure):
    _fields_ = [("order", C.c_int32), ("rate", C.c_int32), ("bark_map_size", C.c_int32), ("amp_bits", C.c_int32),
                ("amp_ofs", C.c_int32)]


class MappingConfig(C.Structure):
    _fields_ = [("coupling_steps", C.c_int32),
                ("coupling_magnitude", C.c_uint8 * MAX_COUPLING),
                ("coupling_angle", C.c_uint8 * MAX_COUPLING),
                ("channel_floor", C.c_uint8 * (MAX_CHANNELS + 1)),
                ("residue_begin", C.c_int32 * 2), ("residue_end", C.c_int32 * 2)]


class StreamConfig(C.Structure):
    _fields_ = [("channels", C.c_int32), ("block_size0", C.c_int32), ("block_size1", C.c_int32),
                ("floor_count", C.c_int32), ("floors", C.POINTER(Floor1Config)),
                ("mapping_count", C.c_int32), ("mappings", C.POINTER(MappingConfig)),
                ("clip_samples", C.c_int32),
                ("floor_types", C.POINTER(C.c_uint8)), ("floors0", C.POINTER(Floor0Config))]


class Packet(C.Structure):
    _fields_ = [("stream", C.c_int32), ("flags", C.c_uint8), ("mapping", C.c_uint8),
                ("reserved", C.c_uint16), ("granule", C.c_int64), ("residue_offset", C.c_int64)]


PACKET_DTYPE = np.dtype([("stream", "<i4"), ("flags", "u1"), ("mapping", "u1"), ("reserved", "<u2"),
                         ("granule", "<i8"), ("residue_offset", "<i8")])
assert PACKET_DTYPE.itemsize == C.sizeof(Packet) == 24

# every symbol include/vorbispizza_synth.h declares: (name, restype, argtypes)
_vp = C.c_void_p
_SIGNATURES = [
    ("vpz_abi_version", C.c_int, []),
    ("vpz_error_string", C.c_char_p, [C.c_int]),
    ("vpz_device_count", C.c_int, []),
    ("vpz_context_create", C.c_int, [C.c_int, C.POINTER(_vp)]),
    ("vpz_context_destroy", None, [_vp]),
    ("vpz_context_synchronize", C.c_int, [_vp]),
    ("vpz_context_last_error", C.c_char_p, [_vp]),
    ("vpz_context_stream", _vp, [_vp]),
    ("vpz_context_timer_start", C.c_int, [_vp]),
    ("vpz_context_timer_stop", C.c_int, [_vp, C.POINTER(C.c_float)]),
    ("vpz_device_alloc", C.c_int, [_vp, C.c_uint64, C.POINTER(_vp)]),
    ("vpz_device_free", C.c_int, [_vp, _vp]),
    ("vpz_host_alloc", C.c_int, [_vp, C.c_uint64, C.POINTER(_vp)]),
    ("vpz_host_free", C.c_int, [_vp, _vp]),
    ("vpz_memcpy_h2d", C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    ("vpz_memcpy_d2h", C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    ("vpz_imdct_batch", C.c_int, [_vp, C.c_int, C.c_int64, _vp, _vp, C.c_int, C.c_int]),
    ("vpz_decoder_create", C.c_int, [_vp, C.POINTER(StreamConfig), C.c_int32, C.POINTER(_vp)]),
    ("vpz_decoder_destroy", None, [_vp]),
    ("vpz_decoder_reset", C.c_int, [_vp, C.c_int32]),
    ("vpz_decoder_synth", C.c_int, [_vp, C.c_int64, _vp, _vp, C.c_int64, _vp, _vp, C.c_int64, C.c_int, _vp, _vp,
                                    C.c_int64, C.c_int, C.c_int64, _vp]),
    ("vpz_decoder_last_packet_status", C.c_int, [_vp, _vp, C.c_int64, C.POINTER(C.c_int64)]),
    ("vpz_decoder_set_floor0_data", C.c_int, [_vp, _vp, _vp, C.c_int32]),
    ("vpz_decoder_last_packet_samples", C.c_int, [_vp, _vp, C.c_int64]),
    ("vpz_decoder_has_clipped", C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int32)]),
    ("vpz_decoder_position", C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int64)]),
    ("vpz_decoder_set_position", C.c_int, [_vp, C.c_int32, C.c_int64]),
    ("vpz_decoder_set_residue_format", C.c_int, [_vp, C.c_int32]),
    ("vpz_decoder_set_stream_capacities", C.c_int, [_vp, C.POINTER(C.c_int64), C.c_int32]),
    ("vpz_decoder_set_host_threads", C.c_int, [_vp, C.c_int32]),
]
# include/vorbispizza_synth_debug.h (test-only entry points, not part of the surface a C# host binds)
_DEBUG_SIGNATURES = [
    ("vpz_debug_floor1_indices", C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]
DEBUG_SYMBOLS = [s[0] for s in _DEBUG_SIGNATURES]

_lib = None


class SynthError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        name = lib().vpz_error_string(status).decode()
        super().__init__("vorbispizza_synth: %s (%d)%s" % (name, status, (": " + detail) if detail else ""))


def lib():
    """Load libvorbispizza_synth.so (built in-tree by vorbispizza_amd._build.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libvorbispizza_synth.so is not built: run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64,
        # and a second copy initialised later finds no device.  Importing torch first makes the
        # loader bind this library's NEEDED libamdhip64.so.7 to the copy torch already loaded.
        if not os.environ.get("VPZ_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, restype, argtypes in _SIGNATURES + _DEBUG_SIGNATURES:
            fn = getattr(L, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _sync_producer(*tensors):
    """The C ABI orders work only on the context's own HIP stream.  Tensors produced on torch's
    current stream must be complete before the library reads them, so wait for that stream here."""
    for t in tensors:
        if t is not None and _is_torch(t) and t.is_cuda:
            import torch
            torch.cuda.current_stream(t.device).synchronize()
            return


def _numel(x):
    if x is None:
        return 0
    return int(x.numel()) if _is_torch(x) else int(np.asarray(x).size)


def _ptr(x):
    if x is None:
        return None
    if _is_torch(x):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x.ctypes.data)


class Context:
    """vpz_context: one GPU, one HIP stream, the per-block-size tables."""

    def __init__(self, device=0):
        self._h = _vp()
        rc = lib().vpz_context_create(device, C.byref(self._h))
        if rc != OK:
            self._h = None
            raise SynthError(rc, "vpz_context_create(device=%d)" % device)
        self.device = device
        self._children = []  # weak references to decoders: they must be destroyed before the context

    def _check(self, rc):
        if rc != OK:
            raise SynthError(rc, lib().vpz_context_last_error(self._h).decode())

    def close(self):
        if self._h:
            for ref in self._children:
                child = ref()
                if child is not None:
                    child.close()
            self._children = []
            lib().vpz_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        """Text of the last failure on this context (what the C# host would put into its exception)."""
        return lib().vpz_context_last_error(self._h).decode()

    def synchronize(self):
        self._check(lib().vpz_context_synchronize(self._h))

    @property
    def stream(self):
        return lib().vpz_context_stream(self._h)

    def timer_start(self):
        self._check(lib().vpz_context_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        self._check(lib().vpz_context_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def imdct_batch(self, spectra, n, mode=IMDCT_FAST, out=None):
        """`Mdct.Reverse` (Mdct.cs:15-19) per row.  numpy in -> numpy out (host memory, synchronous);
        torch cuda tensor in -> torch cuda tensor out (device memory, asynchronous on the context stream)."""
        half = max(n // 2, 1)
        if _is_torch(spectra):
            import torch
            assert spectra.is_cuda and spectra.dtype == torch.float32 and spectra.is_contiguous()
            count = spectra.numel() // half
            if out is None:
                out = torch.empty((count, n), dtype=torch.float32, device=spectra.device)
            _sync_producer(spectra)
            self._check(lib().vpz_imdct_batch(self._h, n, count, _ptr(spectra), _ptr(out), MEM_DEVICE, mode))
            return out
        spectra = np.ascontiguousarray(spectra, dtype=np.float32).reshape(-1, half)
        count = spectra.shape[0]
        if out is None:
            out = np.empty((count, n), dtype=np.float32)
        self._check(lib().vpz_imdct_batch(self._h, n, count, _ptr(spectra), _ptr(out), MEM_HOST, mode))
        return out


def make_packets(count):
    return np.zeros(count, dtype=PACKET_DTYPE)


class Decoder:
    """vpz_decoder: synthesis state of `n_streams` streams that share one setup header."""

    def __init__(self, ctx, channels, block_size0, block_size1, floors=(), mappings=(), n_streams=1,
                 clip_samples=False):
        """floors: [(x_list, multiplier)] for a type-1 floor or {"order":, "rate":, "bark_map_size":, "amp_bits":,
        "amp_ofs":} for a type-0 floor; mappings: [{"coupling": [(mag, ang)], "channel_floor": [..],
        "residue_begin": (short, long), "residue_end": (short, long)}] -- the last two optional (ABI v4: the residue's
        support in bins per channel; 0 / absent = the whole block)"""
        self.ctx = ctx
        self.channels, self.size0, self.size1, self.n_streams = channels, block_size0, block_size1, n_streams
        fl = (Floor1Config * max(1, len(floors)))()
        ftypes = (C.c_uint8 * max(1, len(floors)))()
        fl0 = (Floor0Config * max(1, len(floors)))()
        for i, f in enumerate(floors):
            if isinstance(f, dict):
                ftypes[i] = 0
                fl0[i] = Floor0Config(f["order"], f["rate"], f["bark_map_size"], f["amp_bits"], f["amp_ofs"])
                continue
            ftypes[i] = 1
            xl, mult = f
            if len(xl) > MAX_FLOOR1_POSTS:
                raise SynthError(E_INVALID_ARG, "floor1 X list too long")
            fl[i].x_count = len(xl)
            fl[i].multiplier = mult
            for j, x in enumerate(xl):
                fl[i].x_list[j] = int(x)
        mp = (MappingConfig * max(1, len(mappings)))()
        for i, m in enumerate(mappings):
            cp = m.get("coupling", [])
            mp[i].coupling_steps = len(cp)
            for j, (mag, ang) in enumerate(cp):
                mp[i].coupling_magnitude[j] = mag
                mp[i].coupling_angle[j] = ang
            for c, f in enumerate(m.get("channel_floor", [0] * channels)):
                mp[i].channel_floor[c] = f
            for b in range(2):
                mp[i].residue_begin[b] = int(m.get("residue_begin", (0, 0))[b])
                mp[i].residue_end[b] = int(m.get("residue_end", (0, 0))[b])
        cfg = StreamConfig(channels, block_size0, block_size1, len(floors), fl, len(mappings), mp,
                           1 if clip_samples else 0, ftypes, fl0)
        self._h = _vp()
        rc = lib().vpz_decoder_create(ctx._h, C.byref(cfg), n_streams, C.byref(self._h))
        if rc != OK:
            self._h = None
            raise SynthError(rc, lib().vpz_context_last_error(ctx._h).decode())
        import weakref
        ctx._children.append(weakref.ref(self))

    def close(self):
        if self._h and self.ctx._h:
            lib().vpz_decoder_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, stream=-1):
        self.ctx._check(lib().vpz_decoder_reset(self._h, stream))

    def synth_raw(self, packets, residue, posts, post_counts, pcm_out, stream_out_offset, capacity,
                  out_layout, channel_stride, mem_space, residue_floats=None, n_records=None, on_mismatch="raise"):
        """Thin call of vpz_decoder_synth; returns samples_written (int64 array, per stream).
        residue_floats / n_records default to the extents of the arrays handed in.  A packet skipped by the window check
        (StreamDecoder.cs:777-778) is a per-packet status in the ABI; on_mismatch="raise" turns it into
        SynthError(E_WINDOW_MISMATCH) AFTER the call has done all its work -- what a host does for the stream concerned --,
        "ignore" leaves it to last_packet_status()."""
        written = np.zeros(self.n_streams, dtype=np.int64)
        packets = np.ascontiguousarray(packets, dtype=PACKET_DTYPE)
        # ABI v5: the element type of `residue` is the array's (numpy or torch int16: VPZ_RESIDUE_I16, anything else float32)
        want = RESIDUE_I16 if str(getattr(residue, "dtype", "")) in ("int16", "torch.int16") else RESIDUE_F32
        if getattr(self, "residue_format", RESIDUE_F32) != want:
            self.set_residue_format(want)
        offs = None if stream_out_offset is None else np.ascontiguousarray(stream_out_offset, dtype=np.int64)
        if mem_space == MEM_DEVICE:
            _sync_producer(residue, posts, post_counts, pcm_out)
        if residue_floats is None:
            residue_floats = _numel(residue)
        if n_records is None:
            n_records = 0 if post_counts is None else _numel(post_counts)
        rc = lib().vpz_decoder_synth(self._h, len(packets), _ptr(packets), _ptr(residue), residue_floats, _ptr(posts),
                                     _ptr(post_counts), n_records, mem_space, _ptr(pcm_out), _ptr(offs), capacity,
                                     out_layout, channel_stride, _ptr(written))
        self.ctx._check(rc)
        self.last_written = written
        if on_mismatch == "raise" and self.last_mismatches() > 0:
            raise SynthError(E_WINDOW_MISMATCH, "a packet's previous tail is longer than its window slope "
                             "(StreamDecoder.cs:777-778 throws); the packet was skipped, everything else was synthesised")
        return written

    def last_mismatches(self):
        n = C.c_int64()
        self.ctx._check(lib().vpz_decoder_last_packet_status(self._h, None, 0, C.byref(n)))
        return n.value

    def last_packet_status(self, n_packets):
        out = np.zeros(n_packets, dtype=np.int32)
        self.ctx._check(lib().vpz_decoder_last_packet_status(self._h, _ptr(out), n_packets, None))
        return out

    def synth(self, packets, residue, posts=None, post_counts=None, out_layout=OUT_PLANAR, capacity=None,
              on_mismatch="raise"):
        """Host-memory convenience: returns a list (per stream) of PCM arrays, [channels, samples]
        for OUT_PLANAR or [samples, channels] for OUT_INTERLEAVED (int16 arrays for the _S16 layouts)."""
        packets = np.ascontiguousarray(packets, dtype=PACKET_DTYPE)
        # (int16 values -- ABI v5 -- go over as they are: synth_raw tells the decoder)
        residue = np.ascontiguousarray(residue, dtype=np.int16 if getattr(residue, "dtype", None) == np.int16 else np.float32)
        if posts is not None:
            posts = np.ascontiguousarray(posts, dtype=np.int16)
            post_counts = np.ascontiguousarray(post_counts, dtype=np.uint8)
        if capacity is None:
            per = np.zeros(self.n_streams, dtype=np.int64)
            for s, f in zip(packets["stream"], packets["flags"]):
                per[s] += (self.size1 if f & PKT_BLOCK_FLAG else self.size0)
            capacity = int(per.max()) + 1
        C_ = self.channels
        s16 = out_layout in (OUT_INTERLEAVED_S16, OUT_PLANAR_S16)
        out = np.zeros(self.n_streams * C_ * capacity, dtype=np.int16 if s16 else np.float32)
        offs = np.arange(self.n_streams, dtype=np.int64) * (C_ * capacity)
        written = self.synth_raw(packets, residue, posts, post_counts, out, offs, capacity, out_layout,
                                 capacity, MEM_HOST, on_mismatch=on_mismatch)
        res = []
        for s in range(self.n_streams):
            blk = out[offs[s]: offs[s] + C_ * capacity]
            if out_layout in (OUT_PLANAR, OUT_PLANAR_S16):
                res.append(blk.reshape(C_, capacity)[:, :written[s]].copy())
            else:
                res.append(blk[: written[s] * C_].reshape(written[s], C_).copy())
        return res

    def debug_floor1_indices(self, posts, post_counts, record_floor, record_long):
        """Test-only (vorbispizza_synth_debug.h): the integers of the Floor1 device path for a batch of channel
        records.  Returns (curve [records, size1/2] uint8, final_y [records, 64] int16, step_flags [records, 64]
        uint8, active_count [records] uint8)."""
        posts = np.ascontiguousarray(posts, dtype=np.int16).reshape(-1, 64)
        n = posts.shape[0]
        post_counts = np.ascontiguousarray(post_counts, dtype=np.uint8)
        record_floor = np.ascontiguousarray(record_floor, dtype=np.uint8)
        record_long = np.ascontiguousarray(record_long, dtype=np.uint8)
        assert len(post_counts) == len(record_floor) == len(record_long) == n
        curve = np.full((n, self.size1 // 2), 0xEE, dtype=np.uint8)
        final_y = np.zeros((n, 64), dtype=np.int16)
        flags = np.zeros((n, 64), dtype=np.uint8)
        active = np.zeros(n, dtype=np.uint8)
        self.ctx._check(lib().vpz_debug_floor1_indices(self._h, n, _ptr(posts), _ptr(post_counts), _ptr(record_floor),
                                                       _ptr(record_long), _ptr(curve), _ptr(final_y), _ptr(flags),
                                                       _ptr(active)))
        return curve, final_y, flags, active

    def set_floor0_data(self, amp, coeff):
        """amp [records], coeff [records, stride] (numpy, host memory) for the next synth call."""
        self._f0 = (np.ascontiguousarray(amp, dtype=np.float32), np.ascontiguousarray(coeff, dtype=np.float32))
        self.ctx._check(lib().vpz_decoder_set_floor0_data(self._h, _ptr(self._f0[0]), _ptr(self._f0[1]),
                                                          self._f0[1].shape[1]))

    def last_packet_samples(self, n_packets):
        out = np.zeros(n_packets, dtype=np.int32)
        self.ctx._check(lib().vpz_decoder_last_packet_samples(self._h, _ptr(out), n_packets))
        return out

    def has_clipped(self, stream=0):
        v = C.c_int32()
        self.ctx._check(lib().vpz_decoder_has_clipped(self._h, stream, C.byref(v)))
        return bool(v.value)

    def position(self, stream=0):
        v = C.c_int64()
        self.ctx._check(lib().vpz_decoder_position(self._h, stream, C.byref(v)))
        return v.value

    def set_residue_format(self, fmt):
        """RESIDUE_F32 (default) or RESIDUE_I16: the element type of `residue` in the synth calls that follow (ABI v5)"""
        self.ctx._check(lib().vpz_decoder_set_residue_format(self._h, int(fmt)))
        self.residue_format = int(fmt)

    def set_host_threads(self, n):
        """Host threads this decoder's synth calls may use for their integer half (0: the process's CPUs, 1: none but the caller's)."""
        self.ctx._check(lib().vpz_decoder_set_host_threads(self._h, int(n)))

    def set_stream_capacities(self, capacities):
        """per-stream output bounds for the synth calls that follow (each tightens the call's stream_out_capacity); None removes them"""
        if capacities is None:
            self.ctx._check(lib().vpz_decoder_set_stream_capacities(self._h, None, 0))
            return
        caps = np.ascontiguousarray(capacities, dtype=np.int64)
        self.ctx._check(lib().vpz_decoder_set_stream_capacities(self._h, caps.ctypes.data_as(C.POINTER(C.c_int64)), int(caps.size)))

    def set_position(self, position, stream=0):
        self.ctx._check(lib().vpz_decoder_set_position(self._h, stream, int(position)))
