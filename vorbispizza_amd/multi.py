"""ctypes view of include/vorbispizza_multi.h -- the in-process multi-device dispatcher of libvorbispizza_host.so (one host
process, one context group per MI355X, streams partitioned contiguously, no collective).  What a C# host P/Invokes;
tests and bench.py use it from here."""
import ctypes as C

import numpy as np

from . import capi, front

OK, E_ARG, E_DEVICE, E_NOMEM = 0, -1, -2, -3
E_OPEN, E_CAPACITY, E_SYNTH, E_SETUP = -10, -11, -12, -13


class Options(C.Structure):
    _fields_ = [("host_threads", C.c_int32), ("streams_per_call", C.c_int32), ("contexts_per_device", C.c_int32),
                ("clip_samples", C.c_int32), ("slots_per_device", C.c_int32), ("float_residue", C.c_int32),
                ("reserved", C.c_int32 * 2)]


class StreamResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("device_slot", C.c_int32), ("channels", C.c_int32), ("sample_rate", C.c_int32),
                ("samples", C.c_int64), ("packets", C.c_int64), ("skipped_packets", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [("wall_s", C.c_double), ("device_wall_s", C.c_double * 16), ("device_decode_s", C.c_double * 16),
                ("device_synth_s", C.c_double * 16), ("device_streams", C.c_int64 * 16), ("device_samples", C.c_int64 * 16),
                ("threads_per_device", C.c_int32), ("pinned_mib", C.c_int32)]


RESULT_DTYPE = np.dtype([("status", "<i4"), ("device_slot", "<i4"), ("channels", "<i4"), ("sample_rate", "<i4"),
                         ("samples", "<i8"), ("packets", "<i8"), ("skipped_packets", "<i8")])
assert RESULT_DTYPE.itemsize == C.sizeof(StreamResult)

EXPORTED_SYMBOLS = ["vpzm_create", "vpzm_destroy", "vpzm_last_error", "vpzm_device_count", "vpzm_decode_library"]
_bound = False


def lib():
    global _bound
    L = front.lib()
    if not _bound:
        vp = C.c_void_p
        L.vpzm_create.argtypes = [vp, C.c_int32, C.POINTER(Options), C.POINTER(vp)]
        L.vpzm_create.restype = C.c_int
        L.vpzm_destroy.argtypes = [vp]
        L.vpzm_destroy.restype = None
        L.vpzm_last_error.argtypes = [vp]
        L.vpzm_last_error.restype = C.c_char_p
        L.vpzm_device_count.argtypes = [vp]
        L.vpzm_device_count.restype = C.c_int
        L.vpzm_decode_library.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, vp, vp, vp, vp, C.POINTER(Stats)]
        L.vpzm_decode_library.restype = C.c_int
        _bound = True
    return L


class MultiError(RuntimeError):
    pass


class Dispatcher:
    """vpzm_dispatcher: one context group per entry of `device_ids` (an id may repeat: several groups on one GPU)."""

    def __init__(self, device_ids, host_threads=0, streams_per_call=0, contexts_per_device=0, clip_samples=False,
                 slots_per_device=0, float_residue=False):
        ids = (C.c_int32 * len(device_ids))(*[int(d) for d in device_ids])
        opt = Options(host_threads, streams_per_call, contexts_per_device, 1 if clip_samples else 0, slots_per_device,
                      1 if float_residue else 0)
        self._h = C.c_void_p()
        rc = lib().vpzm_create(ids, len(device_ids), C.byref(opt), C.byref(self._h))
        if rc != OK:
            self._h = None
            raise MultiError("vpzm_create failed (status %d)" % rc)
        self.n_devices = len(device_ids)

    def close(self):
        if getattr(self, "_h", None):
            lib().vpzm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return lib().vpzm_last_error(self._h).decode()

    def decode_library(self, datas, pcm_out, pcm_offset, pcm_capacity, s16=False):
        """datas: list of numpy uint8 arrays (containers); pcm_out: numpy float32 / int16 array (or a torch pinned tensor's
        .numpy()); pcm_offset / pcm_capacity: per stream, elements / samples per channel.  Returns (results, stats):
        a numpy record array (RESULT_DTYPE) and the Stats struct."""
        n = len(datas)
        ptrs = (C.c_void_p * n)(*[d.ctypes.data for d in datas])
        sizes = (C.c_uint64 * n)(*[d.size for d in datas])
        offs = np.ascontiguousarray(pcm_offset, dtype=np.int64)
        caps = np.ascontiguousarray(pcm_capacity, dtype=np.int64)
        assert pcm_out.dtype == (np.int16 if s16 else np.float32) and pcm_out.flags["C_CONTIGUOUS"]
        results = np.zeros(n, dtype=RESULT_DTYPE)
        stats = Stats()
        rc = lib().vpzm_decode_library(self._h, n, ptrs, sizes, capi.OUT_INTERLEAVED_S16 if s16 else capi.OUT_INTERLEAVED,
                                       pcm_out.ctypes.data, offs.ctypes.data, caps.ctypes.data, results.ctypes.data,
                                       C.byref(stats))
        if rc != OK:
            raise MultiError("vpzm_decode_library failed (status %d): %s" % (rc, self.last_error()))
        return results, stats
