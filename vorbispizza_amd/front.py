"""ctypes view of the C++ CPU front end (vorbispizza_amd/host): Ogg demux, Vorbis setup headers and the
per-packet entropy decode -- the stage the reference host keeps on the CPU.  It yields exactly the arrays
vpz_decoder_synth consumes."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VPZ_LIB_DIR: another build of the two libraries, for A/B runs of two builds on one GPU box -- tools/ab_builds.sh)
LIB_PATH = os.path.join(os.environ.get("VPZ_LIB_DIR") or os.path.join(_HERE, "lib"), "libvorbispizza_host.so")
_lib = None


class Info(C.Structure):
    _fields_ = [("channels", C.c_int32), ("sample_rate", C.c_int32), ("block_size0", C.c_int32),
                ("block_size1", C.c_int32), ("floor_count", C.c_int32), ("residue_count", C.c_int32),
                ("mapping_count", C.c_int32), ("mode_count", C.c_int32), ("codebook_count", C.c_int32),
                ("audio_packets", C.c_int64), ("last_granule", C.c_int64), ("residue_floats", C.c_int64),
                ("pages", C.c_int32), ("bad_crc_pages", C.c_int32), ("stream_serial", C.c_int32), ("reserved", C.c_int32)]


class FrontError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            from . import _build
            _build.build()
        capi.lib()  # the host library links against libvorbispizza_synth.so (and needs torch's HIP first)
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.vpzh_open_memory.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
        L.vpzh_open_memory.restype = C.c_int
        L.vpzh_open_memory_stream.argtypes = [vp, C.c_uint64, C.c_int32, C.POINTER(vp)]
        L.vpzh_open_memory_stream.restype = C.c_int
        L.vpzh_close.argtypes = [vp]
        L.vpzh_close.restype = None
        L.vpzh_last_error.argtypes = [vp]
        L.vpzh_last_error.restype = C.c_char_p
        L.vpzh_get_info.argtypes = [vp, C.POINTER(Info)]
        L.vpzh_get_info.restype = C.c_int
        L.vpzh_get_floor1.argtypes = [vp, C.c_int, C.POINTER(capi.Floor1Config)]
        L.vpzh_get_floor1.restype = C.c_int
        L.vpzh_get_floor_type.argtypes = [vp, C.c_int]
        L.vpzh_get_floor_type.restype = C.c_int
        L.vpzh_get_floor0.argtypes = [vp, C.c_int, C.POINTER(capi.Floor0Config)]
        L.vpzh_get_floor0.restype = C.c_int
        L.vpzh_max_floor0_order.argtypes = [vp]
        L.vpzh_max_floor0_order.restype = C.c_int
        L.vpzh_decode_range_ex.argtypes = [vp, C.c_int64, C.c_int64, C.c_int32, C.c_int64, vp, vp, vp, vp,
                                           C.POINTER(C.c_int64), vp, vp, C.c_int32]
        L.vpzh_decode_range_ex.restype = C.c_int
        L.vpzh_decode_range_i16.argtypes = L.vpzh_decode_range_ex.argtypes
        L.vpzh_decode_range_i16.restype = C.c_int
        L.vpzh_residue_is_integral.argtypes = [vp]
        L.vpzh_residue_is_integral.restype = C.c_int
        L.vpzh_get_mapping.argtypes = [vp, C.c_int, C.POINTER(capi.MappingConfig)]
        L.vpzh_get_mapping.restype = C.c_int
        L.vpzh_get_residue_type.argtypes = [vp, C.c_int]
        L.vpzh_get_residue_type.restype = C.c_int
        L.vpzh_decode_all.argtypes = [vp, C.c_int32, C.c_int64, vp, vp, vp, vp]
        L.vpzh_decode_all.restype = C.c_int
        L.vpzh_decode_range.argtypes = [vp, C.c_int64, C.c_int64, C.c_int32, C.c_int64, vp, vp, vp, vp,
                                        C.POINTER(C.c_int64)]
        L.vpzh_decode_range.restype = C.c_int
        L.vpzh_decode_many.argtypes = [C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp,
                                       C.POINTER(C.c_int64)]
        L.vpzh_decode_many.restype = C.c_int
        L.vpzh_decode_many_progress.argtypes = [C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_int64, vp, vp, vp,
                                                vp, C.POINTER(C.c_int64), vp]
        L.vpzh_decode_many_progress.restype = C.c_int
        L.vpzh_default_threads.argtypes = []
        L.vpzh_default_threads.restype = C.c_int
        L.vpzh_decode_failures.argtypes = [vp, C.POINTER(C.c_int64)]
        L.vpzh_decode_failures.restype = C.c_int64
        # VorbisReader mirror (include/vorbispizza_reader.h)
        L.vpzr_open_memory.argtypes = [vp, vp, C.c_uint64, C.POINTER(vp)]
        L.vpzr_open_memory.restype = C.c_int
        L.vpzr_close.argtypes = [vp]
        L.vpzr_close.restype = None
        L.vpzr_last_error.argtypes = [vp]
        L.vpzr_last_error.restype = C.c_char_p
        L.vpzr_switch_streams.argtypes = [vp, C.c_int]
        L.vpzr_switch_streams.restype = C.c_int
        for name in ("vpzr_channels", "vpzr_sample_rate", "vpzr_is_end_of_stream", "vpzr_has_clipped",
                     "vpzr_find_next_stream", "vpzr_stream_count", "vpzr_stream_serial"):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = C.c_int
        L.vpzr_sample_position.argtypes = [vp]
        L.vpzr_sample_position.restype = C.c_int64
        L.vpzr_set_clip_samples.argtypes = [vp, C.c_int]
        L.vpzr_set_clip_samples.restype = C.c_int
        L.vpzr_set_batch_packets.argtypes = [vp, C.c_int]
        L.vpzr_set_batch_packets.restype = C.c_int
        L.vpzr_seek_to.argtypes = [vp, C.c_int64, C.c_int]
        L.vpzr_seek_to.restype = C.c_int
        L.vpzr_total_samples.argtypes = [vp]
        L.vpzr_total_samples.restype = C.c_int64
        L.vpzh_total_samples.argtypes = [vp]
        L.vpzh_total_samples.restype = C.c_int64
        L.vpzh_seek.argtypes = [vp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.vpzh_seek.restype = C.c_int
        L.vpzr_read_samples.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int)]
        L.vpzr_read_samples.restype = C.c_int64
        L.vpzr_read_samples_planar.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int)]
        L.vpzr_read_samples_planar.restype = C.c_int64
        L.vpzr_read_samples_s16.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int)]
        L.vpzr_read_samples_s16.restype = C.c_int64
        L.vpzr_set_sample_format.argtypes = [vp, C.c_int]
        L.vpzr_set_sample_format.restype = C.c_int
        _lib = L
    return _lib


class OggVorbisFile:
    """One logical Vorbis stream of an .ogg file, entropy-decoded on the CPU.

    Attributes mirror what StreamDecoder holds after LoadStreamHeader / LoadBooks
    (StreamDecoder.cs:213-321): channels, sample_rate, block sizes, floors [(x_list, multiplier)],
    mappings [{"coupling": [(mag, ang)], "channel_floor": [...], "residue_begin": (short, long), "residue_end": (short, long)}]
    (the last two: the residue's support per block size, ABI v4)."""

    def __init__(self, path_or_bytes, stream_index=0):
        data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
        self._data = np.frombuffer(bytes(data), dtype=np.uint8)
        self._h = C.c_void_p()
        rc = lib().vpzh_open_memory_stream(self._data.ctypes.data, self._data.size, stream_index, C.byref(self._h))
        if rc != 0:
            msg = lib().vpzh_last_error(self._h).decode() if self._h else "open failed"
            self.close()
            raise FrontError("%s (status %d)" % (msg, rc))
        info = Info()
        lib().vpzh_get_info(self._h, C.byref(info))
        self.info = info
        self.channels, self.sample_rate = info.channels, info.sample_rate
        self.block_size0, self.block_size1 = info.block_size0, info.block_size1
        self.audio_packets, self.last_granule = info.audio_packets, info.last_granule
        self.floors, self.mappings = [], []
        for i in range(info.floor_count):
            if lib().vpzh_get_floor_type(self._h, i) == 0:
                f0 = capi.Floor0Config()
                lib().vpzh_get_floor0(self._h, i, C.byref(f0))
                self.floors.append({"order": f0.order, "rate": f0.rate, "bark_map_size": f0.bark_map_size,
                                    "amp_bits": f0.amp_bits, "amp_ofs": f0.amp_ofs})
                continue
            f = capi.Floor1Config()
            if lib().vpzh_get_floor1(self._h, i, C.byref(f)) != 0:
                raise FrontError("floor %d cannot be represented" % i)
            self.floors.append((list(f.x_list[: f.x_count]), f.multiplier))
        self.floor0_stride = lib().vpzh_max_floor0_order(self._h)
        self.floor0_data = None  # (amp [records], coeff [records, stride]) of the last decode_packets call
        for i in range(info.mapping_count):
            m = capi.MappingConfig()
            lib().vpzh_get_mapping(self._h, i, C.byref(m))
            self.mappings.append({
                "coupling": [(m.coupling_magnitude[j], m.coupling_angle[j]) for j in range(m.coupling_steps)],
                "channel_floor": list(m.channel_floor[: self.channels]),
                "residue_begin": tuple(m.residue_begin), "residue_end": tuple(m.residue_end)})
        self.residue_types = [lib().vpzh_get_residue_type(self._h, i) for i in range(info.residue_count)]

    @property
    def total_samples(self):
        """PacketProvider.GetGranuleCount: counted samples per channel, capped by the last page granule."""
        return lib().vpzh_total_samples(self._h)

    def seek(self, sample_position):
        """PacketProvider.SeekTo(pos, preRoll=1): (index of the pre-roll packet, samples of the packet after it
        that precede the position)."""
        first, roll = C.c_int64(), C.c_int64()
        if lib().vpzh_seek(self._h, int(sample_position), C.byref(first), C.byref(roll)) != 0:
            raise FrontError(lib().vpzh_last_error(self._h).decode())
        return first.value, roll.value

    def close(self):
        if getattr(self, "_h", None):
            lib().vpzh_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode_failures(self):
        """(count, index of the first) packets of the last decode call whose entropy decode threw the way the
        reference's DecodeNextPacket does; they were handed over as not-decoded packets."""
        first = C.c_int64(-1)
        n = lib().vpzh_decode_failures(self._h, C.byref(first))
        return int(n), int(first.value)

    def last_error(self):
        return lib().vpzh_last_error(self._h).decode()

    @property
    def residue_is_integral(self):
        """every residue value of the stream is an integer of 16 bits (decided from the setup header; vorbispizza_front.h)"""
        return bool(lib().vpzh_residue_is_integral(self._h))

    def decode_packets(self, stream_id=0, residue_base=0, int16=False):
        """Entropy-decode every audio packet.  Returns (packets, residue, posts, post_counts) in the
        layout of vpz_decoder_synth; int16=True: the residue as 16-bit integers (vpzh_decode_range_i16; only for a stream whose
        residue_is_integral)."""
        n, C_ = self.audio_packets, self.channels
        packets = capi.make_packets(n)
        residue = np.zeros(max(1, self.info.residue_floats), dtype=np.int16 if int16 else np.float32)
        posts = np.zeros((n * C_, 64), dtype=np.int16)
        counts = np.zeros(n * C_, dtype=np.uint8)
        amp = coeff = None
        if self.floor0_stride > 0:
            amp = np.zeros(n * C_, dtype=np.float32)
            coeff = np.zeros((n * C_, self.floor0_stride), dtype=np.float32)
        fn = lib().vpzh_decode_range_i16 if int16 else lib().vpzh_decode_range_ex
        rc = fn(self._h, 0, n, stream_id, residue_base, packets.ctypes.data, residue.ctypes.data,
                                        posts.ctypes.data, counts.ctypes.data, None,
                                        None if amp is None else amp.ctypes.data,
                                        None if coeff is None else coeff.ctypes.data, self.floor0_stride)
        self.floor0_data = None if amp is None else (amp, coeff)
        if rc != 0:
            raise FrontError(lib().vpzh_last_error(self._h).decode())
        return packets, residue[: self.info.residue_floats], posts, counts

    def decode_into(self, packets, residue, posts, counts, stream_id=0, residue_base=0):
        """decode_packets into caller-owned arrays (typically slices of one batch buffer shared by many
        streams): packets [audio_packets] of capi.PACKET_DTYPE, residue float32 [>= info.residue_floats]
        that starts at float index `residue_base` of the batch buffer, posts int16 [audio_packets * channels,
        64], counts uint8 [audio_packets * channels].  Distinct handles may be decoded from distinct threads
        (the call releases the GIL); type-0 floors are not supported here."""
        n = self.audio_packets
        assert self.floor0_stride == 0
        assert len(packets) == n and residue.size >= self.info.residue_floats and len(counts) == n * self.channels
        assert all(a.flags["C_CONTIGUOUS"] for a in (packets, residue, posts, counts))
        rc = lib().vpzh_decode_range_ex(self._h, 0, n, stream_id, residue_base, packets.ctypes.data, residue.ctypes.data,
                                        posts.ctypes.data, counts.ctypes.data, None, None, None, 0)
        if rc != 0:
            raise FrontError(lib().vpzh_last_error(self._h).decode())


def decode_many(datas, packet_base, residue_base, packets, residue, posts, counts, threads=0, stream_id0=0, residue_origin=0,
                packet_room=None, residue_room=None, done=None, channels=None):
    """vpzh_decode_many: opens and entropy-decodes the containers `datas` (numpy uint8 arrays) on `threads` host threads of
    the library's own (no Python in the loop, the GIL is released for the whole call), stream k into packets[packet_base[k]:],
    residue[residue_base[k]:], posts / counts at record packet_base[k] * channels.  packet_room / residue_room: what each slice
    holds (default: up to the next stream's base, or the end of the array).  done (optional, zeroed int32 array of len(datas)):
    vpzh_decode_many_progress -- entry k turns 1 (-1: refused) when stream k is complete, so that another thread can hand
    finished streams on while this call is still running.  channels: what the batch arrays are laid out for (default: what
    the shape of `posts` says); a container with another count is refused.  Returns the number of packets that failed."""
    n = len(datas)
    if channels is None:
        channels = max(1, posts.size // (64 * max(1, len(packets))))
    ptrs = (C.c_void_p * n)(*[d.ctypes.data for d in datas])
    sizes = (C.c_uint64 * n)(*[d.size for d in datas])
    pb = np.ascontiguousarray(packet_base, dtype=np.int64)
    rb = np.ascontiguousarray(residue_base, dtype=np.int64)

    def room(bases, total, given):
        if given is not None:
            return np.ascontiguousarray(given, dtype=np.int64)
        order = np.argsort(bases, kind="stable")
        ends = np.empty(n, dtype=np.int64)
        ends[order] = np.concatenate([bases[order][1:], [total]])
        return np.ascontiguousarray(ends - bases, dtype=np.int64)

    pr, rr = room(pb, len(packets), packet_room), room(rb, residue.size, residue_room)
    failed = C.c_int64(0)
    if done is not None:
        assert done.dtype == np.int32 and done.size >= n and done.flags["C_CONTIGUOUS"]
        rc = lib().vpzh_decode_many_progress(n, int(channels), ptrs, sizes, int(threads), int(stream_id0), pb.ctypes.data, pr.ctypes.data,
                                             rb.ctypes.data, rr.ctypes.data, int(residue_origin), packets.ctypes.data,
                                             residue.ctypes.data, posts.ctypes.data, counts.ctypes.data, C.byref(failed),
                                             done.ctypes.data)
    else:
        rc = lib().vpzh_decode_many(n, int(channels), ptrs, sizes, int(threads), int(stream_id0), pb.ctypes.data, pr.ctypes.data, rb.ctypes.data,
                                    rr.ctypes.data, int(residue_origin), packets.ctypes.data, residue.ctypes.data,
                                    posts.ctypes.data, counts.ctypes.data, C.byref(failed))
    if rc != 0:
        raise FrontError("vpzh_decode_many failed (status %d)" % rc)
    return failed.value


class VorbisReader:
    """Mirror of NVorbis.VorbisReader's read surface (VorbisReader.cs:232-253) on top of the GPU back
    end: ReadSamples(buffer) interleaved / ReadSamples(buffer, samplesToRead, channelStride) planar,
    at most one packet's worth per call, 0 at the end."""

    def __init__(self, ctx, path_or_bytes, clip_samples=True, batch_packets=128, s16=False):
        data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
        self._data = np.frombuffer(bytes(data), dtype=np.uint8)
        self._ctx = ctx
        self._h = C.c_void_p()
        rc = lib().vpzr_open_memory(ctx._h, self._data.ctypes.data, self._data.size, C.byref(self._h))
        if rc != 0:
            msg = lib().vpzr_last_error(self._h).decode() if self._h else ""
            self.Dispose()
            raise FrontError("Could not load the specified container. %s" % msg)
        lib().vpzr_set_clip_samples(self._h, int(clip_samples))
        lib().vpzr_set_sample_format(self._h, 1 if s16 else 0)
        lib().vpzr_set_batch_packets(self._h, batch_packets)
        import weakref
        ctx._children.append(weakref.ref(self))  # the reader owns a decoder: it must go before the context

    def close(self):
        self.Dispose()

    Channels = property(lambda self: lib().vpzr_channels(self._h))
    SampleRate = property(lambda self: lib().vpzr_sample_rate(self._h))
    SamplePosition = property(lambda self: lib().vpzr_sample_position(self._h))
    IsEndOfStream = property(lambda self: bool(lib().vpzr_is_end_of_stream(self._h)))
    HasClipped = property(lambda self: bool(lib().vpzr_has_clipped(self._h)))
    TotalSamples = property(lambda self: lib().vpzr_total_samples(self._h))
    StreamSerial = property(lambda self: lib().vpzr_stream_serial(self._h))
    StreamCount = property(lambda self: lib().vpzr_stream_count(self._h))

    def FindNextStream(self):
        """VorbisReader.FindNextStream (VorbisReader.cs:191-194)"""
        return bool(lib().vpzr_find_next_stream(self._h))

    def SwitchStreams(self, index):
        """VorbisReader.SwitchStreams (VorbisReader.cs:197-217): True when channels or sample rate changed."""
        rc = lib().vpzr_switch_streams(self._h, int(index))
        if rc < 0:
            raise capi.SynthError(rc, "SwitchStreams: index out of range")
        return bool(rc)

    def SeekTo(self, samplePosition, seekOrigin=0):
        """StreamDecoder.SeekTo(long, SeekOrigin) (StreamDecoder.cs:815-881); seekOrigin 0 Begin, 1 Current, 2 End.
        Raises capi.SynthError where the reference throws (SeekOutOfRange / ArgumentOutOfRange / PreRoll)."""
        rc = lib().vpzr_seek_to(self._h, int(samplePosition), int(seekOrigin))
        if rc != 0:
            raise capi.SynthError(rc, lib().vpzr_last_error(self._h).decode())

    def ReadSamples(self, buffer, samplesToRead=None, channelStride=None):
        st = C.c_int(0)
        buf = buffer.reshape(-1)
        if buf.dtype == np.int16:
            n = lib().vpzr_read_samples_s16(self._h, buf.ctypes.data, buf.size, C.byref(st))
        elif samplesToRead is None:
            n = lib().vpzr_read_samples(self._h, buf.ctypes.data, buf.size, C.byref(st))
        else:
            n = lib().vpzr_read_samples_planar(self._h, buf.ctypes.data, buf.size, samplesToRead, channelStride,
                                               C.byref(st))
        if st.value != 0:
            raise capi.SynthError(st.value, lib().vpzr_last_error(self._h).decode())
        return int(n)

    def Dispose(self):
        if getattr(self, "_h", None):
            if getattr(self._ctx, "_h", None):  # a reader outliving its context cannot free device state any more
                lib().vpzr_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass
