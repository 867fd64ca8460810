// Drop-in reader over libvorbispizza_host.so (include/vorbispizza_reader.h): the members of NVorbis.VorbisReader the
// synthesis path serves (IVorbisReader.cs:139,142; VorbisReader.cs:191-253), forwarded one to one.  The host library
// keeps what the C# host keeps on the CPU (Ogg paging, VorbisPacket bit reading, Huffman / codebook decode) in C++ and
// calls vpz_decoder_synth for every batch of packets.  To be added on the reference side as
// NVorbis/Native/GpuVorbisReader.cs; libvorbispizza_host.so links libvorbispizza_synth.so, both sit next to the assembly.
//
// tests/test_reader_gpu.py exercises exactly these entry points through ctypes (vorbispizza_amd.front.VorbisReader),
// written the way NVorbis.Tests/AssetTest.cs is.
using System;
using System.IO;
using System.Runtime.InteropServices;

namespace NVorbis.Native
{
    public sealed unsafe class GpuVorbisReader : IDisposable
    {
        private const string Host = "vorbispizza_host", Synth = "vorbispizza_synth";
        [DllImport(Synth, CallingConvention = CallingConvention.Cdecl)] static extern int vpz_context_create(int device, out IntPtr ctx);
        [DllImport(Synth, CallingConvention = CallingConvention.Cdecl)] static extern void vpz_context_destroy(IntPtr ctx);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_open_memory(IntPtr ctx, byte* data, ulong size, out IntPtr reader);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern void vpzr_close(IntPtr reader);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern IntPtr vpzr_last_error(IntPtr reader);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_find_next_stream(IntPtr reader);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_stream_count(IntPtr reader);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_switch_streams(IntPtr reader, int index);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_stream_serial(IntPtr reader);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_channels(IntPtr reader);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_sample_rate(IntPtr reader);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern long vpzr_sample_position(IntPtr reader);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern long vpzr_total_samples(IntPtr reader);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_is_end_of_stream(IntPtr reader);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_has_clipped(IntPtr reader);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_set_clip_samples(IntPtr reader, int clip);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_set_batch_packets(IntPtr reader, int packets);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_set_sample_format(IntPtr reader, int format);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern int vpzr_seek_to(IntPtr reader, long samplePosition, int origin);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern long vpzr_read_samples(IntPtr reader, float* buffer, long length, out int status);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern long vpzr_read_samples_s16(IntPtr reader, short* buffer, long length, out int status);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] static extern long vpzr_read_samples_planar(IntPtr reader, float* buffer, long length, long samplesToRead, long channelStride, out int status);

        private IntPtr _ctx, _reader;

        /// <param name="sixteenBit">deliver `(int)(x * 32768f)` clamped samples (what AssetTest.cs:131-132 computes),
        /// converted on the GPU; read them with ReadSamples(Span&lt;short&gt;).</param>
        public GpuVorbisReader(Stream stream, int device = 0, bool sixteenBit = false)
        {
            using var ms = new MemoryStream();
            stream.CopyTo(ms);
            byte[] data = ms.ToArray();          // vpzr_open_memory copies the packets out: pinned for the call only
            Check(vpz_context_create(device, out _ctx), "vpz_context_create");
            int rc;
            fixed (byte* p = data) rc = vpzr_open_memory(_ctx, p, (ulong)data.Length, out _reader);
            if (rc != 0) throw new ArgumentException("Could not load the specified container. " + LastError(), nameof(stream));
            if (sixteenBit) Check(vpzr_set_sample_format(_reader, 1), "vpzr_set_sample_format");
        }

        // chained / multiplexed containers: VorbisReader.Streams, FindNextStream, SwitchStreams (VorbisReader.cs:191-217)
        public int StreamCount => vpzr_stream_count(_reader);
        public int StreamSerial => vpzr_stream_serial(_reader);
        public bool FindNextStream() => vpzr_find_next_stream(_reader) != 0;
        public bool SwitchStreams(int index)
        {
            int rc = vpzr_switch_streams(_reader, index);
            if (rc < 0) throw new ArgumentOutOfRangeException(nameof(index));
            return rc != 0;      // true: channel count or sample rate changed
        }

        public int Channels => vpzr_channels(_reader);
        public int SampleRate => vpzr_sample_rate(_reader);
        public long SamplePosition { get => vpzr_sample_position(_reader); set => SeekTo(value); }
        public long TotalSamples => vpzr_total_samples(_reader);
        public bool IsEndOfStream => vpzr_is_end_of_stream(_reader) != 0;
        public bool HasClipped => vpzr_has_clipped(_reader) != 0;
        public bool ClipSamples { set => vpzr_set_clip_samples(_reader, value ? 1 : 0); }   // before the first read
        /// <summary>Packets synthesised per GPU call (default 128): latency against throughput.</summary>
        public int BatchPackets { set => Check(vpzr_set_batch_packets(_reader, value), nameof(BatchPackets)); }

        public void SeekTo(long samplePosition, SeekOrigin origin = SeekOrigin.Begin)      // StreamDecoder.cs:815-881
            => Check(vpzr_seek_to(_reader, samplePosition, (int)origin), "SeekTo");         // Begin / Current / End = 0 / 1 / 2

        public int ReadSamples(Span<float> buffer)                                            // interleaved
        {
            fixed (float* p = buffer) { long n = vpzr_read_samples(_reader, p, buffer.Length, out int st); Check(st, "ReadSamples"); return (int)n; }
        }

        public int ReadSamples(Span<short> buffer)                                            // interleaved, 16-bit readers
        {
            fixed (short* p = buffer) { long n = vpzr_read_samples_s16(_reader, p, buffer.Length, out int st); Check(st, "ReadSamples"); return (int)n; }
        }

        public int ReadSamples(Span<float> buffer, int samplesToRead, int channelStride)      // planar
        {
            fixed (float* p = buffer) { long n = vpzr_read_samples_planar(_reader, p, buffer.Length, samplesToRead, channelStride, out int st); Check(st, "ReadSamples"); return (int)n; }
        }

        private string LastError() => _reader != IntPtr.Zero ? Marshal.PtrToStringAnsi(vpzr_last_error(_reader)) ?? "" : "";
        private void Check(int status, string what)
        {
            if (status == 0) return;
            if (status == -1) throw new ArgumentOutOfRangeException(what, LastError());   // VPZ_E_INVALID_ARG
            // VPZ_E_WINDOW_MISMATCH: `windowSlope.AsSpan(0, packetLen)` in OverlapBuffers (StreamDecoder.cs:777-778) throws this
            // out of the Read that reaches the packet; the next Read goes on behind it
            if (status == -5) throw new ArgumentOutOfRangeException(what, LastError());
            throw new InvalidOperationException(what + " failed (" + status + "): " + LastError());
        }

        public void Dispose()
        {
            if (_reader != IntPtr.Zero) { vpzr_close(_reader); _reader = IntPtr.Zero; }       // the reader (and its decoder) first,
            if (_ctx != IntPtr.Zero) { vpz_context_destroy(_ctx); _ctx = IntPtr.Zero; }       // then the context
        }
    }
}
