// P/Invoke binding of include/vorbispizza_synth.h (ABI version 5) for the reference host.
//
// To be added on the reference side as NVorbis/Native/VorbisPizzaSynth.cs.  The style is the one the repository
// already uses for libvorbisfile (NVorbis.Tests/Bindings/Vorbisfile.cs:43-107: DllImport + Cdecl, LayoutKind.Sequential
// structs, negative status codes, caller-owned pinned buffers) with handle ownership as in
// NVorbis.Tests/Bindings/NativeDecoder.cs:6,16,46-55 (SafeHandle).  No .NET toolchain exists in the pipeline this file
// was written in: tests/test_csharp_binding_cpu.py parses this file and holds every DllImport (name, argument count and
// kinds), every Sequential struct (fields in order, array bounds) and every constant to include/vorbispizza_synth.h.
using System;
using System.Runtime.InteropServices;
using Microsoft.Win32.SafeHandles;

namespace NVorbis.Native
{
    internal static unsafe class VorbisPizzaSynth
    {
        private const string Lib = "vorbispizza_synth";              // libvorbispizza_synth.so
        public const int AbiVersion = 6;

        // status codes (vorbispizza_synth.h), mapped to exceptions by ThrowOnError below the way
        // NativeDecoder.cs:145-161 maps OV_*
        public const int Ok = 0, EInvalidArg = -1, EUnsupported = -2, EHip = -3, ENoMem = -4, EWindowMismatch = -5,
                         ENoDevice = -6, ECapacity = -7;
        public const int MemHost = 0, MemDevice = 1;
        public const int ResidueF32 = 0, ResidueI16 = 1;             // vpz_decoder_set_residue_format (ABI v5)
        public const int OutInterleaved = 0, OutPlanar = 1, OutInterleavedS16 = 2, OutPlanarS16 = 3;
        public const int ImdctFast = 0, ImdctExact = 1;
        public const int PostsStride = 64;                            // Floor1.Data.Posts = new int[64], Floor1.cs:17

        [Flags]
        public enum PacketFlags : byte
        {
            BlockFlag = 0x01,    // Mode._blockFlag
            PrevFlag = 0x02,     // Mode.cs:39, first bit
            NextFlag = 0x04,     // Mode.cs:39, second bit
            Eos = 0x08,          // packet.IsEndOfStream
            NotDecoded = 0x10,   // DecodeNextPacket returned null (StreamDecoder.cs:758-761)
            Interleaved = 0x20,  // residue is the Residue2 vector [n/2][channels] (Residue2.cs:31-34)
            NoFloor = 0x40,      // residue already is the floored spectrum
            Resync = 0x80,       // packet.IsResync (StreamDecoder.cs:718-722)
        }

        [StructLayout(LayoutKind.Sequential)]
        public struct Floor1Config { public int XCount; public int Multiplier; public fixed int XList[65]; }

        [StructLayout(LayoutKind.Sequential)]
        public struct Floor0Config { public int Order, Rate, BarkMapSize, AmpBits, AmpOfs; }   // Floor0.cs:29-35

        [StructLayout(LayoutKind.Sequential)]
        public struct MappingConfig
        {
            public int CouplingSteps;
            public fixed byte CouplingMagnitude[256];
            public fixed byte CouplingAngle[256];
            public fixed byte ChannelFloor[256];     // _submapFloor[_mux[ch]] per channel
            // ABI v4: the residue's support per block size ([0] Size0, [1] Size1), in bins of one channel -- what
            // Residue0.Decode clamps itself to (Residue0.cs:122-125: Math.Min(_begin, halfSize) .. Math.Min(_end, halfSize));
            // smallest begin / largest end over the mapping's submaps; a Residue2's range divided by its channel count
            // (rounded outward).  End 0 = not stated (the whole block).  Residue0 needs `internal int Begin => _begin;
            // internal int End => _end;` for the host to fill these in.
            public fixed int ResidueBegin[2];
            public fixed int ResidueEnd[2];
        }

        [StructLayout(LayoutKind.Sequential)]
        public struct StreamConfig
        {
            public int Channels, BlockSize0, BlockSize1;
            public int FloorCount; public Floor1Config* Floors;
            public int MappingCount; public MappingConfig* Mappings;
            public int ClipSamples;
            public byte* FloorTypes;          // null: every floor is type 1; else 0 / 1 per floor
            public Floor0Config* Floors0;     // read at the indices whose type is 0
        }

        [StructLayout(LayoutKind.Sequential)]
        public struct Packet
        {
            public int Stream; public PacketFlags Flags; public byte Mapping; public ushort Reserved;
            public long Granule;          // packet.GranulePosition, -1 if none
            public long ResidueOffset;    // float index of this packet's residue in the batch buffer
        }

        public sealed class ContextHandle : SafeHandleZeroOrMinusOneIsInvalid
        {
            public ContextHandle() : base(true) { }
            protected override bool ReleaseHandle() { vpz_context_destroy(handle); return true; }
        }

        public sealed class DecoderHandle : SafeHandleZeroOrMinusOneIsInvalid
        {
            public DecoderHandle() : base(true) { }
            protected override bool ReleaseHandle() { vpz_decoder_destroy(handle); return true; }
        }

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_abi_version();
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr vpz_error_string(int status);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_device_count();

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_context_create(int device, out ContextHandle ctx);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] private static extern void vpz_context_destroy(IntPtr ctx);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_context_synchronize(ContextHandle ctx);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr vpz_context_last_error(ContextHandle ctx);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr vpz_context_stream(ContextHandle ctx);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_context_timer_start(ContextHandle ctx);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_context_timer_stop(ContextHandle ctx, out float elapsedMs);

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_device_alloc(ContextHandle ctx, ulong bytes, out IntPtr devPtr);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_device_free(ContextHandle ctx, IntPtr devPtr);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_host_alloc(ContextHandle ctx, ulong bytes, out IntPtr hostPtr);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_host_free(ContextHandle ctx, IntPtr hostPtr);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_memcpy_h2d(ContextHandle ctx, IntPtr devDst, void* hostSrc, ulong bytes);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_memcpy_d2h(ContextHandle ctx, void* hostDst, IntPtr devSrc, ulong bytes);

        // == Mdct.Reverse(samples, buf2, n) per row (Mdct.cs:15-19)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int vpz_imdct_batch(ContextHandle ctx, int n, long count, float* spectra, float* output, int memSpace, int mode);

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int vpz_decoder_create(ContextHandle ctx, StreamConfig* cfg, int nStreams, out DecoderHandle decoder);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] private static extern void vpz_decoder_destroy(IntPtr decoder);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_decoder_reset(DecoderHandle decoder, int stream);

        // == Mapping.DecodePacket tail (Mapping.cs:166-195) + ReadNextPacket / OverlapBuffers (StreamDecoder.cs:640-694,
        // 764-791) + Store* (:515-638) for every packet of the batch.  pcmOut: float*, or short* for the S16 layouts.
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int vpz_decoder_synth(DecoderHandle decoder, long nPackets, Packet* packets, float* residue,
            long residueFloats, short* posts, byte* postCounts, long nRecords, int memSpace, void* pcmOut,
            long* streamOutOffset, long streamOutCapacity, int outLayout, long channelStride, long* samplesWritten);
        // per-packet status of the last call: Ok, or EWindowMismatch where OverlapBuffers would have thrown
        // (StreamDecoder.cs:777-778); nNotOk: how many packets are not Ok (status may be null with capacity 0)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int vpz_decoder_last_packet_status(DecoderHandle decoder, int* status, long capacity, long* nNotOk);

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int vpz_decoder_set_floor0_data(DecoderHandle decoder, float* amp, float* coeff, int coeffStride);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int vpz_decoder_last_packet_samples(DecoderHandle decoder, int* samples, long capacity);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_decoder_has_clipped(DecoderHandle decoder, int stream, out int hasClipped);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_decoder_position(DecoderHandle decoder, int stream, out long position);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_decoder_set_position(DecoderHandle decoder, int stream, long position);
        // ABI v5: `residue` of the following synth calls is short* (the same values as 16-bit integers: exact for integral residues --
        // every libvorbis stream --, half the bytes over the host link); ResidueF32 switches back
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_decoder_set_residue_format(DecoderHandle decoder, int format);
        // output areas of different sizes in one batch: capacity[s] tightens stream_out_capacity for stream s; (null, 0) removes the bounds
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_decoder_set_stream_capacities(DecoderHandle decoder, long* capacity, int n);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int vpz_decoder_set_host_threads(DecoderHandle decoder, int n);

        /// <summary>Status code to the exception the reference throws for the same condition.</summary>
        public static void ThrowOnError(int status, ContextHandle? ctx, string what)
        {
            if (status == Ok) return;
            string detail = ctx != null && !ctx.IsInvalid ? Marshal.PtrToStringAnsi(vpz_context_last_error(ctx)) ?? "" : "";
            string name = Marshal.PtrToStringAnsi(vpz_error_string(status)) ?? status.ToString();
            switch (status)
            {
                case EWindowMismatch:   // windowSlope.AsSpan(0, packetLen) in OverlapBuffers (StreamDecoder.cs:777-778)
                    throw new ArgumentOutOfRangeException(what, detail);
                case EInvalidArg: throw new ArgumentException(what + ": " + detail);
                case ENoMem: throw new OutOfMemoryException(what + ": " + detail);
                default: throw new InvalidOperationException(what + " failed (" + name + "): " + detail);
            }
        }
    }
}
