// P/Invoke binding of include/vorbispizza_multi.h -- the in-process multi-device dispatcher of libvorbispizza_host.so -- for
// the reference host.  One C# process, one context group per MI355X, the streams of a library partitioned contiguously over
// them, no collective: what a VorbisReader with N StreamDecoders (VorbisReader.cs:56-85) becomes when the synthesis of whole
// files is handed to the GPUs.  Style of NVorbis.Tests/Bindings/Vorbisfile.cs:43-107 (cdecl, LayoutKind.Sequential PODs,
// int status, SafeHandle-owned opaque handle, caller buffers pinned with `fixed` for the call).
using System;
using System.Runtime.InteropServices;

namespace NVorbis.Native
{
    public static unsafe class VorbisPizzaMulti
    {
        private const string Host = "vorbispizza_host";

        public const int Ok = 0, EArg = -1, EDevice = -2, ENomem = -3;
        // per-stream statuses
        public const int EOpen = -10, ECapacity = -11, ESynth = -12, ESetup = -13;

        [StructLayout(LayoutKind.Sequential)]
        public struct Options
        {
            public int HostThreads;         // entropy-decode threads over all devices (0: the CPUs the process may use, plus one per context)
            public int StreamsPerCall;      // streams per vpz_decoder_synth call (0: 16)
            public int ContextsPerDevice;   // contexts / issuing threads per device (0: 4 with 8 or more host threads per device, else 2)
            public int ClipSamples;         // StreamDecoder.ClipSamples
            public int SlotsPerDevice;      // sub-batches in flight per device (0: 4 * contexts + 4)
            public int FloatResidue;        // 0: integral residues cross the link as int16 (same values, half the bytes); non-zero: always float32
            public fixed int Reserved[2];
        }

        [StructLayout(LayoutKind.Sequential)]
        public struct StreamResult
        {
            public int Status, DeviceSlot, Channels, SampleRate;
            public long Samples, Packets, SkippedPackets;
        }

        [StructLayout(LayoutKind.Sequential)]
        public struct Stats
        {
            public double WallS;
            public fixed double DeviceWallS[16];
            public fixed double DeviceDecodeS[16];
            public fixed double DeviceSynthS[16];
            public fixed long DeviceStreams[16];
            public fixed long DeviceSamples[16];
            public int ThreadsPerDevice, PinnedMib;
        }

        public sealed class DispatcherHandle : SafeHandle
        {
            public DispatcherHandle() : base(IntPtr.Zero, true) { }
            public override bool IsInvalid => handle == IntPtr.Zero;
            protected override bool ReleaseHandle() { vpzm_destroy(handle); return true; }
        }

        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] public static extern int vpzm_create(int* deviceIds, int nDevices, Options* opt, out DispatcherHandle dispatcher);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] private static extern void vpzm_destroy(IntPtr dispatcher);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr vpzm_last_error(DispatcherHandle dispatcher);
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)] public static extern int vpzm_device_count(DispatcherHandle dispatcher);
        // containers data[k] / size[k] -> interleaved PCM at pcmOut + pcmOffset[k] (float32 or, for VPZ_OUT_INTERLEAVED_S16, int16)
        [DllImport(Host, CallingConvention = CallingConvention.Cdecl)]
        public static extern int vpzm_decode_library(DispatcherHandle dispatcher, int n, byte** data, ulong* size, int outLayout, void* pcmOut,
                                                     long* pcmOffset, long* pcmCapacity, StreamResult* results, Stats* stats);

        /// <summary>Decodes a library of in-memory .ogg files on every device of `deviceIds`: the C# a host adds around the call
        /// (pin the arrays, hand them over, read the per-stream results).  Calls from several threads on one dispatcher are
        /// safe: the library takes them in turn.</summary>
        public static StreamResult[] DecodeLibrary(DispatcherHandle dispatcher, byte[][] files, float[] pcm, long[] offsets, long[] capacities)
        {
            int n = files.Length;
            var results = new StreamResult[n];
            var pins = new GCHandle[n];
            var ptrs = new byte*[n];
            var sizes = new ulong[n];
            try
            {
                for (int k = 0; k < n; k++)
                {
                    pins[k] = GCHandle.Alloc(files[k], GCHandleType.Pinned);
                    ptrs[k] = (byte*)pins[k].AddrOfPinnedObject();
                    sizes[k] = (ulong)files[k].Length;
                }
                fixed (byte** pd = ptrs) fixed (ulong* ps = sizes) fixed (float* pp = pcm) fixed (long* po = offsets) fixed (long* pc = capacities)
                fixed (StreamResult* pr = results)
                {
                    int rc = vpzm_decode_library(dispatcher, n, pd, ps, VorbisPizzaSynth.OutInterleaved, pp, po, pc, pr, null);
                    if (rc != Ok) throw new InvalidOperationException("vpzm_decode_library: " + Marshal.PtrToStringAnsi(vpzm_last_error(dispatcher)));
                }
            }
            finally
            {
                foreach (var p in pins) if (p.IsAllocated) p.Free();
            }
            return results;
        }
    }
}
