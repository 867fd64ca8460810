// The patch of Mapping / StreamDecoder that queues frames instead of synthesising them (SURVEY.md 8b "what calls it"):
// the reference keeps Ogg paging, VorbisPacket bit reading and the entropy decode of floor / residue exactly where
// they are; this class collects what they produce per packet and hands whole batches to vpz_decoder_synth.
// To be added on the reference side (NVorbis/Native/GpuSynthesisBatch.cs) together with the call-site edits listed in
// INTEGRATION.md section 2.  Pure glue: no sample arithmetic happens in C#.
using System;
using System.Collections.Generic;
using System.Buffers;

namespace NVorbis.Native
{
    internal sealed unsafe class GpuSynthesisBatch : IDisposable
    {
        private readonly VorbisPizzaSynth.ContextHandle _ctx;
        private readonly VorbisPizzaSynth.DecoderHandle _decoder;
        private readonly int _channels, _halfSize1;
        private VorbisPizzaSynth.Packet[] _packets;
        private float[] _residue;       // batch buffer: packet p's residue at _packets[p].ResidueOffset
        private short[] _posts;         // [record][64], record = packet * channels + channel
        private byte[] _postCounts;     // Floor1.Data.PostCount per record (0 => ExecuteChannel false)
        private int _count;
        private long _residueUsed;

        /// <summary>Called once from StreamDecoder.LoadBooks (StreamDecoder.cs:262-321) with the setup-header products.</summary>
        public GpuSynthesisBatch(VorbisPizzaSynth.ContextHandle ctx, int channels, int blockSize0, int blockSize1,
                                 ReadOnlySpan<VorbisPizzaSynth.Floor1Config> floors,
                                 ReadOnlySpan<VorbisPizzaSynth.MappingConfig> mappings, bool clipSamples, int capacityPackets = 128)
        {
            _ctx = ctx;
            _channels = channels;
            _halfSize1 = blockSize1 / 2;
            _packets = new VorbisPizzaSynth.Packet[capacityPackets];
            _residue = new float[(long)capacityPackets * channels * _halfSize1];
            _posts = new short[capacityPackets * channels * VorbisPizzaSynth.PostsStride];
            _postCounts = new byte[capacityPackets * channels];
            fixed (VorbisPizzaSynth.Floor1Config* pf = floors)
            fixed (VorbisPizzaSynth.MappingConfig* pm = mappings)
            {
                var cfg = new VorbisPizzaSynth.StreamConfig
                {
                    Channels = channels, BlockSize0 = blockSize0, BlockSize1 = blockSize1,
                    FloorCount = floors.Length, Floors = pf, MappingCount = mappings.Length, Mappings = pm,
                    ClipSamples = clipSamples ? 1 : 0, FloorTypes = null, Floors0 = null,
                };
                VorbisPizzaSynth.ThrowOnError(VorbisPizzaSynth.vpz_decoder_create(ctx, &cfg, 1, out _decoder), ctx, "vpz_decoder_create");
            }
        }

        public bool IsFull => _count == _packets.Length;
        public int Count => _count;

        /// <summary>The residue span Mapping.DecodePacket decodes into instead of its ChannelBuffer (Mapping.cs:136-163):
        /// planar [channel][blockSize / 2], or the Residue2 vector [blockSize / 2][channels] with Interleaved set.</summary>
        public Span<float> NextResidue(int blockSize) => _residue.AsSpan((int)_residueUsed, _channels * (blockSize / 2));

        /// <summary>The posts row Floor1.Unpack fills for one channel (Floor1.cs:167-218).</summary>
        public Span<short> NextPosts(int channel) => _posts.AsSpan((_count * _channels + channel) * VorbisPizzaSynth.PostsStride, VorbisPizzaSynth.PostsStride);

        /// <summary>Appends the packet DecodeNextPacket has just entropy-decoded (StreamDecoder.cs:696-762).  A null decode is
        /// appended with NotDecoded so that the EOS drain (:451-455) happens where the reference does it.</summary>
        public void Append(VorbisPizzaSynth.PacketFlags flags, byte mappingIndex, long granulePosition, int blockSize,
                           ReadOnlySpan<byte> postCountPerChannel)
        {
            ref VorbisPizzaSynth.Packet p = ref _packets[_count];
            p.Stream = 0;
            p.Flags = flags;
            p.Mapping = mappingIndex;
            p.Granule = granulePosition;
            p.ResidueOffset = _residueUsed;
            postCountPerChannel.CopyTo(_postCounts.AsSpan(_count * _channels, _channels));
            if ((flags & VorbisPizzaSynth.PacketFlags.NotDecoded) == 0) _residueUsed += (long)_channels * (blockSize / 2);
            _count++;
        }

        /// <summary>One vpz_decoder_synth call for everything queued: PCM straight into the caller's span, interleaved as
        /// StreamDecoder.Read(Span&lt;float&gt;) returns it.  perPacketSamples receives what each packet contributed, so that
        /// Read can keep handing out at most one packet's worth per call (`while (idx == 0)`, StreamDecoder.cs:436).</summary>
        public long Flush(Span<float> pcm, Span<int> perPacketSamples)
        {
            if (_count == 0) return 0;
            long written = 0, notOk = 0;
            int rc;
            fixed (VorbisPizzaSynth.Packet* pk = _packets)
            fixed (float* res = _residue)
            fixed (short* po = _posts)
            fixed (byte* pc = _postCounts)
            fixed (float* dst = pcm)
            fixed (int* per = perPacketSamples)
            {
                // the extents are the queued part of the arrays: the library refuses a packet that points beyond them
                rc = VorbisPizzaSynth.vpz_decoder_synth(_decoder, _count, pk, res, _residueUsed, po, pc, (long)_count * _channels,
                                                        VorbisPizzaSynth.MemHost, dst, null, pcm.Length / _channels,
                                                        VorbisPizzaSynth.OutInterleaved, 0, &written);
                VorbisPizzaSynth.vpz_decoder_last_packet_samples(_decoder, per, Math.Min(perPacketSamples.Length, _count));
                if (rc == VorbisPizzaSynth.Ok) VorbisPizzaSynth.vpz_decoder_last_packet_status(_decoder, null, 0, &notOk);
            }
            int queued = _count;
            _count = 0;
            _residueUsed = 0;
            VorbisPizzaSynth.ThrowOnError(rc, _ctx, "vpz_decoder_synth");
            // A window mismatch costs only the offending packet, like the exception out of OverlapBuffers
            // (StreamDecoder.cs:777-778): the batch is valid; PacketStatus tells the Read that reaches the packet to throw.
            LastBatchHadMismatch = notOk > 0;
            _lastQueued = queued;
            return written;
        }

        /// <summary>Status of each packet of the last Flush (Ok / EWindowMismatch), for the Read loop that hands the
        /// batch out packet by packet.</summary>
        public void PacketStatus(Span<int> status)
        {
            fixed (int* st = status)
                VorbisPizzaSynth.vpz_decoder_last_packet_status(_decoder, st, Math.Min(status.Length, _lastQueued), null);
        }

        public bool LastBatchHadMismatch { get; private set; }
        private int _lastQueued;

        public void Reset()                                   // StreamDecoder.ResetDecoder (:357-369)
        {
            _count = 0;
            _residueUsed = 0;
            VorbisPizzaSynth.ThrowOnError(VorbisPizzaSynth.vpz_decoder_reset(_decoder, 0), _ctx, "vpz_decoder_reset");
        }

        public long Position
        {
            get { VorbisPizzaSynth.vpz_decoder_position(_decoder, 0, out long p); return p; }
            set => VorbisPizzaSynth.ThrowOnError(VorbisPizzaSynth.vpz_decoder_set_position(_decoder, 0, value), _ctx, "vpz_decoder_set_position");
        }

        public bool HasClipped { get { VorbisPizzaSynth.vpz_decoder_has_clipped(_decoder, 0, out int c); return c != 0; } }

        public void Dispose() => _decoder.Dispose();

        /// <summary>One decoder configuration for streams that come with different setup headers but the same channel
        /// count and block sizes: the union of their floors (equal ones shared) and of their mappings.  A packet of setup
        /// k with mapping index m carries mappingBase[k] + m in the merged batch -- one unwrap and one synthesis launch
        /// cover all the streams, with longer runs per wavefront (vorbispizza_amd/sharding.py: merge_setups is the
        /// same thing in Python; 128 streams of two kinds: 0.366 -> 0.320 ms per batch).</summary>
        public static (VorbisPizzaSynth.Floor1Config[] Floors, VorbisPizzaSynth.MappingConfig[] Mappings, int[] MappingBase)
            MergeSetups(IReadOnlyList<(VorbisPizzaSynth.Floor1Config[] Floors, VorbisPizzaSynth.MappingConfig[] Mappings)> setups,
                        int channels)
        {
            var floors = new List<VorbisPizzaSynth.Floor1Config>();
            var mappings = new List<VorbisPizzaSynth.MappingConfig>();
            var bases = new int[setups.Count];
            static bool Same(in VorbisPizzaSynth.Floor1Config a, in VorbisPizzaSynth.Floor1Config b)
            {
                if (a.XCount != b.XCount || a.Multiplier != b.Multiplier) return false;
                for (int i = 0; i < a.XCount; i++) if (a.XList[i] != b.XList[i]) return false;
                return true;
            }
            for (int k = 0; k < setups.Count; k++)
            {
                var remap = new byte[setups[k].Floors.Length];
                for (int f = 0; f < remap.Length; f++)
                {
                    int at = floors.FindIndex(x => Same(x, setups[k].Floors[f]));
                    if (at < 0) { at = floors.Count; floors.Add(setups[k].Floors[f]); }
                    remap[f] = (byte)at;
                }
                bases[k] = mappings.Count;
                foreach (var m in setups[k].Mappings)
                {
                    var merged = m;                                    // coupling steps unchanged
                    for (int ch = 0; ch < channels; ch++) merged.ChannelFloor[ch] = remap[m.ChannelFloor[ch]];
                    mappings.Add(merged);
                }
            }
            if (mappings.Count > 256) throw new ArgumentException("more than 256 mappings do not fit Packet.Mapping");
            return (floors.ToArray(), mappings.ToArray(), bases);
        }
    }
}
