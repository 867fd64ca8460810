// Device-visible descriptors of one vpz_decoder_synth batch.  The host-side state machine
// (vpz_decoder.hip: ReadNextPacket / GetPacketInfo, integers only) resolves every packet into a
// FrameDesc; the kernels never see stream state other than these and the saved h tails.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace vpz {

// frame flags
constexpr uint32_t kFrameLong = 1u;        // block size = size1 (else size0)
constexpr uint32_t kFrameSlope1 = 2u;      // PacketInfo.LeftUseSize1: overlap uses the size1 slope
constexpr uint32_t kFrameNoFloor = 8u;     // spectrum is already floored (VPZ_PKT_NO_FLOOR)
constexpr uint32_t kFrameDrain = 4u;       // no new block: emit the previous tail un-windowed
                                           // (StreamDecoder.cs:451-455, quirk q4)
// group mode (the waves of a packet's channels share a workgroup and stage the packet in LDS):
constexpr uint32_t kFrameInterleaved = 16u;  // spec_off addresses the Residue2 vector [n/2][channels]
constexpr int kFrameBatchShift = 5;          // bits 5..7: this frame heads a batch of (value + 1) consecutive short blocks
                                             // that one pass synthesises (set by the kernel's own run builder only)
constexpr int kFrameStepsShift = 8;          // bits 8..15: coupling steps of the packet's mapping
constexpr int kFrameStepsOffShift = 16;      // bits 16..23: first step (pair index, < kGroupMaxStepPairs) in the steps table
constexpr uint32_t kFrameStepsOffMask = 0xFFu;
// bits 24..27: how many of the block's eight point groups (group m = bins [m * blocksize/16, (m + 1) * blocksize/16): the lane's
// m-th point in every transform layout) lie wholly beyond the residue's support (ABI v4, vpz_mapping_config.residue_end) -- their
// bins are +0.0 by the setup header's word and are neither loaded (group mode) nor de-coupled nor floor-multiplied.  0..8.
// bit 28: the steady state of every stream -- a 2048 block after a 2048 block with long windows on both sides: PacketInfo
// LeftStart 0, the overlap 1024 samples from position 1024 of the previous block, 1024 samples out, the size1 slope.  Settled
// once where the descriptor is built (one lane per frame, or the host), so that the frame loop tests one bit instead of eight
// fields (40 scalar instructions per pass, profiles/r4_isa_*.txt).
constexpr uint32_t kFrameSteady = 1u << 28;
__host__ __device__ inline bool frame_is_steady(uint32_t flags, int size1, int left_start, int packet_len, int prev_end, int out_count)
{
    return size1 == 2048 && (flags & kFrameLong) && (flags & kFrameSlope1) && !(flags & kFrameDrain) && left_start == 0 &&
           packet_len == 1024 && prev_end == 1024 && out_count == 1024;
}
constexpr int kFrameSkipShift = 24;
constexpr uint32_t kFrameSkipMask = 0xFu;
// per-mapping word (SynthArgs.map_bits): steps count / offset as in the frame flags, the skip of a long block in bits 24..27,
// of a short block in bits 28..31
constexpr int kMapSkipShortShift = 28;
constexpr int kGroupMaxChannels = 8;         // channels that fit one workgroup of 8 waves
constexpr int kFloor0Marker = 255;           // active-post count of a record whose floor is type 0 and is applied by the fused kernel
constexpr int kFloor0MaxBark = 1024;         // bark_map_size the fused route takes (the curve sits in a wave's LDS row)
constexpr int kGroupMaxStepPairs = 128;      // coupling steps (pairs) of all mappings staged in LDS

struct FrameDesc {        // 32 bytes: staged per run into LDS by the wavefront that owns the run
    int64_t spec_off;     // float offset of channel 0's spectrum; channel c at + c*(blocksize/2)
    int64_t out_off;      // first output sample (per channel) of this frame inside its stream
    int32_t rec;          // first channel record (packet_index * channels) for exec / floor data
    uint16_t left_start;  // PacketInfo.LeftStart (emission start inside this block's output)
    uint16_t packet_len;  // overlap length = prevStop - prevEnd (0 when there is no previous block)
    uint16_t prev_end;    // position of the previous block's tail in ITS output (prevEnd)
    uint16_t out_count;   // samples emitted per channel: rightStart' - LeftStart
    uint32_t flags;
};
static_assert(sizeof(FrameDesc) == 32, "FrameDesc is staged as two 16-byte words");
constexpr int kMaxRunLength = 32;
constexpr int kMaxRunLengthGeneral = 16;  // the general-size kernel variant trades descriptor space for tables
constexpr int kMaxRunLengthBig = 32;      // synth_big_kernel (4096 / 8192 blocks): a recomputed block is a whole big transform
#ifndef VPZ_DUAL_WAVES
#define VPZ_DUAL_WAVES 4   // wavefronts per workgroup of the stereo fast path (tuning builds: -DVPZ_DUAL_WAVES=10, one workgroup per CU)
#endif
constexpr int kMaxRunLengthDual = VPZ_DUAL_WAVES >= 10 ? 44 : 63;     // the stereo fast path: a run's frames (+ the recomputed one) are one lane each while
                                          // its descriptors are built; runs cut to equal COST need the room -- a run rich in
                                          // short blocks holds many frames (capped at 32 frames, such runs were done in 3/4 of
                                          // the time of the others: the launch waited for the all-long ones)

// any-block-size path (generic_*_kernel): one record per packet, full IMDCT outputs live in HBM
constexpr uint32_t kFrameSaveState = 16u;  // last block of its stream in this batch
struct GenericFrame {
    int64_t spec_off;    // planar spectrum in the temp, channel c at + c*(n/2)
    int64_t y_off;       // full IMDCT output of channel 0 in ybuf, channel c at + c*n
    int64_t prev_y_off;  // previous block's output (channel 0); -1: saved state, -2: none
    int64_t out_off;
    int32_t rec, stream, n, prev_n;
    int32_t left_start, packet_len, prev_end, out_count;
    uint32_t flags;
    int32_t reserved;
};

// how a run obtains the block that precedes its first frame
constexpr int32_t kPreNone = 0;     // the stream has no previous block (first packet / after reset)
constexpr int32_t kPreState = 1;    // previous block's h is in the decoder's device state
constexpr int32_t kPreRecompute = 2;// previous block is frame first-1 of this batch: recompute it
// The stereo fast path only: the previous block is the LAST block of the run that the wavefront before this one in the same
// workgroup walks (runs r - 1 and r of one stream, back to back, r not the first of its workgroup), and the run's first frame is
// the steady state (a 2048 block after a 2048 block, long windows on both sides).  Nothing is recomputed: the wave transforms its
// first frame, parks the lower half of its h in registers, walks the rest of its run, and emits the first frame's PCM LAST -- over
// the tail the neighbour's last tail save left in ITS LDS rows (a flag per wave in LDS says when that has happened; the
// neighbour never waits for anybody, so the chain of waits ends at the workgroup's first wave).
constexpr int32_t kPreNeighbour = 3;
constexpr uint32_t kRunSaveState = 1u;  // run ends the stream's batch: save h of its last block
// COMPACT runs: the host uploads two bytes per frame (the packet's window flags and its mapping index) instead of a
// 32-byte FrameDesc; the wave that owns the run derives its descriptors -- window geometry from the flag pairs,
// residue / output offsets by a prefix sum over the run -- into the LDS area explicit descriptors are staged in.
// Possible when the batch is what real hosts produce (see run_state_machine_parallel) and the packets' residues
// lie back to back.
constexpr uint32_t kRunCompact = 2u;
constexpr uint32_t kRunLastTrimmed = 4u;  // the run's last frame: out_count / left_start as given (EOS trim, :658-666)
// compact frame byte
constexpr uint32_t kCfNoFloor = 8u;       // bits 0..2: vpz_packet.flags & 7
constexpr uint32_t kCfInterleaved = 16u;  // residue is the Residue2 vector AND the batch runs in group mode
constexpr uint32_t kCfSkip = 128u;        // the packet was skipped (window mismatch): a frame that does nothing

struct RunDesc {         // 64 bytes
    int32_t first;       // index of the first frame (FrameDesc index, or packet index for a compact run)
    int32_t count;       // frames in the run
    int32_t stream;
    int32_t pre_kind;
    uint32_t flags;
    int32_t prev_long;   // kPreState: 1 if the saved block was a size1 block
    // ---- compact runs only; "first staged frame" = frame `first`, or `first - 1` when that block is recomputed
    int32_t rec_base;    // channel record of the first staged frame; consecutive frames are consecutive packets
    uint16_t prev_end0, prev_stop0;  // window of the block in front of the first staged frame (has_prev0)
    int64_t spec_base;   // residue offset of the first staged frame
    int64_t out_base;    // output offset (samples per channel, inside the stream) of frame `first`
    uint16_t last_out_count, last_left_start;  // kRunLastTrimmed
    uint8_t has_prev0;
    uint8_t pad[3];
    int32_t clip_epoch;  // what the run stores into clipped[stream] when a sample was clipped (see vpz_decoder_reset)
    int32_t state_slot;  // which copy of the stream's saved state is current (0 / 1)
};
static_assert(sizeof(RunDesc) == 64, "RunDesc layout");

// Mode.GetPacketInfo by (block | prev << 1 | next << 2), what the compact path needs of it
struct PacketGeom {
    uint16_t left_start, right_start, right_end, left_use_size1;
};

// Floor1 static tables on the device (Floor1.cs:30-31, 108-149), one per vpz_floor1_config, laid out for the walk of
// floor1_unwrap_kernel: everything a step needs that does not depend on the packet comes with ONE 8-byte read, which
// can be issued a step ahead.
struct FloorDev {
    int32_t x_count;
    int32_t multiplier;
    int32_t range;
    int32_t reserved;
    // UnwrapPosts step i >= 2 (Floor1.cs:286-352): .x = low neighbour | high neighbour << 8 | (x[i] - x[low]) << 16,
    //                                             .y = x[high] - x[low]   (RenderPoint's adx)
    uint32_t step[64][2];
    // the posts in X order (Floor1.cs:129-149): post index | x << 16
    uint32_t sorted[64];
};

// The tuning switches of the fused kernels (VPZ_SYNTH_ABLATE, VPZ_GROUP_DMA) exist in tuning builds only
// (VPZ_EXTRA_HIPCC_FLAGS=-DVPZ_TUNING): in the product every test of them folds away at compile time.  They used to be
// read from the kernel arguments inside the frame loop -- scalar loads whose `s_waitcnt lgkmcnt(0)` drains the wave's LDS
// queue as well (profiles/r4_isa_*.txt).
#ifdef VPZ_TUNING
#define VPZ_ABLATE(a) ((a).ablate)
#define VPZ_GROUP_DMA(a) ((a).group_dma != 0)
#else
__host__ __device__ constexpr int vpz_no_switches() { return 0; }  // (a call, so that `x && (0 & bit)` draws no warning)
#define VPZ_ABLATE(a) vpz_no_switches()
#define VPZ_GROUP_DMA(a) false
#endif

struct SynthArgs {
    const FrameDesc *frames;    // explicit descriptors (nullptr when every run is compact)
    const uint8_t *cflags;      // compact runs: [frame] flag byte, [frame] mapping index,
    const uint8_t *cmap;
    const uint8_t *run_inline;  // stereo fast path, compact runs: [run][32] -- flag byte and mapping index of the run's first 16 STAGED
                                // frames (16 + 16 bytes, zero beyond the run), at an address the wave knows before its run record
                                // has arrived; nullptr: the bytes are read from cflags / cmap at the run's `first`
    const uint32_t *map_bits;   // [mapping] group-mode flag bits of a floored frame of that mapping (stage / steps)
                                // (pair route: [pair][mapping], the steps of that pair only, in the pair's own step table)
    const uint8_t *pair_ch;     // pair route (synth_pairs_kernel): [pair][2] the pair's channels -- the first is "channel 0" of its steps
    int32_t n_pairs;            // channels / 2
    int32_t n_mappings;
    PacketGeom geom[8];
    const RunDesc *runs;
    int32_t n_runs;
    int32_t channels;
    int32_t size0, size1;
    const float *spec;          // spectra: caller residue (planar, or interleaved in group mode) or the coupling temp
    const uint8_t *ccount;      // [rec] active floor posts (floor1_unwrap_kernel); 0 => ExecuteChannel false;
                                // nullptr => every channel executes, no floor
    const int32_t *cposts;      // [rec][64] active posts in X order: x | (finalY * multiplier) << 16
    const uint8_t *steps;       // coupling steps of all mappings, pairs (mag | 0x80 where a level starts, ang)   (group mode)
    int32_t n_step_pairs;
    int32_t max_steps;          // most coupling LEVELS any mapping has: barriers per frame in group mode
    int32_t group;              // 1: channels of a run share a workgroup (LDS staging), 0: waves are independent
    int32_t group_dma;          // group mode: an interleaved packet lands in the group's rows AS IT IS ([bin][C], LDS-DMA) and every
                                // wave picks its channel up with the inverse coupling applied in registers (even channel
                                // counts, no channel in more than one step of any mapping)
    // type-0 floors in the stereo fast path (a record whose post count is kFloor0Marker): the record's curve over the bark
    // indices, curve[rec * f0_stride + k] (floor0_curve_kernel), and per (floor, block size) the bark index of every bin in the
    // order a lane holds its bins: f0_bark[(floor * 2 + long) * 1024 + lane * 16 + 2 * m + e] = barkMap[2 * (lane + 64 m) + e]
    // for a 2048 block, lane < 8 and bin 2 * (lane + 8 m) + e for a 256 one
    const float *f0_curve;
    const uint16_t *f0_bark;
    int32_t f0_stride;
    int32_t spec_i16;           // stereo fast path, floored batches: `spec` points to 16-bit integers (VPZ_RESIDUE_I16, ABI v5) -- the kernel
                                // widens them in registers (exact: every int16 is a float32), no float32 copy of the residue exists
    const float *inv_db;        // 256 floats
    float *state_h;             // [2][stream][channel][size1/2]: two copies -- a run that starts from the saved state reads
                                // copy RunDesc.state_slot, the run that ends its stream's batch writes the OTHER one (the
                                // two may be different wavefronts of one launch: no order between them)
    int64_t state_slot_floats;  // floats per copy
    const float2 *tw_long;      // fast tables of size1 (BlockTables::d_fast)
    const float2 *tw_short;     // fast tables of size0
    const float *slope0, *slope1;
    float *big_tail;            // synth_big_kernel, 8192 decoders: [run x channel][2048] the upper half of the wave's previous h
    float *out;
    const int64_t *stream_out_off;  // nullptr => 0
    int64_t channel_stride;
    int32_t interleaved;
    int32_t s16;                // PCM as int16 (`(int)(x * 32768f)` clamped) instead of float32
    int32_t clip;
    int32_t *clipped;           // [stream] sticky HasClipped
    int32_t no_batch;           // 1: one short block per pass (the host sets it when the runs were not cut by cost -- in runs of
                                // equal LENGTH the ones rich in short blocks would be done early -- or for VPZ_NO_BATCH=1)
    int32_t ablate;             // TUNING BUILDS ONLY (-DVPZ_TUNING; VPZ_SYNTH_ABLATE; wrong results, right timing): 1 no window / overlap-add / stores, 2 no
                                // transform, 4 no input loads, 8 no curve, 16 no coupling, 32 no staging (group mode) / no stores
                                // but the arithmetic (stereo path), 64 render every bin (group mode) / prologue only (stereo path),
                                // 128 no batches of short blocks (also set by the host when runs were not cut by cost); stereo
                                // path only: 256 no floor multiply, 512 no zero-tail bound, 1024 no tail save; group mode only: 2048 the interleaved
                                // packet lands in the rows linearly by LDS-DMA (wrong results), 4096 ... as a per-channel dword gather (right results)
    unsigned long long *stamps; // diagnostic builds only (-DVPZ_STAMPS): [16] cycles per phase, summed over the waves
};

}  // namespace vpz
