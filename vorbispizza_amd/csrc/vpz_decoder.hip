// vpz_decoder_*: host half of the synthesis path.  Mirrors, with integers only, what
// StreamDecoder.ReadNextPacket / Read and Mode.GetPacketInfo decide per packet
// (StreamDecoder.cs:418-498, 640-694; Mode.cs:30-66) and turns a batch of packets into frame /
// run descriptors for the kernels in synth_kernels.hip.  No sample arithmetic happens here.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "host_pool.hpp"
#include "synth_desc.hpp"
#include "vpz_internal.hpp"
#include "../../include/vorbispizza_synth_debug.h"

namespace vpz {

hipError_t launch_floor1_unwrap(int n_rec, const int16_t *posts, const uint8_t *post_counts, const uint8_t *rec_info,
                                const FloorDev *floors, int n_floors, int32_t *cposts, uint8_t *ccount, int16_t *dbg_y,
                                uint8_t *dbg_f, hipStream_t stream, int f0_fused = 0);
hipError_t launch_floor0_curves(int n_rec, const uint8_t *rec_info, const void *floors, const float *amp, const float *coeff,
                                int coeff_stride, int k_stride, const float *wtab, float *curve, const uint8_t *post_counts,
                                uint8_t *ccount, int32_t *cposts, hipStream_t stream);
hipError_t launch_floor0_wtab(const void *floors, int n_floors, int k_stride, float *wtab, hipStream_t stream);
hipError_t launch_floor1_render(int n_rec, const int32_t *cposts, const uint8_t *ccount, const uint8_t *rec_info,
                                int half0, int half1, uint8_t *curve_y, hipStream_t stream);
hipError_t launch_coupling(const void *pkts, int n_pkts, const uint8_t *steps, int channels,
                           const float *residue, float *temp, int max_half, hipStream_t stream);
hipError_t launch_synth(const SynthArgs &args, bool has_floor, hipStream_t stream);
bool synth_supports_sizes(int size0, int size1);
bool synth_needs_general(int size0, int size1);
int synth_resident_waves(bool has_floor, int num_cu, int channels, bool group);
bool synth_group_supported(int channels);
bool synth_dual_supported(int channels, int size0, int size1);
hipError_t launch_synth_dual(const SynthArgs &args, bool has_floor, bool interleaved_in, hipStream_t stream);
int synth_dual_resident_slots(bool has_floor, int num_cu);
int synth_dual_waves();
bool synth_pairs_supported(int channels, int size0, int size1);
hipError_t launch_synth_pairs(const SynthArgs &args, bool has_floor, bool interleaved_in, hipStream_t stream);
bool synth_big_supported(int size0, int size1);
hipError_t launch_synth_big(const SynthArgs &args, bool has_floor, hipStream_t stream);
int synth_big_resident_waves(bool has_floor, int num_cu, int size0, int size1);
int64_t synth_big_tail_floats(int size1, int64_t n_items);
hipError_t launch_generic_floor(const GenericFrame *frames, int n_frames, int channels, int half1, float *spec,
                                const uint8_t *post_counts, const uint8_t *curve_y, const float *inv_db,
                                hipStream_t stream);
hipError_t launch_generic_ola(const GenericFrame *frames, int n_frames, int channels, int size0, int size1,
                              const float *ybuf, float *state_y, const float *slope0, const float *slope1, float *out,
                              const int64_t *stream_out_off, int64_t channel_stride, int interleaved, int clip,
                              int32_t *clipped, int s16, hipStream_t stream);
hipError_t launch_generic_save_state(const GenericFrame *frames, const int32_t *save_list, int n_save, int channels,
                                     int size1, const float *ybuf, float *state_y, hipStream_t stream);
hipError_t launch_floor0_apply(const void *recs, int n_recs, const void *floors, const int32_t *bark_maps,
                               const float *amp, const float *coeff, int coeff_stride, float *spec,
                               hipStream_t stream);
size_t floor0_dev_size();
size_t floor0_rec_size();
void fill_floor0_dev(void *dst, int order, int bark_map_size, int amp_ofs, int64_t off_short, int64_t off_long);
void fill_floor0_rec(void *dst, int64_t spec_off, int rec, int floor, int half, int is_long);
size_t coupling_packet_size();
void fill_coupling_packet(void *dst, int64_t src_off, int64_t dst_off, int32_t half, int32_t steps_off,
                          int32_t steps, int32_t interleaved);

static const uint32_t k_inverse_db_bits[256] = {
#include "floor1_inverse_db_bits.inc"
};

// StreamDecoder.cs:45-49 + position / EOS bookkeeping, per stream
struct StreamState {
    bool has_prev = false;      // _prevPacketBuf != null
    bool prev_long = false;     // block flag of the packet held in _prevPacketBuf
    int prev_start = 0, prev_end = 0, prev_stop = 0;
    int64_t current_position = 0;
    bool has_position = true;   // ProcessHeaderPackets: _currentPosition = 0; _hasPosition = true (:165-168)
    bool eos_found = false;
    bool has_clipped = false;
    int32_t clip_epoch = 1;     // resets so far + 1: clipped[stream] == clip_epoch <=> HasClipped
    int32_t state_slot = 0;     // which of the two device copies of the saved overlap state is current: a batch reads it in
                                // the stream's first run and writes the other copy in its last one -- two wavefronts of
                                // one launch with no order between them (the last run of a stream rich in short blocks can
                                // be done before the first one has read)
};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

// Pinned host arena for the per-call descriptor uploads: copies out of it are truly asynchronous,
// and the next call waits (on an event) only for the previous call's uploads before reusing it.
struct PinnedArena {
    char *base = nullptr;
    char *mapped = nullptr;   // the same memory as the GPU addresses it (hipHostGetDevicePointer)
    size_t cap = 0, used = 0;
    hipEvent_t uploaded = nullptr;   // the arena is in its device mirror
    bool pending = false;
    DevBuf dev;  // device mirror, same layout: one hipMemcpyAsync per call
};

// Mode.cs:30-66
struct PacketInfo {
    int length, left_use_size1, left_start, left_end, right_start, right_end;
};
// VPZ_RESIDUE_I16: 16-bit residue values to the float32 the synthesis kernels read (exact: every int16 is a float32).  Eight values
// per lane and step where the source is 16-byte aligned (a staging buffer always is), one otherwise.
__global__ __launch_bounds__(256) void widen_i16_kernel(const int16_t *__restrict__ in, float *__restrict__ out, long n)
{
    const long stride = (long)gridDim.x * 256 * 8;
    if ((reinterpret_cast<uintptr_t>(in) & 15) == 0) {
        for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
            if (i + 8 <= n) {
                const uint4 v = *reinterpret_cast<const uint4 *>(in + i);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                float f[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f[2 * k] = (float)(int16_t)(w[k] & 0xFFFFu);
                    f[2 * k + 1] = (float)(int16_t)(w[k] >> 16);
                }
                *reinterpret_cast<float4 *>(out + i) = make_float4(f[0], f[1], f[2], f[3]);
                *reinterpret_cast<float4 *>(out + i + 4) = make_float4(f[4], f[5], f[6], f[7]);
            } else {
                for (long j = i; j < n; ++j) out[j] = (float)in[j];
            }
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = (float)in[i];
    }
}
static hipError_t launch_widen_i16(const void *in, float *out, int64_t n, int num_cu, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const int64_t want = (n + 256 * 8 - 1) / (256 * 8);
    const int grid = (int)std::min<int64_t>(want, (int64_t)num_cu * 8);
    hipLaunchKernelGGL(widen_i16_kernel, dim3(grid), dim3(256), 0, stream, static_cast<const int16_t *>(in), out, (long)n);
    return hipGetLastError();
}

struct Decoder {
    Context *ctx = nullptr;
    PinnedArena arenas[2];
    int arena_idx = 0;
    std::vector<int64_t> s_base, s_cnt, out_count, anchor_pkt;  // per-stream scratch of a synth call
    int channels = 0, size0 = 0, size1 = 0, clip = 0;
    int n_streams = 0;
    std::vector<int64_t> stream_caps;  // vpz_decoder_set_stream_capacities: empty, or one bound per stream
    std::vector<vpz_floor1_config> floors;
    std::vector<vpz_mapping_config> mappings;
    std::vector<StreamState> states;
    BlockTables *t0 = nullptr, *t1 = nullptr;
    FloorDev *d_floors = nullptr;
    float *d_state_h = nullptr;
    int32_t *d_clipped = nullptr;
    uint8_t *d_steps = nullptr;              // coupling steps of all mappings, pairs (mag, ang)
    uint8_t *d_steps_lvl = nullptr;          // the same with bit 7 of `mag` set where a level of disjoint steps starts
    uint32_t *d_map_bits = nullptr;          // per mapping: group-mode frame flag bits (stage / steps) of a floored frame
    std::vector<uint8_t> mapping_uses_floor0;
    std::vector<int64_t> out_off_scratch;    // compact batches: output offset of every frame (host only)
    std::vector<int32_t> trim_out_count, trim_left_start;  // per stream: EOS-trimmed last frame of the batch, -1: none
    bool no_compact = false;                 // VPZ_NO_COMPACT=1: always upload explicit frame descriptors (A/B tests)
    std::vector<uint8_t> cut_code;           // cut_runs: one byte per packet (block size, batch eligibility)
    int cut_hint_R = 0;                      // ... and the cost target the last call's runs were fitted with
    int64_t cut_hint_slots = 0, cut_hint_target = 0, cut_hint_frames = -1, cut_hint_runs = 0;
    int cut_hint_streams = -1;
    int64_t cut_hint_heavy = -1;             // runs that start below this cost position get the heavier target (cut_runs: THE SKEW)
    struct CutSeg { int32_t stream, off, cnt; };  // packets [s_base[stream] + off, + cnt): what one thread cuts into runs
    std::vector<CutSeg> cut_segs;            // the streams of a call, long ones of a batch of few streams in pieces (cut_runs)
    std::vector<int64_t> s_units;            // cost of each segment's packets in this call (cut_runs), then
    std::vector<int64_t> cut_prefix;         // ... the cost of all segments in front of each one
    size_t zero_copy_max = 8u << 20;         // arenas up to this size are read in place by the kernels (VPZ_ZERO_COPY_MAX;
                                             // half a million packets, 1.5 MB: 2.42 -> 2.35 ms against the copy)
    std::vector<int32_t> mapping_steps_off;  // per mapping: offset into d_steps (pairs*2), -1 none
    std::vector<uint8_t> mapping_skip[2];    // per mapping and block size: point groups (of 8) beyond the residue's support (ABI v4)
    DevBuf b_curve, b_temp, b_cposts, b_ccount;
    // group mode of synth_kernel (channels of a packet share a workgroup; de-interleave + coupling in LDS)
    bool group_ok = false;       // channel count, step tables and floor types allow it
    bool group_dma = false;      // ... and interleaved packets may land in LDS as they are (SynthArgs.group_dma)
    // stereo fast path (synth_dual.hip: one wavefront per stream synthesises both channels, coupling in registers)
    bool dual_ok = false;        // two channels, 256 / 2048 blocks, type-1 floors only (VPZ_NO_DUAL=1: off, for A/B tests)
    // ... and its kernel for channel PAIRS (synth_pairs.hip): 4, 6, 8, ... channels that the coupling steps of all mappings join two
    // by two -- every pair is a stereo stream to the arithmetic (VPZ_NO_PAIRS=1: off, for A/B and bit-equality tests)
    bool pairs = false;          // (implies dual_ok)
    bool pairs_always = false;   // VPZ_PAIRS=1: the pair route wherever it can run, also where group mode is as fast or faster
    uint8_t *d_pair_ch = nullptr;         // [pair][2]: the pair's channels, coupled ones first ("channel 0" of its steps), in channel order
    uint32_t *d_pair_map_bits = nullptr;  // [pair][mapping]: SynthArgs.map_bits of the pair route
    uint8_t *d_pair_steps = nullptr;      // the pairs' step lists: (0 | 1: which of the pair's channels is the magnitude, unused)
    int n_pair_step_pairs = 0;
    int max_steps = 0, n_step_pairs = 0;
    int host_threads = 0;        // parties of the parallel state machine (VPZ_HOST_THREADS; 0: pick)
    int64_t par_min_packets = 16384;  // batches below this take the serial state machine (VPZ_PAR_MIN_PACKETS)
    DevBuf b_in_res, b_in_posts, b_in_counts, b_out;  // VPZ_MEM_HOST staging
    DevBuf b_in_res16;              // VPZ_RESIDUE_I16: the int16 values as they came over the link, widened into b_in_res
    int residue_format = VPZ_RESIDUE_F32;
    DevBuf b_ybuf;                                    // any-block-size path
    DevBuf b_bigtail;                                 // synth_big_kernel, 8192 decoders: the waves' tails (SynthArgs.big_tail)
    bool generic = false;  // a block size the fused kernels do not take (64, 128): three-pass path (synth_kernels.hip)
    bool big = false;      // the long block is 4096 or 8192 samples: synth_big_kernel (synth_big.hip; VPZ_NO_BIG=1: the three-pass path)
    // type-0 floors (Floor0.cs)
    std::vector<uint8_t> floor_types;
    std::vector<vpz_floor0_config> floors0;
    void *d_floors0 = nullptr;
    int32_t *d_bark_maps = nullptr;
    // type-0 floors inside the stereo fast path (floor0_curve_kernel + floor0_multiply): possible when every type-0 floor's
    // bark map has at most kFloor0MaxBark entries; f0_k = the largest of them (row length of the per-record curves)
    bool f0_fused = false;
    bool has_floor1 = false;  // some floor of the setup is of type 1
    int f0_k = 0;
    uint16_t *d_f0_bark = nullptr;   // [floor][short / long][1024]: bark index of every bin in lane order (SynthArgs.f0_bark)
    float *d_f0_w = nullptr;         // [floor][f0_k]: 2 cos(pi k / bark_map_size), Floor0's wMap (floor0_wtab_kernel)
    DevBuf b_f0curve;
    DevBuf b_in_amp, b_in_coeff;
    const float *f0_amp = nullptr, *f0_coeff = nullptr;
    int32_t f0_stride = 0;
    PacketInfo packet_info[8];  // Mode.GetPacketInfo by (block | prev << 1 | next << 2) == vpz_packet.flags & 7
    int run_length_override = 0;
    int dual_run = 8;  // preferred run length of the stereo fast path's chained runs (VPZ_DUAL_RUN)
    bool no_direct_i16 = false;
    bool no_run_inline = false;
    bool no_chain = false;  // VPZ_NO_CHAIN=1 (A/B tests): no run of the stereo fast path takes its predecessor's tail over in LDS
    int ablate = 0;  // VPZ_SYNTH_ABLATE, tuning experiments only
    bool no_early_upload = false;  // VPZ_NO_EARLY_UPLOAD=1 (A/B tests): a host-memory call's H2D copies stay behind its host pass
    std::vector<int32_t> packet_samples;  // per packet of the last synth call
    std::vector<int64_t> mismatch_packets;  // packets of the last synth call that failed the window check (skipped)
};

static int grow(Context *ctx, DevBuf &b, size_t need)
{
    return ensure_stage(ctx, &b.p, &b.bytes, need ? need : 1);
}

static int arena_begin(Context *ctx, PinnedArena &A, size_t need)
{
    if (A.pending) {
        VPZ_HIP_TRY(ctx, hipEventSynchronize(A.uploaded));
        A.pending = false;
    }
    // (the event says "the arena's last readers are done", nothing about memory: without the system-scope fence a default event
    // carries -- a write-back and invalidation of the device's caches behind every call -- the next call's kernels follow this
    // call's without that pause; VPZ_ARENA_EVENT_FENCE=1 restores the default event for A/B runs)
    if (!A.uploaded) {
        static const bool fence = getenv("VPZ_ARENA_EVENT_FENCE") && atoi(getenv("VPZ_ARENA_EVENT_FENCE"));
        VPZ_HIP_TRY(ctx, hipEventCreateWithFlags(&A.uploaded, hipEventDisableTiming | (fence ? 0u : (unsigned)hipEventDisableSystemFence)));
    }
    if (A.cap < need) {
        if (A.base) VPZ_HIP_TRY(ctx, hipHostFree(A.base));
        A.base = nullptr;
        A.mapped = nullptr;
        A.cap = 0;
        const size_t want = need + need / 2 + 4096;
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&A.base), want, hipHostMallocMapped);
        if (e != hipSuccess) return set_error(ctx, VPZ_E_NOMEM, "hipHostMalloc(descriptor arena)", e);
        A.cap = want;
        void *m = nullptr;
        A.mapped = hipHostGetDevicePointer(&m, A.base, 0) == hipSuccess ? static_cast<char *>(m) : nullptr;
        if (!A.mapped) (void)hipGetLastError();
    }
    A.used = 0;
    return VPZ_OK;
}

// Carves `count` objects out of the call's arena.  open_arena sizes the arena for everything a call can ask for; should a
// request not fit after all, ArenaOverflow is thrown -- vpz_decoder_synth turns it into VPZ_E_NOMEM -- instead of a write
// beyond the allocation.
struct ArenaOverflow {};
template <typename T>
static T *arena_alloc(PinnedArena &A, size_t count)
{
    A.used = (A.used + 63) & ~(size_t)63;
    if (A.used > A.cap || sizeof(T) * count > A.cap - A.used) throw ArenaOverflow();
    T *p = reinterpret_cast<T *>(A.base + A.used);
    A.used += sizeof(T) * count;
    return p;
}

static PacketInfo get_packet_info(int size0, int size1, bool block_flag, bool prev_flag, bool next_flag)
{
    PacketInfo pi;
    const int size = block_flag ? size1 : size0;
    const bool prev = block_flag ? prev_flag : true;
    const bool next = block_flag ? next_flag : true;
    const int center = size / 2;
    if (prev) {
        pi.left_start = 0; pi.left_end = center; pi.length = size / 2; pi.left_use_size1 = block_flag ? 1 : 0;
    } else {
        pi.left_start = (size - size0) / 4; pi.left_end = (size + size0) / 4; pi.length = size0 / 2;
        pi.left_use_size1 = 0;
    }
    if (next) { pi.right_start = center; pi.right_end = size; }
    else { pi.right_start = (size * 3 - size0) / 4; pi.right_end = (size * 3 + size0) / 4; }
    return pi;
}

// Floor1.cs:108-149: neighbours and sort order of the X list
static int build_floor(const vpz_floor1_config &c, FloorDev *f)
{
    memset(f, 0, sizeof *f);
    if (c.x_count < 2 || c.x_count > 64 || c.multiplier < 1 || c.multiplier > 4) return VPZ_E_INVALID_ARG;
    static const int range_lookup[4] = {128, 64, 43, 32};  // Floor1.cs:36
    f->x_count = c.x_count;
    f->multiplier = c.multiplier;
    f->range = range_lookup[c.multiplier - 1] * 2;
    for (int i = 0; i < c.x_count; ++i)
        if (c.x_list[i] < 0 || c.x_list[i] > 32767) return VPZ_E_INVALID_ARG;
    // Floor1.cs:96-97: `_xList[0] = 0; _xList[1] = 1 << rangeBits`.  The render starts its first segment at the
    // post with x == 0 (bin 0 looks its segment up by counting the posts at or below it)
    if (c.x_list[0] != 0 || c.x_list[1] <= 0) return VPZ_E_INVALID_ARG;
    std::vector<int> order(c.x_count);
    for (int i = 0; i < c.x_count; ++i) order[i] = i;
    // sortIdx[0], [1] start as 0, 1 and take part in the exchange sort like every other entry
    for (int i = 0; i < c.x_count - 1; ++i)
        for (int j = i + 1; j < c.x_count; ++j) {
            if (c.x_list[i] == c.x_list[j]) return VPZ_E_INVALID_ARG;  // InvalidDataException :141
            if (c.x_list[order[i]] > c.x_list[order[j]]) std::swap(order[i], order[j]);
        }
    for (int i = 0; i < c.x_count; ++i) f->sorted[i] = (uint32_t)order[i] | ((uint32_t)c.x_list[order[i]] << 16);
    for (int i = 2; i < c.x_count; ++i) {
        int lo = 0, hi = 1;
        for (int j = 2; j < i; ++j) {
            const int t = c.x_list[j];
            if (t < c.x_list[i]) { if (t > c.x_list[lo]) lo = j; }
            else                 { if (t < c.x_list[hi]) hi = j; }
        }
        // (a post below x[0] = 0 cannot exist; one above x[1] keeps hi = 1 and the reference extrapolates: the
        // differences below are what RenderPoint computes, whatever their sign)
        f->step[i][0] = (uint32_t)lo | ((uint32_t)hi << 8) | ((uint32_t)((c.x_list[i] - c.x_list[lo]) & 0xFFFF) << 16);
        f->step[i][1] = (uint32_t)((c.x_list[hi] - c.x_list[lo]) & 0xFFFF);
    }
    return VPZ_OK;
}


}  // namespace vpz

struct vpz_decoder {
    vpz::Decoder impl;
};

using namespace vpz;

extern "C" {

int vpz_decoder_create(vpz_context *c, const vpz_stream_config *cfg, int32_t n_streams, vpz_decoder **out)
{
    if (!c || !cfg || !out || n_streams <= 0) return VPZ_E_INVALID_ARG;
    *out = nullptr;
    Context *ctx = &c->impl;
    if (cfg->channels < 1 || cfg->channels > VPZ_MAX_CHANNELS)
        return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_create: channels out of range");
    if (cfg->block_size0 > cfg->block_size1)
        return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_create: block_size0 > block_size1");
    for (int bs : {cfg->block_size0, cfg->block_size1})
        if (bs < 64 || bs > 8192 || (bs & (bs - 1)) != 0)
            return set_error(ctx, VPZ_E_UNSUPPORTED, "vpz_decoder_create: block sizes must be powers of two in [64, 8192]");
    if (cfg->floor_count < 0 || cfg->mapping_count < 0 || (cfg->floor_count && !cfg->floors && !cfg->floor_types) ||
        (cfg->mapping_count && !cfg->mappings) || cfg->mapping_count > 256 || cfg->floor_count > 64)
        return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_create: bad floor / mapping tables");
    VPZ_HIP_TRY(ctx, hipSetDevice(ctx->device));

    vpz_decoder *d = new (std::nothrow) vpz_decoder();
    if (!d) return VPZ_E_NOMEM;
    Decoder &D = d->impl;
    D.ctx = ctx;
    D.channels = cfg->channels;
    D.size0 = cfg->block_size0;
    D.size1 = cfg->block_size1;
    D.clip = cfg->clip_samples ? 1 : 0;
    D.n_streams = n_streams;
    {
        const char *no_big = getenv("VPZ_NO_BIG");  // A/B and bit-equality tests: the three-pass path for 4096 / 8192 blocks
        D.big = synth_big_supported(cfg->block_size0, cfg->block_size1) && !(no_big && atoi(no_big));
    }
    D.generic = !synth_supports_sizes(cfg->block_size0, cfg->block_size1) && !D.big;
    D.states.assign(n_streams, StreamState());
    D.floors.assign((size_t)cfg->floor_count, vpz_floor1_config{});
    D.floor_types.assign((size_t)cfg->floor_count, 1);
    D.floors0.assign((size_t)cfg->floor_count, vpz_floor0_config{});
    for (int i = 0; i < cfg->floor_count; ++i) {
        if (cfg->floor_types) D.floor_types[i] = cfg->floor_types[i];
        if (D.floor_types[i] == 1 && cfg->floors) D.floors[i] = cfg->floors[i];
        if (D.floor_types[i] == 0 && cfg->floors0) D.floors0[i] = cfg->floors0[i];
    }
    D.mappings.assign(cfg->mappings, cfg->mappings + cfg->mapping_count);
    static_assert(VPZ_PKT_BLOCK_FLAG == 1 && VPZ_PKT_PREV_FLAG == 2 && VPZ_PKT_NEXT_FLAG == 4, "packet_info index");
    for (int f = 0; f < 8; ++f) D.packet_info[f] = get_packet_info(D.size0, D.size1, f & 1, f & 2, f & 4);
    if (const char *e = getenv("VPZ_RUN_LENGTH")) D.run_length_override = atoi(e);
    if (const char *e = getenv("VPZ_NO_DIRECT_I16")) D.no_direct_i16 = atoi(e) != 0;  // A/B and bit-equality tests: always widen int16 residue first
    if (const char *e = getenv("VPZ_NO_RUN_INLINE")) D.no_run_inline = atoi(e) != 0;  // A/B tests: flag bytes from cflags / cmap only
    if (const char *e = getenv("VPZ_NO_CHAIN")) D.no_chain = atoi(e) != 0;  // A/B tests: every run recomputes its predecessor block
    if (const char *e = getenv("VPZ_DUAL_RUN")) D.dual_run = std::max(4, atoi(e));
    if (const char *e = getenv("VPZ_SYNTH_ABLATE")) D.ablate = atoi(e);
    if (const char *e = getenv("VPZ_NO_BATCH")) {  // A/B tests: one short block per pass, runs of equal length
        if (atoi(e)) D.ablate |= 128;
    }
    if (const char *e = getenv("VPZ_HOST_THREADS")) D.host_threads = atoi(e);
    if (const char *e = getenv("VPZ_NO_EARLY_UPLOAD")) D.no_early_upload = atoi(e) != 0;
    if (const char *e = getenv("VPZ_PAR_MIN_PACKETS")) D.par_min_packets = atoll(e);

    int rc = VPZ_OK;
    std::vector<FloorDev> fdev(std::max<size_t>(1, D.floors.size()));
    for (size_t i = 0; i < D.floors.size() && rc == VPZ_OK; ++i) {
        if (D.floor_types[i] == 1) {
            rc = build_floor(D.floors[i], &fdev[i]);
        } else if (D.floor_types[i] == 0) {  // Floor0.cs:50-51
            const vpz_floor0_config &f0 = D.floors0[i];
            if (f0.order < 1 || f0.order > 255 || f0.rate < 1 || f0.bark_map_size < 1 || f0.amp_bits < 0 || f0.amp_bits > 63)
                rc = VPZ_E_INVALID_ARG;
        } else {
            rc = VPZ_E_INVALID_ARG;
        }
    }
    std::vector<uint8_t> steps;
    for (size_t m = 0; m < D.mappings.size() && rc == VPZ_OK; ++m) {
        const vpz_mapping_config &mc = D.mappings[m];
        if (mc.coupling_steps < 0 || mc.coupling_steps > VPZ_MAX_COUPLING) { rc = VPZ_E_INVALID_ARG; break; }
        while (mc.coupling_steps && steps.size() % 8) steps.push_back(0);  // (group mode reads 8 bytes of steps at once)
        D.mapping_steps_off.push_back(mc.coupling_steps ? (int32_t)steps.size() : -1);
        for (int i = 0; i < mc.coupling_steps; ++i) {
            const int mag = mc.coupling_magnitude[i], ang = mc.coupling_angle[i];
            if (mag == ang || mag >= D.channels || ang >= D.channels) { rc = VPZ_E_INVALID_ARG; break; }  // Mapping.cs:41
            steps.push_back((uint8_t)mag);
            steps.push_back((uint8_t)ang);
        }
        for (int ch = 0; ch < D.channels && rc == VPZ_OK; ++ch)
            if (D.floors.size() && mc.channel_floor[ch] >= D.floors.size()) rc = VPZ_E_INVALID_ARG;
        D.max_steps = std::max(D.max_steps, (int)mc.coupling_steps);
        // the residue's support (ABI v4; Residue0.cs:122-125): whole point groups of blocksize/16 bins beyond residue_end
        // are skipped.  (residue_begin is validated and kept for the record: real streams begin at bin 0.)
        for (int b = 0; b < 2 && rc == VPZ_OK; ++b) {
            const int half = (b ? D.size1 : D.size0) / 2;
            const int begin = mc.residue_begin[b], end_raw = mc.residue_end[b];
            if (begin < 0 || end_raw < 0 || (end_raw != 0 && begin > end_raw)) { rc = VPZ_E_INVALID_ARG; break; }
            const int end = end_raw == 0 ? half : std::min(end_raw, half);
            const int per_group = std::max(1, half / 8);
            const int groups = std::min(8, (end + per_group - 1) / per_group);
            D.mapping_skip[b].push_back((uint8_t)(half >= 8 ? 8 - groups : 0));
        }
    }
    if (const char *e = getenv("VPZ_NO_SUPPORT"))  // A/B tests: ignore the declared support (load and multiply the zeros)
        if (atoi(e))
            for (int b = 0; b < 2; ++b) std::fill(D.mapping_skip[b].begin(), D.mapping_skip[b].end(), (uint8_t)0);
    // group mode applies a mapping's steps in reverse order with a workgroup barrier only where a step touches a
    // channel an earlier step of the same LEVEL touched: mark those steps, count the levels
    std::vector<uint8_t> steps_lvl = steps;
    int max_levels = 0;
    for (size_t m = 0; m < D.mappings.size() && rc == VPZ_OK; ++m) {
        const int n = D.mappings[m].coupling_steps, off = D.mapping_steps_off[m];
        int levels = n > 0 ? 1 : 0;
        uint32_t used[8] = {};
        for (int i = n - 1; i >= 0; --i) {
            const uint8_t mag = steps[off + 2 * i], ang = steps[off + 2 * i + 1];
            const bool clash = (used[mag >> 5] >> (mag & 31) & 1) || (used[ang >> 5] >> (ang & 31) & 1);
            if (clash) {
                ++levels;
                memset(used, 0, sizeof used);
                if (mag < 128) steps_lvl[off + 2 * i] |= 0x80;
            }
            used[mag >> 5] |= 1u << (mag & 31);
            used[ang >> 5] |= 1u << (ang & 31);
        }
        max_levels = std::max(max_levels, levels);
    }
    D.n_step_pairs = (int)(steps.size() / 2);
    {
        bool has_floor0 = false;
        for (uint8_t t : D.floor_types) has_floor0 |= (t == 0);
        for (uint8_t t : D.floor_types) D.has_floor1 |= (t != 0);
        const char *no_group = getenv("VPZ_NO_GROUP");  // tuning / A-B tests: force the separate coupling pass
        D.group_ok = synth_group_supported(D.channels) && !D.generic && !D.big && !has_floor0 && D.max_steps <= 255 &&
                     D.n_step_pairs <= kGroupMaxStepPairs && !(no_group && atoi(no_group));
        const char *no_dual = getenv("VPZ_NO_DUAL");
        // (type-0 floors ride in the stereo fast path when their bark maps fit a wave's LDS row; VPZ_NO_F0_FUSED=1: the old route)
        const char *no_f0 = getenv("VPZ_NO_F0_FUSED");
        D.f0_fused = has_floor0 && !(no_f0 && atoi(no_f0));
        for (size_t i = 0; i < D.floors0.size() && rc == VPZ_OK; ++i)
            if (D.floor_types[i] == 0) {
                if (D.floors0[i].bark_map_size < 1 || D.floors0[i].bark_map_size > kFloor0MaxBark) D.f0_fused = false;
                D.f0_k = std::max(D.f0_k, (int)D.floors0[i].bark_map_size);
            }
        D.f0_k = (D.f0_k + 255) & ~255;  // (rows of whole 256-value rounds: floor0_curve_kernel stores a round without asking)
        D.dual_ok = synth_dual_supported(D.channels, D.size0, D.size1) && !D.generic && (!has_floor0 || D.f0_fused) && D.max_steps <= 255 &&
                    D.n_step_pairs <= kGroupMaxStepPairs && !(no_dual && atoi(no_dual));
        D.f0_fused = D.f0_fused && D.dual_ok;
        // every channel in at most one step of its mapping (then a mapping has one level, and a wave can apply its own step
        // to the pair of values it reads): the packet may stay interleaved in LDS
        bool single_step = true;
        for (size_t m = 0; m < D.mappings.size() && rc == VPZ_OK; ++m) {  // (the step tables are complete only then)
            uint32_t seen[8] = {};
            const int n = D.mappings[m].coupling_steps, off = D.mapping_steps_off[m];
            for (int i = 0; i < n; ++i)
                for (int k = 0; k < 2; ++k) {
                    const uint8_t c = steps[off + 2 * i + k];
                    if (seen[c >> 5] >> (c & 31) & 1) single_step = false;
                    seen[c >> 5] |= 1u << (c & 31);
                }
        }
        const char *want_dma = getenv("VPZ_GROUP_DMA");  // measured slower than staging through registers: opt-in
        D.group_dma = D.group_ok && single_step && (D.channels & 1) == 0 && D.max_steps <= 4 && want_dma && atoi(want_dma);
        D.max_steps = max_levels;  // from here on: the barriers a frame's coupling needs in group mode
        const char *nc = getenv("VPZ_NO_COMPACT");
        D.no_compact = nc && atoi(nc);
        if (const char *z = getenv("VPZ_ZERO_COPY_MAX")) D.zero_copy_max = (size_t)atoll(z);
    }
    if (rc != VPZ_OK) {  // (before anything below indexes the floor / step tables with values that failed validation)
        delete d;
        return set_error(ctx, rc, "vpz_decoder_create: invalid floor1 / mapping configuration");
    }
    std::vector<uint32_t> map_bits(std::max<size_t>(1, D.mappings.size()), 0u);
    D.mapping_uses_floor0.assign(D.mappings.size(), 0);
    for (size_t m = 0; m < D.mappings.size(); ++m) {
        const int n = D.mappings[m].coupling_steps;
        if ((D.group_ok || D.dual_ok) && n > 0)
            map_bits[m] = ((uint32_t)n << kFrameStepsShift) |
                          ((uint32_t)(D.mapping_steps_off[m] / 2) << kFrameStepsOffShift);
        map_bits[m] |= ((uint32_t)D.mapping_skip[1][m] << kFrameSkipShift) | ((uint32_t)D.mapping_skip[0][m] << kMapSkipShortShift);
        for (int ch = 0; ch < D.channels; ++ch)
            if (!D.floor_types.empty() && D.floor_types[D.mappings[m].channel_floor[ch]] == 0) D.mapping_uses_floor0[m] = 1;
    }
    // The pair route: do the coupling steps of ALL mappings join the channels two by two (a channel has at most one partner, the
    // same in every mapping)?  Then the coupled pairs and, two by two in channel order, the uncoupled channels are the pairs;
    // every (pair, mapping) gets its own step list -- the mapping's steps between the pair's channels, in the mapping's order.
    std::vector<uint8_t> pair_ch, pair_steps;
    std::vector<uint32_t> pair_map_bits;
    {
        // (VPZ_NO_GROUP=1 asks for the separate coupling pass for everything with more than two channels: no pairs either)
        const char *no_dual = getenv("VPZ_NO_DUAL"), *no_pairs = getenv("VPZ_NO_PAIRS"), *no_group = getenv("VPZ_NO_GROUP");
        bool has_floor0 = false;
        for (uint8_t t : D.floor_types) has_floor0 |= (t == 0);
        const int C = D.channels;
        bool ok = synth_pairs_supported(C, D.size0, D.size1) && !D.generic && !has_floor0 && !(no_dual && atoi(no_dual)) &&
                  !(no_pairs && atoi(no_pairs)) && !(no_group && atoi(no_group)) && D.mappings.size() <= 255;
        std::vector<int> partner((size_t)std::max(C, 1), -1);
        for (size_t m = 0; m < D.mappings.size() && ok; ++m) {
            const int n = D.mappings[m].coupling_steps, off = D.mapping_steps_off[m];
            for (int i = 0; i < n && ok; ++i) {
                const int mag = steps[off + 2 * i], ang = steps[off + 2 * i + 1];
                if (mag >= C || ang >= C || mag == ang) ok = false;
                else if (partner[mag] < 0 && partner[ang] < 0) { partner[mag] = ang; partner[ang] = mag; }
                else if (partner[mag] != ang || partner[ang] != mag) ok = false;
            }
        }
        if (ok) {
            int lone = -1;
            for (int c = 0; c < C; ++c) {
                if (partner[c] > c) { pair_ch.push_back((uint8_t)c); pair_ch.push_back((uint8_t)partner[c]); }
                else if (partner[c] < 0) {
                    if (lone < 0) lone = c;
                    else { pair_ch.push_back((uint8_t)lone); pair_ch.push_back((uint8_t)c); lone = -1; }
                }
            }
            ok = lone < 0 && (int)pair_ch.size() == C;
        }
        if (ok) {
            const size_t nm = std::max<size_t>(1, D.mappings.size());
            const int n_pairs = C / 2;
            pair_map_bits.assign((size_t)n_pairs * nm, 0u);
            for (int p = 0; p < n_pairs && ok; ++p)
                for (size_t m = 0; m < D.mappings.size() && ok; ++m) {
                    const int n = D.mappings[m].coupling_steps, off = D.mapping_steps_off[m];
                    const int a = pair_ch[2 * p], b = pair_ch[2 * p + 1];
                    const size_t first = pair_steps.size() / 2;
                    int cnt = 0;
                    for (int i = 0; i < n; ++i) {
                        const int mag = steps[off + 2 * i];
                        if (mag != a && mag != b) continue;
                        pair_steps.push_back(mag == a ? 0 : 1);
                        pair_steps.push_back(mag == a ? 1 : 0);
                        ++cnt;
                    }
                    if (cnt > 255 || first > 255) ok = false;
                    uint32_t w = ((uint32_t)D.mapping_skip[1][m] << kFrameSkipShift) | ((uint32_t)D.mapping_skip[0][m] << kMapSkipShortShift);
                    if (cnt > 0) w |= ((uint32_t)cnt << kFrameStepsShift) | ((uint32_t)first << kFrameStepsOffShift);
                    pair_map_bits[(size_t)p * nm + m] = w;
                }
            ok = ok && pair_steps.size() / 2 <= (size_t)kGroupMaxStepPairs;
        }
        D.pairs = ok;
        if (const char *e = getenv("VPZ_PAIRS")) D.pairs_always = atoi(e) != 0;
        if (ok) {
            D.dual_ok = true;
            D.n_pair_step_pairs = (int)(pair_steps.size() / 2);
        }
    }
    if ((rc = get_tables(ctx, D.size0, &D.t0)) != VPZ_OK || (rc = get_tables(ctx, D.size1, &D.t1)) != VPZ_OK) {
        delete d;
        return rc;
    }
    hipError_t e = hipSuccess;
    const size_t state_bytes = 2 * sizeof(float) * (size_t)n_streams * D.channels * (D.size1 / 2);  // two copies, see StreamState
    if (!ctx->d_inv_db) {
        // (+ 64 floats of +0.0 behind the table: where the pair route sends the loads of bins beyond a residue's declared support)
        e = hipMalloc((void **)&ctx->d_inv_db, (256 + 64) * sizeof(float));
        if (e == hipSuccess) e = hipMemset(ctx->d_inv_db, 0, (256 + 64) * sizeof(float));
        if (e == hipSuccess)
            e = hipMemcpy(ctx->d_inv_db, k_inverse_db_bits, 256 * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&D.d_floors, sizeof(FloorDev) * fdev.size());
    if (e == hipSuccess) e = hipMemcpy(D.d_floors, fdev.data(), sizeof(FloorDev) * fdev.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&D.d_state_h, state_bytes);
    if (e == hipSuccess) e = hipMemset(D.d_state_h, 0, state_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&D.d_clipped, sizeof(int32_t) * (size_t)n_streams);
    if (e == hipSuccess) e = hipMemset(D.d_clipped, 0, sizeof(int32_t) * (size_t)n_streams);
    if (e == hipSuccess) e = hipMalloc((void **)&D.d_steps, steps.size() ? steps.size() : 1);
    if (e == hipSuccess && !steps.empty()) e = hipMemcpy(D.d_steps, steps.data(), steps.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&D.d_map_bits, sizeof(uint32_t) * map_bits.size());
    if (e == hipSuccess) e = hipMemcpy(D.d_map_bits, map_bits.data(), sizeof(uint32_t) * map_bits.size(), hipMemcpyHostToDevice);
    if (D.pairs) {
        if (e == hipSuccess) e = hipMalloc((void **)&D.d_pair_ch, pair_ch.size());
        if (e == hipSuccess) e = hipMemcpy(D.d_pair_ch, pair_ch.data(), pair_ch.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void **)&D.d_pair_map_bits, sizeof(uint32_t) * pair_map_bits.size());
        if (e == hipSuccess)
            e = hipMemcpy(D.d_pair_map_bits, pair_map_bits.data(), sizeof(uint32_t) * pair_map_bits.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void **)&D.d_pair_steps, pair_steps.size() + 8);
        if (e == hipSuccess && !pair_steps.empty())
            e = hipMemcpy(D.d_pair_steps, pair_steps.data(), pair_steps.size(), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&D.d_steps_lvl, steps_lvl.size() ? steps_lvl.size() : 1);
    if (e == hipSuccess && !steps_lvl.empty())
        e = hipMemcpy(D.d_steps_lvl, steps_lvl.data(), steps_lvl.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        // Floor0 ctor tables (Floor0.cs:82-95): one bark map per block size, n+1 ints, last bin left at 0
        std::vector<int32_t> maps;
        std::vector<uint16_t> f0_bark;
        std::vector<char> devs(floor0_dev_size() * std::max<size_t>(1, D.floors0.size()), 0);
        bool any0 = false;
        for (size_t i = 0; i < D.floors0.size(); ++i) {
            if (D.floor_types[i] != 0) continue;
            any0 = true;
            const vpz_floor0_config &f0 = D.floors0[i];
            int64_t off[2];
            const int halves[2] = {D.size0 / 2, D.size1 / 2};
            for (int b = 0; b < 2; ++b) {
                const int n = halves[b];
                off[b] = (int64_t)maps.size();
                auto to_bark = [](double lsp) -> float {
                    return (float)(13.1 * atan(0.00074 * lsp) + 2.24 * atan(0.0000000185 * lsp * lsp) + .0001 * lsp);
                };
                const float scale = (float)f0.bark_map_size / to_bark(f0.rate / 2.0);
                std::vector<int32_t> m((size_t)n + 1, 0);
                for (int k = 0; k < n + 1 - 2; ++k) {
                    const int v = (int)floor((double)(to_bark((f0.rate / 2.0) / n * k) * scale));
                    m[k] = std::min(f0.bark_map_size - 1, v);
                }
                m[n] = -1;
                maps.insert(maps.end(), m.begin(), m.end());
            }
            fill_floor0_dev(devs.data() + i * floor0_dev_size(), f0.order, f0.bark_map_size, f0.amp_ofs, off[0], off[1]);
            if (D.f0_fused) {  // the same maps in the order a lane of the stereo fast path holds its bins (SynthArgs.f0_bark)
                if (f0_bark.size() < (D.floors0.size() * 2) * 1024) f0_bark.assign((D.floors0.size() * 2) * 1024, 0);
                for (int b = 0; b < 2; ++b) {
                    const int n = halves[b], lpb = n / 16;
                    if (n != 128 && n != 1024) continue;
                    const int32_t *mp = maps.data() + off[b];
                    uint16_t *T = f0_bark.data() + (i * 2 + (size_t)b) * 1024;
                    for (int l = 0; l < lpb; ++l)
                        for (int mm = 0; mm < 8; ++mm)
                            for (int e2 = 0; e2 < 2; ++e2) {
                                const int v = mp[2 * (l + lpb * mm) + e2];
                                T[l * 16 + 2 * mm + e2] = (uint16_t)std::min(std::max(v, 0), D.f0_k - 1);
                            }
                }
            }
        }
        if (any0) {
            e = hipMalloc(&D.d_floors0, devs.size());
            if (e == hipSuccess) e = hipMemcpy(D.d_floors0, devs.data(), devs.size(), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc((void **)&D.d_bark_maps, sizeof(int32_t) * maps.size());
            if (e == hipSuccess) e = hipMemcpy(D.d_bark_maps, maps.data(), sizeof(int32_t) * maps.size(), hipMemcpyHostToDevice);
            if (e == hipSuccess && D.f0_fused && !f0_bark.empty()) {
                e = hipMalloc((void **)&D.d_f0_bark, sizeof(uint16_t) * f0_bark.size());
                if (e == hipSuccess) e = hipMemcpy(D.d_f0_bark, f0_bark.data(), sizeof(uint16_t) * f0_bark.size(), hipMemcpyHostToDevice);
                if (e == hipSuccess) e = hipMalloc((void **)&D.d_f0_w, sizeof(float) * D.floors0.size() * (size_t)D.f0_k);
                if (e == hipSuccess) e = launch_floor0_wtab(D.d_floors0, (int)D.floors0.size(), D.f0_k, D.d_f0_w, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            }
        }
    }
    if (e != hipSuccess) {
        vpz_decoder_destroy(d);
        return set_error(ctx, VPZ_E_NOMEM, "vpz_decoder_create: device allocation", e);
    }
    *out = d;
    return VPZ_OK;
}

void vpz_decoder_destroy(vpz_decoder *d)
{
    if (!d) return;
    Decoder &D = d->impl;
    if (D.ctx) {
        (void)hipSetDevice(D.ctx->device);
        (void)hipStreamSynchronize(D.ctx->stream);
    }
    if (D.d_f0_bark) (void)hipFree(D.d_f0_bark);
    if (D.d_f0_w) (void)hipFree(D.d_f0_w);
    DevBuf *bufs[] = {&D.b_f0curve, &D.b_in_amp, &D.b_in_coeff, &D.b_ybuf, &D.b_bigtail, &D.b_curve, &D.b_temp, &D.b_cposts, &D.b_ccount, &D.b_in_res, &D.b_in_res16, &D.b_in_posts,
                      &D.b_in_counts, &D.b_out, &D.arenas[0].dev, &D.arenas[1].dev};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (D.d_floors) (void)hipFree(D.d_floors);
    if (D.d_state_h) (void)hipFree(D.d_state_h);
    if (D.d_clipped) (void)hipFree(D.d_clipped);
    if (D.d_steps) (void)hipFree(D.d_steps);
    if (D.d_steps_lvl) (void)hipFree(D.d_steps_lvl);
    if (D.d_map_bits) (void)hipFree(D.d_map_bits);
    if (D.d_pair_ch) (void)hipFree(D.d_pair_ch);
    if (D.d_pair_map_bits) (void)hipFree(D.d_pair_map_bits);
    if (D.d_pair_steps) (void)hipFree(D.d_pair_steps);
    if (D.d_floors0) (void)hipFree(D.d_floors0);
    if (D.d_bark_maps) (void)hipFree(D.d_bark_maps);
    for (PinnedArena &A : D.arenas) {
        if (A.base) (void)hipHostFree(A.base);
        if (A.uploaded) (void)hipEventDestroy(A.uploaded);
    }
    delete d;
}

int vpz_decoder_reset(vpz_decoder *d, int32_t stream)
{
    if (!d) return VPZ_E_INVALID_ARG;
    Decoder &D = d->impl;
    if (stream >= D.n_streams) return set_error(D.ctx, VPZ_E_INVALID_ARG, "vpz_decoder_reset: bad stream");
    VPZ_HIP_TRY(D.ctx, hipSetDevice(D.ctx->device));
    const int lo = stream < 0 ? 0 : stream, hi = stream < 0 ? D.n_streams : stream + 1;
    for (int s = lo; s < hi; ++s) {  // StreamDecoder.cs:357-369: the position value itself is kept
        const int64_t pos = D.states[s].current_position;
        const int32_t epoch = D.states[s].clip_epoch;
        D.states[s] = StreamState();
        D.states[s].current_position = pos;
        D.states[s].has_position = false;
        // `_hasClipped = false`: the fused kernels record clipping as "the stream's epoch", so moving on to the next
        // epoch clears the flag without any device work; the any-block-size path sets a plain 1 and is cleared below
        D.states[s].clip_epoch = D.generic ? 1 : epoch + 1;
    }
    if (D.generic)
        VPZ_HIP_TRY(D.ctx, hipMemsetAsync(D.d_clipped + lo, 0, sizeof(int32_t) * (size_t)(hi - lo), D.ctx->stream));
    return VPZ_OK;
}

// One vpz_decoder_synth call, stage by stage.  Everything here is host work on integers; the sample
// arithmetic is in the kernels the last stage enqueues.
namespace {

struct SynthCall {
    // ---- the call's arguments
    Decoder &D;
    Context *ctx;
    const int64_t n_packets;
    const vpz_packet *packets;
    const float *residue;
    const int16_t *posts;
    const uint8_t *post_counts;
    const int mem_space;
    void *pcm_out;
    const int64_t *stream_out_offset;
    const int64_t stream_out_capacity;
    int64_t capacity_of(int s) const  // (vpz_decoder_set_stream_capacities tightens the call's bound per stream)
    {
        return D.stream_caps.empty() ? stream_out_capacity : std::min(stream_out_capacity, D.stream_caps[(size_t)s]);
    }
    const int out_layout;
    const int64_t channel_stride;
    // ---- derived
    const int C, half0, half1;
    const int64_t n_rec;
    const bool have_posts;
    const bool out_interleaved, out_s16;  // VPZ_OUT_* decomposed
    const size_t out_elem;                // bytes per PCM sample
    PinnedArena *A = nullptr;
    std::vector<StreamState> st;  // working copy of the stream states: committed when the batch is accepted
    std::vector<uint8_t> started_with_prev, started_prev_long;
    // ---- products of the state machine
    FrameDesc *frames = nullptr;
    size_t n_frames = 0;
    uint8_t *rec_floor = nullptr;  // per channel record: floor index | 0x40 type-0 | 0x80 long block
    bool any_floor = false, any_floor0 = false, need_coupling = false;
    bool any_short = false;       // the batch holds short blocks (run cutting by cost only pays then)
    bool group_align_ok = true;   // every interleaved packet starts on a 16-byte boundary (group mode loads 16 bytes)
    bool use_group = false;       // decided after pass 1: synth_kernel's group mode instead of the coupling pass
    bool use_dual = false;        // ... or the stereo fast path (synth_dual.hip), which takes precedence
    bool cut_by_cost = false;     // cut_runs balanced the runs by cost (short blocks ride in batches): only then do the kernels
                                  // batch -- in runs of equal LENGTH the ones rich in short blocks would be done early and the
                                  // launch would wait for the others (configs[2]: 0.205 ms without batches, 0.211 with)
    bool ilv_seen = false, planar_seen = false;  // layouts of the packets that become frames (the dual kernel wants one)
    bool align2_ok = true;        // every packet starts on an 8-byte boundary (planar packets are read 8 bytes at a time)
    bool compact = false;         // every run compact: two bytes per frame instead of a FrameDesc (parallel pass only)
    uint8_t *cflags = nullptr, *cmap = nullptr;
    uint8_t *run_inline = nullptr;  // [run][32]: the flag bytes of a run's first 16 staged frames (SynthArgs.run_inline)
    int64_t mismatches = 0, res_extent = 0;
    // ---- descriptor tables (pinned arena; dev() gives the device mirror's address)
    RunDesc *runs = nullptr;
    size_t n_runs = 0, runs_cap = 0, runs_arena_mark = (size_t)-1;
    uint8_t *cpk = nullptr;
    int n_cpk = 0;
    int64_t temp_floats = 0;
    uint8_t *f0recs = nullptr;
    int n_f0 = 0;
    int64_t *offs = nullptr;
    GenericFrame *gf = nullptr;
    int64_t *src0 = nullptr, *dst0 = nullptr, *src1 = nullptr, *dst1 = nullptr;
    int32_t *save_list = nullptr;
    size_t n0 = 0, n1 = 0, n_save = 0;
    int64_t y_floats = 0;
    // ---- device views
    const float *d_res = nullptr, *d_amp = nullptr, *d_coeff = nullptr;
    const int16_t *d_posts = nullptr;
    const uint8_t *d_counts = nullptr;
    void *d_out = nullptr;
    bool early_residue = false, early_posts = false;  // stage_inputs_early has the copies under way
    bool spec_i16 = false;        // the synth kernel reads the 16-bit residue in place (SynthArgs.spec_i16)
    int64_t early_res_extent = 0;

    SynthCall(Decoder &dec, int64_t n, const vpz_packet *pk, const float *res, const int16_t *po, const uint8_t *pc,
              int mem, void *out, const int64_t *out_off, int64_t out_cap, int layout, int64_t stride)
        : D(dec), ctx(dec.ctx), n_packets(n), packets(pk), residue(res), posts(po), post_counts(pc), mem_space(mem),
          pcm_out(out), stream_out_offset(out_off), stream_out_capacity(out_cap), out_layout(layout),
          channel_stride(stride), C(dec.channels), half0(dec.size0 / 2), half1(dec.size1 / 2),
          n_rec(n * dec.channels), have_posts(po && pc && !dec.floors.empty()),
          out_interleaved(layout == VPZ_OUT_INTERLEAVED || layout == VPZ_OUT_INTERLEAVED_S16),
          out_s16(layout == VPZ_OUT_INTERLEAVED_S16 || layout == VPZ_OUT_PLANAR_S16),
          out_elem(out_s16 ? sizeof(int16_t) : sizeof(float))
    {
    }

    // device view of a table carved out of the arena: its copy in the device mirror, or -- for small arenas --
    // the pinned host memory itself, which the GPU reads over the link (see stage_inputs)
    void *dev(const void *host_ptr) const
    {
        if (!host_ptr) return nullptr;
        char *base = zero_copy ? A->mapped : static_cast<char *>(A->dev.p);
        return base + (static_cast<const char *>(host_ptr) - A->base);
    }
    // The stereo fast path takes a batch whose packets all have ONE input layout (its loads are unconditional: the
    // layout is a template parameter) and start where its loads are aligned: 16 bytes for the Residue2 vector, 8 for planar.
    bool dual_usable() const
    {
        if (!D.dual_ok || (any_floor0 && !D.f0_fused) || (ilv_seen && planar_seen)) return false;
        // (int16 values are widened into the decoder's own, aligned staging buffer whatever the memory space)
        // (the pair route reads 8 bytes -- two adjacent channels of a bin, or two bins of a channel -- whatever the layout)
        const bool wide = ilv_seen && !D.pairs;
        const bool dev_ok = mem_space == VPZ_MEM_HOST || D.residue_format == VPZ_RESIDUE_I16 ||
                            (reinterpret_cast<uintptr_t>(residue) & (wide ? 15 : 7)) == 0;
        if (!(dev_ok && (wide ? group_align_ok : align2_ok))) return false;
        // Pairs or group mode, where both can take the call (measured on BASELINE configs[3], 6 channels, profiles/r5_ab_pairs.txt):
        // planar packets to planar PCM the pairs are 8 % faster; the Residue2 vector read by columns (a third of every line per
        // workgroup) ties with group mode's staging; interleaved PCM written by columns loses 20 % against the packet's waves writing
        // whole rows together.  VPZ_PAIRS=1 (tests, A/B): the pairs wherever they can.
        // Beyond eight channels (no group mode) the columns of the Residue2 vector cost what the separate coupling pass costs
        // (10 channels: 0.390 against 0.375 ms): the pairs take planar packets to planar PCM there too.
        if (D.pairs && !D.pairs_always && (ilv_seen || out_interleaved)) return false;
        return true;
    }
    bool group_usable() const
    {
        return D.group_ok && group_align_ok &&
               (mem_space == VPZ_MEM_HOST || D.residue_format == VPZ_RESIDUE_I16 || (reinterpret_cast<uintptr_t>(residue) & 15) == 0);
    }
    bool zero_copy = false;
    bool host_failed = false;  // a share of a fork-join threw (allocation): the call returns VPZ_E_NOMEM

    // Every per-call table (frame / run descriptors, coupling packets, per-record floor info, output
    // offsets, ...) is carved out of ONE pinned arena that goes to its device mirror in a single copy;
    // two arenas alternate so the host can prepare call k+1 while call k's upload is still queued.
    int open_arena()
    {
        bool has_floor0_type = false;
        for (uint8_t t : D.floor_types) has_floor0_type |= (t == 0);
        const size_t np = (size_t)n_packets;
        // (runs hold >= 3 frames unless VPZ_RUN_LENGTH says otherwise: see cut_runs)
        size_t need = (sizeof(FrameDesc) + (D.run_length_override > 0 ? sizeof(RunDesc) + 32 : sizeof(RunDesc) / 2 + 16) + 2 +
                       coupling_packet_size()) * np +
                      sizeof(RunDesc) * ((size_t)D.n_streams + 1 + 4 * 64) + (have_posts ? (size_t)n_rec : 0) +  // (+ the pieces of long streams)
                      sizeof(int64_t) * (size_t)D.n_streams + 4096;
        if (has_floor0_type) need += floor0_rec_size() * (size_t)n_rec + 64;
        if (D.generic)
            need += (sizeof(GenericFrame) + 4 * sizeof(int64_t) * (size_t)C) * np +
                    sizeof(int32_t) * ((size_t)D.n_streams + 1) + 1024;
        D.arena_idx ^= 1;
        A = &D.arenas[D.arena_idx];
        int rc = arena_begin(ctx, *A, need);
        if (rc != VPZ_OK) return rc;
        return grow(ctx, A->dev, A->cap);
    }

    // flag bits of a decoded packet's frame: block / window selection, and what group mode needs to know about the
    // packet's input (layout, coupling steps of its mapping)
    uint32_t frame_flags(const vpz_packet &pk, const PacketInfo &pi) const
    {
        const bool bf = pk.flags & VPZ_PKT_BLOCK_FLAG, no_floor = pk.flags & VPZ_PKT_NO_FLOOR;
        uint32_t f = (bf ? kFrameLong : 0u) | (pi.left_use_size1 ? kFrameSlope1 : 0u) | (no_floor ? kFrameNoFloor : 0u);
        if (D.group_ok || D.dual_ok) {
            const int steps = no_floor ? 0 : D.mappings[pk.mapping].coupling_steps;
            if (pk.flags & VPZ_PKT_INTERLEAVED) f |= kFrameInterleaved;
            if (steps > 0)
                f |= ((uint32_t)steps << kFrameStepsShift) |
                     ((uint32_t)(D.mapping_steps_off[pk.mapping] / 2) << kFrameStepsOffShift);
        }
        if (!no_floor) f |= (uint32_t)D.mapping_skip[bf ? 1 : 0][pk.mapping] << kFrameSkipShift;
        return f;
    }

    // Pass 1 for large batches, split over the host cores.  It takes the batches real hosts produce: packets sorted by
    // stream, all decoded, no resync, and the only packet of a stream that may carry an EOS flag or fail the window
    // check (StreamDecoder.cs:777-778) is the stream's LAST one in the batch.  Then a packet's frame depends only on
    // its own flags and on the packet before it (its window geometry); output offsets are a prefix sum -- chunk-local
    // sums first, the chunks' bases serially, the descriptors in a second sweep -- and everything that needs the
    // stream position (granule pick-up :459-463, EOS trim :658-666) is settled once per stream afterwards.  Any other
    // batch returns 0 and the serial state machine below runs instead, so the rare paths of ReadNextPacket live in one
    // place.  Returns 1: done, 0: not applicable, < 0: error.
    int run_state_machine_parallel(int64_t *samples_written)
    {
        if (n_packets < D.par_min_packets) return 0;
        int parties = D.host_threads;
        if (parties <= 0) {
            cpu_set_t set;
            parties = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
            // (one process per GPU: the node's cores are shared with the other ranks torchrun started here)
            if (const char *lw = getenv("LOCAL_WORLD_SIZE")) parties /= std::max(1, atoi(lw));
            parties = std::max(1, std::min(parties, 16));
        }
        if (parties < 2) return 0;
        if (!ctx->host_pool || static_cast<HostPool *>(ctx->host_pool)->parties() != parties) {
            if (ctx->host_pool) ctx->host_pool_free(ctx->host_pool);
            static const int spin_us = getenv("VPZ_HOST_SPIN_US") ? atoi(getenv("VPZ_HOST_SPIN_US")) : 50;
            ctx->host_pool = new HostPool(parties, spin_us);
            ctx->host_pool_free = [](void *p) { delete static_cast<HostPool *>(p); };
        }
        HostPool &pool = *static_cast<HostPool *>(ctx->host_pool);
        constexpr int64_t kNone = INT64_MAX;

        struct Chunk {
            int64_t lo = 0, hi = 0;
            bool ok = true;
            int64_t lead_sum = 0;     // samples of the packets that continue the previous chunk's last stream
            int64_t lead_end = 0;     // first packet that does not
            int64_t lead_anchor = kNone;  // first packet with a granule position among them
            int64_t tail_sum = 0;     // samples of the chunk's last stream segment
            int64_t base = 0;         // filled between the sweeps: samples of the leading stream before this chunk
            int64_t res_extent = 0;
            bool any_floor = false, any_floor0 = false, need_coupling = false, align_ok = true, any_short = false;
            bool ilv = false, planar = false, align2 = true;
            bool dense = true;        // every packet's residue starts where its predecessor's (same stream) ends
            char pad[64];
        };
        std::vector<Chunk> chunks((size_t)parties);
        const int64_t per = (n_packets + parties - 1) / parties;
        for (int c = 0; c < parties; ++c) {
            chunks[c].lo = std::min<int64_t>(n_packets, per * c);
            chunks[c].hi = std::min<int64_t>(n_packets, per * (c + 1));
        }
        int32_t *psamples = D.packet_samples.data();
        std::vector<int64_t> &s_base = D.s_base, &s_cnt = D.s_cnt, &out_count = D.out_count;
        s_base.assign((size_t)D.n_streams + 1, 0);
        s_cnt.assign((size_t)D.n_streams, 0);
        out_count.assign((size_t)D.n_streams, 0);
        D.anchor_pkt.assign((size_t)D.n_streams, kNone);  // per stream: first packet that carries a granule position

        // what precedes packet p in its stream: the packet before it, or the stream's saved state
        auto prev_of = [&](int64_t p, bool &has_prev, int &prev_end, int &prev_stop) {
            const vpz_packet &pk = packets[p];
            if (p > 0 && packets[p - 1].stream == pk.stream) {
                const PacketInfo &ppi = D.packet_info[packets[p - 1].flags & 7];
                has_prev = true;
                prev_end = ppi.right_start;
                prev_stop = ppi.right_end;
            } else {
                const StreamState &S = D.states[pk.stream];
                has_prev = S.has_prev;
                prev_end = S.prev_end;
                prev_stop = S.prev_stop;
            }
        };
        auto is_last_of_stream = [&](int64_t p) { return p + 1 == n_packets || packets[p + 1].stream != packets[p].stream; };
        auto mismatch_at = [&](int64_t p) {
            bool has_prev;
            int prev_end, prev_stop;
            prev_of(p, has_prev, prev_end, prev_stop);
            const PacketInfo &pi = D.packet_info[packets[p].flags & 7];
            return has_prev && prev_stop - prev_end > (pi.left_use_size1 ? half1 : half0);
        };

        // sweep A: validation, samples per packet (before any EOS trim), chunk-local sums
        const bool ok_a = pool.run([&](int c) {
            Chunk &K = chunks[c];
            int64_t run = 0;
            bool leading = true;
            K.lead_end = K.lo;
            for (int64_t p = K.lo; p < K.hi; ++p) {
                const vpz_packet &pk = packets[p];
                if (pk.stream < 0 || pk.stream >= D.n_streams || (pk.flags & (VPZ_PKT_NOT_DECODED | VPZ_PKT_RESYNC)) ||
                    pk.residue_offset < 0 || (p > 0 && packets[p - 1].stream > pk.stream)) {
                    K.ok = false;
                    return;
                }
                const bool no_floor = pk.flags & VPZ_PKT_NO_FLOOR;
                if (!no_floor && (pk.mapping >= D.mappings.size() || !have_posts)) { K.ok = false; return; }
                const bool last = is_last_of_stream(p);
                if ((pk.flags & VPZ_PKT_EOS) && !last) { K.ok = false; return; }
                const bool new_stream = p == 0 || packets[p - 1].stream != pk.stream;
                if (new_stream) {
                    if (D.states[pk.stream].eos_found) { K.ok = false; return; }  // Read() ignores the stream from here on
                    if (leading) { K.lead_sum = run; K.lead_end = p; leading = false; }
                    run = 0;
                }
                bool has_prev;
                int prev_end, prev_stop;
                prev_of(p, has_prev, prev_end, prev_stop);
                const PacketInfo &pi = D.packet_info[pk.flags & 7];
                int cnt = 0;
                bool skipped = false;
                if (has_prev) {
                    if (prev_stop - prev_end > (pi.left_use_size1 ? half1 : half0)) {  // window mismatch
                        if (!last) { K.ok = false; return; }
                        skipped = true;
                    } else {
                        cnt = std::max(0, pi.right_start - pi.left_start);
                    }
                }
                psamples[p] = cnt;
                run += cnt;
                if (pk.granule != -1 && !skipped) {
                    if (leading) { if (K.lead_anchor == kNone) K.lead_anchor = p; }
                    else if (D.anchor_pkt[pk.stream] == kNone) D.anchor_pkt[pk.stream] = p;
                }
                const bool bf = pk.flags & VPZ_PKT_BLOCK_FLAG;
                if (!bf) K.any_short = true;
                K.res_extent = std::max(K.res_extent, pk.residue_offset + (int64_t)C * (bf ? half1 : half0));
                if (!no_floor) {
                    K.any_floor = true;
                    if (D.mappings[pk.mapping].coupling_steps > 0) K.need_coupling = true;
                    if (D.mapping_uses_floor0[pk.mapping]) K.any_floor0 = true;
                }
                if (pk.flags & VPZ_PKT_INTERLEAVED) { K.need_coupling = true; K.ilv = true; }
                else K.planar = true;
                if (pk.residue_offset & 3) K.align_ok = false;  // group mode reads every packet in 16-byte pieces
                if (pk.residue_offset & 1) K.align2 = false;
                if (!new_stream) {
                    const vpz_packet &pp = packets[p - 1];
                    const int64_t prev_floats = (int64_t)C * ((pp.flags & VPZ_PKT_BLOCK_FLAG) ? half1 : half0);
                    if (pk.residue_offset != pp.residue_offset + prev_floats) K.dense = false;
                }
            }
            if (leading) { K.lead_sum = run; K.lead_end = K.hi; }
            K.tail_sum = run;
        });
        if (!ok_a) return set_error(ctx, VPZ_E_NOMEM, "vpz_decoder_synth: host pass failed (allocation)");
        for (const Chunk &K : chunks)
            if (!K.ok) {  // the serial pass expects the per-packet counts zeroed
                std::fill(D.packet_samples.begin(), D.packet_samples.end(), 0);
                return 0;
            }
        bool all_dense = true;
        for (const Chunk &K : chunks) {
            all_dense &= K.dense;
            any_floor0 |= K.any_floor0;
            any_short |= K.any_short;
            need_coupling |= K.need_coupling;
            group_align_ok &= K.align_ok;
            align2_ok &= K.align2;
            ilv_seen |= K.ilv;
            planar_seen |= K.planar;
        }
        // Compact runs (two bytes per frame, descriptors built on the device) need consecutive packets with back to
        // back residues and a batch the fused kernel takes as it is (no planar temp, no type-0 floor pass)
        compact = all_dense && !D.generic && !D.big && (!any_floor0 || (D.f0_fused && dual_usable())) && !D.no_compact &&
                  (!need_coupling || group_usable() || dual_usable());
        if (compact) {
            cflags = arena_alloc<uint8_t>(*A, (size_t)n_packets);
            cmap = arena_alloc<uint8_t>(*A, (size_t)n_packets);
            D.out_off_scratch.resize((size_t)n_packets);
        } else {
            frames = arena_alloc<FrameDesc>(*A, (size_t)n_packets);
        }
        rec_floor = have_posts ? arena_alloc<uint8_t>(*A, (size_t)n_rec) : nullptr;
        st = D.states;
        D.trim_out_count.assign((size_t)D.n_streams, -1);
        D.trim_left_start.assign((size_t)D.n_streams, 0);
        started_with_prev.resize(D.n_streams);
        started_prev_long.resize(D.n_streams);
        for (int s = 0; s < D.n_streams; ++s) {
            started_with_prev[s] = st[s].has_prev;
            started_prev_long[s] = st[s].prev_long;
        }
        // bases of the chunks' leading segments; a leading segment's first granule packet belongs to the stream
        // unless an earlier chunk already found one
        {
            int32_t cur_stream = -1;
            int64_t cur_sum = 0;
            for (Chunk &K : chunks) {
                if (K.lo >= K.hi) continue;
                const int32_t first = packets[K.lo].stream, last = packets[K.hi - 1].stream;
                K.base = first == cur_stream ? cur_sum : 0;
                if (K.lead_anchor != kNone && D.anchor_pkt[first] > K.lead_anchor) D.anchor_pkt[first] = K.lead_anchor;
                if (K.lead_end == K.hi) cur_sum = K.base + K.lead_sum;  // one stream all through
                else cur_sum = K.tail_sum;
                cur_stream = last;
                res_extent = std::max(res_extent, K.res_extent);
                any_floor |= K.any_floor;
            }
        }
        if (D.generic) need_coupling = true;
        const bool group_bits = D.group_ok || D.dual_ok;  // (frame_flags' rule: these bits only when the decoder can use them)
        // sweep B: the descriptors, the per-record floor info, where each stream's packets begin and end
        const bool ok_b = pool.run([&](int c) {
            Chunk &K = chunks[c];
            int64_t run = K.base;
            for (int64_t p = K.lo; p < K.hi; ++p) {
                const vpz_packet &pk = packets[p];
                const bool new_stream = p == 0 || packets[p - 1].stream != pk.stream;
                if (new_stream) { run = 0; s_base[pk.stream] = p; }
                bool has_prev;
                int prev_end, prev_stop;
                prev_of(p, has_prev, prev_end, prev_stop);
                const PacketInfo &pi = D.packet_info[pk.flags & 7];
                const bool last = is_last_of_stream(p);
                const bool skipped = last && has_prev && prev_stop - prev_end > (pi.left_use_size1 ? half1 : half0);
                if (compact) {
                    uint8_t cf = (uint8_t)(pk.flags & 7);
                    if (pk.flags & VPZ_PKT_NO_FLOOR) cf |= kCfNoFloor;
                    if ((pk.flags & VPZ_PKT_INTERLEAVED) && group_bits) cf |= kCfInterleaved;
                    if (skipped) cf |= kCfSkip;  // window mismatch: a frame that does nothing
                    cflags[p] = cf;
                    cmap[p] = pk.mapping;
                    D.out_off_scratch[(size_t)p] = run;
                } else {
                    FrameDesc fd{};
                    fd.rec = (int32_t)(p * C);
                    if (skipped) {
                        fd.flags = kFrameDrain;  // skipped packet (window mismatch): a frame that does nothing
                    } else {
                        fd.flags = frame_flags(pk, pi);
                        if (has_prev) {
                            fd.packet_len = (uint16_t)(prev_stop - prev_end);
                            fd.prev_end = (uint16_t)prev_end;
                            fd.left_start = (uint16_t)pi.left_start;
                        } else {
                            fd.left_start = (uint16_t)pi.right_start;  // StreamDecoder.cs:679
                        }
                        fd.out_count = (uint16_t)psamples[p];
                        fd.spec_off = pk.residue_offset;
                    }
                    fd.out_off = run;
                    if (frame_is_steady(fd.flags, D.size1, fd.left_start, fd.packet_len, fd.prev_end, fd.out_count)) fd.flags |= kFrameSteady;
                    frames[p] = fd;
                }
                run += psamples[p];
                if (rec_floor) {
                    if (pk.flags & VPZ_PKT_NO_FLOOR) {
                        for (int ch = 0; ch < C; ++ch) rec_floor[(size_t)(p * C + ch)] = 0;
                    } else {
                        const vpz_mapping_config &mc = D.mappings[pk.mapping];
                        const uint8_t long_bit = (pk.flags & VPZ_PKT_BLOCK_FLAG) ? 0x80 : 0;
                        for (int ch = 0; ch < C; ++ch) {
                            const uint8_t fl = mc.channel_floor[ch];
                            const bool f0 = D.floor_types[fl] == 0;
                            rec_floor[(size_t)(p * C + ch)] = (uint8_t)(fl | long_bit | (f0 ? 0x40 : 0));
                        }
                    }
                }
                if (last) {
                    out_count[pk.stream] = run;
                    s_cnt[pk.stream] = p + 1;  // END of the stream's packets; becomes a count below (the stream's first
                                               // packet may belong to another chunk: no read of s_base here)
                }
            }
        });
        if (!ok_b) return set_error(ctx, VPZ_E_NOMEM, "vpz_decoder_synth: host pass failed (allocation)");
        // once per stream: its state after the batch (ReadNextPacket :640-694 for the last packet), the position
        // (:459-463, :493) and the EOS trim (:658-666)
        for (int s = 0; s < D.n_streams; ++s) {
            if (s_cnt[s] == 0) continue;
            const int64_t L = s_cnt[s] - 1;
            s_cnt[s] -= s_base[s];
            const vpz_packet &pk = packets[L];
            StreamState &S = st[s];
            const PacketInfo &pi = D.packet_info[pk.flags & 7];
            const bool eos = pk.flags & VPZ_PKT_EOS;
            if (eos) S.eos_found = true;
            bool has_prev;
            int prev_end, prev_stop;
            prev_of(L, has_prev, prev_end, prev_stop);
            const bool skipped = mismatch_at(L);
            auto out_off_of = [&](int64_t q) { return compact ? D.out_off_scratch[(size_t)q] : frames[q].out_off; };
            // position in front of the last packet: the stream's own count, re-based where a granule was picked up
            int64_t pos_base = S.current_position;
            bool has_pos = S.has_position;
            const int64_t anchor = D.anchor_pkt[s];
            if (!has_pos && anchor != kNone && anchor < L) {
                has_pos = true;
                pos_base = packets[anchor].granule - (out_off_of(anchor) + psamples[anchor]);
            }
            if (skipped) {
                ++mismatches;
                D.mismatch_packets.push_back(L);
                if (s_cnt[s] > 1) {  // the state is the one the packet before it left
                    const vpz_packet &pp = packets[L - 1];
                    const PacketInfo &ppi = D.packet_info[pp.flags & 7];
                    S.has_prev = true;
                    S.prev_long = pp.flags & VPZ_PKT_BLOCK_FLAG;
                    S.prev_end = ppi.right_start;
                    S.prev_stop = ppi.right_end;
                    S.prev_start = S.prev_end;
                }
            } else {
                int right_start = pi.right_start;
                if (pk.granule != -1 && eos) {  // :658-666
                    const int64_t actual_end = pos_base + out_off_of(L) + (has_prev ? prev_stop - prev_end : 0);
                    const int diff = (int)(actual_end - pk.granule);
                    if (diff > 0) right_start = std::max(right_start - diff, 0);
                }
                const int start = has_prev ? pi.left_start : right_start;  // :674 / :679
                const int d = right_start - start;
                const int cnt = std::max(0, d);
                if (right_start != pi.right_start) {  // trimmed: the last frame, the stream's total
                    out_count[s] += cnt - psamples[L];
                    psamples[L] = cnt;
                    D.trim_out_count[s] = cnt;
                    D.trim_left_start[s] = start;
                    if (!compact) {
                        frames[L].out_count = (uint16_t)cnt;
                        frames[L].left_start = (uint16_t)start;
                        frames[L].flags &= ~kFrameSteady;  // (an EOS trim changed the geometry)
                    }
                }
                if (pk.granule != -1 && !has_pos) {  // :459-463 at the last packet itself
                    has_pos = true;
                    pos_base = pk.granule - d - out_off_of(L);
                }
                S.has_prev = true;
                S.prev_long = pk.flags & VPZ_PKT_BLOCK_FLAG;
                S.prev_end = right_start;
                S.prev_stop = pi.right_end;
                S.prev_start = S.prev_end;
            }
            S.has_position = has_pos;
            S.current_position = pos_base + out_count[s];
        }
        for (int s = 0; s < D.n_streams; ++s)
            if (out_count[s] > capacity_of(s))
                return set_error(ctx, VPZ_E_CAPACITY, "vpz_decoder_synth: stream_out_capacity too small");
        n_frames = (size_t)n_packets;
        for (int s = 0; s < D.n_streams; ++s) samples_written[s] = out_count[s];
        if (any_floor0) need_coupling = true;  // type-0 floors are applied in place on the planar temp
        return 1;
    }

    // Pass 1: StreamDecoder.Read / ReadNextPacket per stream (StreamDecoder.cs:418-498, 640-694) -> one
    // FrameDesc per packet that produces or carries samples, written in place, stream-major.
    int run_state_machine(int64_t *samples_written)
    {
        st = D.states;
        std::vector<int64_t> &s_base = D.s_base, &s_cnt = D.s_cnt, &out_count = D.out_count;
        s_base.assign((size_t)D.n_streams + 1, 0);
        s_cnt.assign((size_t)D.n_streams, 0);
        out_count.assign((size_t)D.n_streams, 0);
        if (D.n_streams > 1) {
            for (int64_t p = 0; p < n_packets; ++p) {
                const int32_t s = packets[p].stream;
                if (s < 0 || s >= D.n_streams)
                    return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_synth: packet stream index out of range");
                ++s_base[(size_t)s + 1];
            }
            for (int s = 0; s < D.n_streams; ++s) s_base[(size_t)s + 1] += s_base[(size_t)s];
        }
        started_with_prev.resize(D.n_streams);
        started_prev_long.resize(D.n_streams);
        for (int s = 0; s < D.n_streams; ++s) {
            started_with_prev[s] = st[s].has_prev;
            started_prev_long[s] = st[s].prev_long;
        }
        frames = arena_alloc<FrameDesc>(*A, (size_t)n_packets);
        rec_floor = have_posts ? arena_alloc<uint8_t>(*A, (size_t)n_rec) : nullptr;
        if (rec_floor) memset(rec_floor, 0, (size_t)n_rec);
        need_coupling = D.generic;  // the generic path always works on its own planar copy

        for (int64_t p = 0; p < n_packets; ++p) {
            const vpz_packet &pk = packets[p];
            if (pk.stream < 0 || pk.stream >= D.n_streams)
                return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_synth: packet stream index out of range");
            StreamState &S = st[pk.stream];
            // Read(): once EOS was seen and the previous packet is drained nothing more is read (:441-447)
            if (S.eos_found && S.prev_start == S.prev_end) continue;
            // DecodeNextPacket :718-722, before the packet's first bit is looked at
            if (pk.flags & VPZ_PKT_RESYNC) S.has_position = false;
            const bool eos = pk.flags & VPZ_PKT_EOS;
            if (eos) S.eos_found = true;  // _eosFound |= isEndOfStream (:647), before the null check
            if (pk.flags & VPZ_PKT_NOT_DECODED) {
                if (eos && S.has_prev && S.prev_stop > S.prev_end) {  // :451-455 drain, un-windowed
                    FrameDesc fd{};
                    fd.flags = kFrameDrain;
                    fd.prev_end = (uint16_t)S.prev_end;
                    fd.out_count = (uint16_t)(S.prev_stop - S.prev_end);
                    fd.out_off = out_count[pk.stream];
                    out_count[pk.stream] += fd.out_count;
                    D.packet_samples[(size_t)p] = fd.out_count;
                    S.current_position += fd.out_count;
                    S.prev_end = S.prev_stop;
                    S.prev_start = S.prev_stop;
                    frames[s_base[pk.stream] + s_cnt[pk.stream]++] = fd;
                }
                continue;
            }
            const bool no_floor = pk.flags & VPZ_PKT_NO_FLOOR;
            if (!no_floor) {
                if (pk.mapping >= D.mappings.size())
                    return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_synth: packet mapping index out of range");
                if (!have_posts)
                    return set_error(ctx, VPZ_E_INVALID_ARG,
                                     "vpz_decoder_synth: posts and a floor table are required unless VPZ_PKT_NO_FLOOR");
            }
            if (pk.residue_offset < 0)
                return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_synth: negative residue offset");

            const bool bf = pk.flags & VPZ_PKT_BLOCK_FLAG;
            const PacketInfo &pi = D.packet_info[pk.flags & 7];  // Mode.GetPacketInfo, tabulated at create
            const int packet_len = S.prev_stop - S.prev_end;  // :654
            int right_start = pi.right_start;
            if (pk.granule != -1 && eos) {  // :658-666
                const int64_t actual_end = S.current_position + packet_len;
                const int diff = (int)(actual_end - pk.granule);
                if (diff > 0) right_start = std::max(right_start - diff, 0);
            }
            FrameDesc fd{};
            fd.rec = (int32_t)(p * C);
            fd.flags = frame_flags(pk, pi);
            if (S.has_prev) {  // :670-675
                const int slope_len = pi.left_use_size1 ? half1 : half0;
                if (packet_len > slope_len) {
                    // windowSlope.AsSpan(0, packetLen) would throw (:778): that Read fails, the packet is
                    // consumed and the decoder state stays as it was.  The rest of the batch is still
                    // synthesised; the call reports the condition at the end.
                    ++mismatches;
                    D.mismatch_packets.push_back(p);
                    continue;
                }
                fd.packet_len = (uint16_t)packet_len;
                fd.prev_end = (uint16_t)S.prev_end;
                S.prev_start = pi.left_start;
            } else {
                fd.packet_len = 0;
                S.prev_start = right_start;  // :679 first packet has no valid data before rightStart
            }
            fd.left_start = (uint16_t)S.prev_start;  // emission starts at the new _prevPacketStart
            S.prev_end = right_start;
            S.prev_stop = pi.right_end;
            S.has_prev = true;
            S.prev_long = bf;
            if (pk.granule != -1 && !S.has_position) {  // :459-463 (idx == 0 here)
                S.has_position = true;
                S.current_position = pk.granule - (S.prev_end - S.prev_start);
            }
            // a trim below LeftStart would make the reference spin (copyLen <= 0, :469-472): emit nothing
            fd.out_count = (uint16_t)std::max(0, S.prev_end - S.prev_start);
            fd.out_off = out_count[pk.stream];
            out_count[pk.stream] += fd.out_count;
            D.packet_samples[(size_t)p] = fd.out_count;
            S.current_position += fd.out_count;
            S.prev_start = S.prev_end;  // everything readable is handed out by this call
            fd.spec_off = pk.residue_offset;  // replaced by the temp offset when the coupling pass runs
            res_extent = std::max(res_extent, pk.residue_offset + (int64_t)C * (bf ? half1 : half0));
            if (!bf) any_short = true;
            if (!no_floor) {
                any_floor = true;
                const vpz_mapping_config &mc = D.mappings[pk.mapping];
                if (mc.coupling_steps > 0) need_coupling = true;
                const uint8_t long_bit = bf ? 0x80 : 0;
                for (int ch = 0; ch < C; ++ch) {
                    const uint8_t fl = mc.channel_floor[ch];
                    const bool f0 = D.floor_types[fl] == 0;
                    any_floor0 |= f0;
                    rec_floor[(size_t)(p * C + ch)] = (uint8_t)(fl | long_bit | (f0 ? 0x40 : 0));
                }
            }
            if (pk.flags & VPZ_PKT_INTERLEAVED) { need_coupling = true; ilv_seen = true; }
            else planar_seen = true;
            if (pk.residue_offset & 3) group_align_ok = false;  // group mode reads every packet in 16-byte pieces
            if (pk.residue_offset & 1) align2_ok = false;
            if (frame_is_steady(fd.flags, D.size1, fd.left_start, fd.packet_len, fd.prev_end, fd.out_count)) fd.flags |= kFrameSteady;
            frames[s_base[pk.stream] + s_cnt[pk.stream]++] = fd;
        }
        for (int s = 0; s < D.n_streams; ++s)
            if (out_count[s] > capacity_of(s))
                return set_error(ctx, VPZ_E_CAPACITY, "vpz_decoder_synth: stream_out_capacity too small");
        // close the gaps skipped packets left between the streams' frame ranges
        for (int s = 0; s < D.n_streams; ++s) {
            if (s_cnt[s] && (size_t)s_base[s] != n_frames)
                memmove(frames + n_frames, frames + s_base[s], sizeof(FrameDesc) * (size_t)s_cnt[s]);
            s_base[s] = (int64_t)n_frames;
            n_frames += (size_t)s_cnt[s];
        }
        for (int s = 0; s < D.n_streams; ++s) samples_written[s] = out_count[s];
        if (any_floor0) need_coupling = true;  // type-0 floors are applied in place on the planar temp
        return VPZ_OK;
    }

    // Pass 2: cut each stream's frames into runs.  A wavefront synthesises R consecutive blocks of one channel
    // (+1 recomputed block in front); R (<= 32) is the value for which the run count fills k whole rounds of
    // the resident waves with the least total work k * (R + 1); short batches fall back to R = 4.
    void cut_runs()
    {
        const int64_t total_frames = (int64_t)n_frames;
        const int r_max = use_dual ? kMaxRunLengthDual : D.big ? kMaxRunLengthBig : (synth_needs_general(D.size0, D.size1) ? kMaxRunLengthGeneral : kMaxRunLength);
        // Group mode synthesises up to eight consecutive SHORT blocks of a run in one pass (synth_kernel's run builder):
        // a block that rides along costs a fraction of a pass.  Runs are cut to equal COST, in eighths of a pass -- a
        // run rich in short blocks holds more frames --, so that every wavefront of the launch has the same amount to do.
        // (cutting by cost walks every packet a few times: with enough streams it is split over the host pool, streams
        // being independent; a small batch is walked on this thread; a large batch of few streams keeps runs of equal
        // length -- the kernel still batches what it finds in them -- rather than spend a millisecond of host time)
        HostPool *pool = static_cast<HostPool *>(ctx->host_pool);
        const bool wide = pool && D.n_streams >= 2 * pool->parties();
        // (... or, with short blocks to batch, is walked in pieces: 65 536 frames of ONE stream cut by cost are a millisecond on one
        // thread and 40 us on sixteen once the cut hint applies -- and worth 12 % of the kernel's time, configs[2])
        const bool split = pool && !wide && pool->parties() > 1 && total_frames > 4096;
        const bool batches = compact && !any_floor0 && (use_dual || (use_group && any_floor)) && any_short &&
                             !synth_needs_general(D.size0, D.size1) && D.size0 == 256 && D.size1 != 256 && !D.generic &&
                             (wide || total_frames <= 4096 || split) && !(D.ablate & 128);
        cut_by_cost = batches;
        const int parties = (batches && (wide || split)) ? pool->parties() : 1;
        // what a thread cuts is a SEGMENT: a stream, or -- few streams, many packets -- a piece of one (runs do not cross
        // segments; the first run of a piece inside a stream recomputes its predecessor like any run that is not a stream's first)
        std::vector<Decoder::CutSeg> &segs = D.cut_segs;
        segs.clear();
        for (int st_i = 0; st_i < D.n_streams; ++st_i) {
            const int64_t cnt = D.s_cnt[st_i];
            int pieces = 1;
            if (batches && split) pieces = (int)std::max<int64_t>(1, std::min<int64_t>(cnt / 1024, (cnt * 4 * parties + total_frames - 1) / total_frames));
            for (int k = 0; k < pieces; ++k) {
                const int64_t a = cnt * k / pieces, b = cnt * (k + 1) / pieces;
                segs.push_back(Decoder::CutSeg{st_i, (int32_t)a, (int32_t)(b - a)});
            }
        }
        const int n_segs = (int)segs.size();
        auto seg_range = [&](int c, int &lo, int &hi) {
            lo = (int)((int64_t)n_segs * c / parties);
            hi = (int)((int64_t)n_segs * (c + 1) / parties);
        };
        auto joins_batch = [&](int64_t p, bool prev_in_run_ok, bool &ok) -> bool {  // does packet p ride with its predecessor?
            ok = false;
            if (!batches || p == 0 || packets[p - 1].stream != packets[p].stream) return false;
            const vpz_packet &pk = packets[p], &pp = packets[p - 1];
            // (the stereo fast path batches planar and already-floored packets too; a batch holds one kind)
            ok = !(pk.flags & VPZ_PKT_BLOCK_FLAG) && !(pk.flags & VPZ_PKT_NOT_DECODED) && !(pp.flags & VPZ_PKT_NOT_DECODED) &&
                 (use_dual || ((pk.flags & VPZ_PKT_INTERLEAVED) && !(pk.flags & VPZ_PKT_NO_FLOOR)));
            // (after a long block it can head a batch; it rides with a short predecessor of its mapping)
            return ok && prev_in_run_ok && !(pp.flags & VPZ_PKT_BLOCK_FLAG) && pk.mapping == pp.mapping &&
                   ((pk.flags ^ pp.flags) & VPZ_PKT_NO_FLOOR) == 0;
        };
        // cost of a pass in eighths of a long block's (group mode, tools/kbench_short_long.py: a short block alone 0.74, eight
        // in one batch 3.2 together; the stereo fast path, fitted to the waves' durations on real streams -- tools/wave_times.sh,
        // wave_times_fit.py: a short block alone or at the head of a batch costs a whole pass, 8.6 / 7.7 eighths, every block
        // riding along 2.0)
        int w_short = use_dual ? 8 : 6, w_member = use_dual ? 2 : 3;
        if (const char *w = getenv("VPZ_CUT_WEIGHTS")) (void)sscanf(w, "%d,%d", &w_short, &w_member);  // tuning
        w_short = std::min(std::max(w_short, 1), 8);  // (a frame costs at most a whole pass: see the runs' capacity below)
        w_member = std::min(std::max(w_member, 1), 8);
        auto unit_cost = [&](int64_t p, bool ok, int pos) -> int {
            if (ok && (pos & 7) != 0) return w_member;
            return (batches && !(packets[p].flags & VPZ_PKT_BLOCK_FLAG)) ? w_short : 8;
        };
        int64_t total_units = 8 * total_frames;
        // what the later walks need to know about a packet, one byte each (bit 0: short block, bit 1: may ride in a batch,
        // bit 2: same mapping as its predecessor) -- they then touch no packet
        std::vector<uint8_t> &code = D.cut_code;
        // a decoder's next batch usually has the shape of its last one: R and the fitted cost target are taken over, the
        // packets are walked ONCE (codes and cut together, one fork-join), and only if the runs do not fit the rounds
        // after all is the whole procedure gone through
        const bool reuse = batches && parties > 1 && D.cut_hint_frames == total_frames && D.cut_hint_streams == D.n_streams &&
                           D.cut_hint_R > 0 && D.run_length_override <= 0;
        auto packet_code = [&](int64_t p, bool ok) -> uint8_t {
            return (uint8_t)(((packets[p].flags & VPZ_PKT_BLOCK_FLAG) ? 0 : 1) | (ok ? 2 : 0) |
                             ((ok && p > 0 && packets[p].mapping == packets[p - 1].mapping &&
                               !(packets[p - 1].flags & VPZ_PKT_BLOCK_FLAG)) ? 4 : 0));
        };
        if (batches && code.size() < (size_t)total_frames) code.resize((size_t)total_frames);
        if (batches) D.s_units.assign((size_t)n_segs, 0);
        if (batches && !reuse) {
            std::vector<int64_t> part(parties, 0);
            auto count = [&](int c) {
                int lo, hi;
                seg_range(c, lo, hi);
                int64_t units = 0;
                for (int s = lo; s < hi; ++s) {
                    int pos = -1;
                    bool prev_ok = false;
                    int64_t mine = 0;
                    for (int64_t p = D.s_base[segs[s].stream] + segs[s].off, e = p + segs[s].cnt; p < e; ++p) {
                        bool ok;
                        const bool link = joins_batch(p, prev_ok, ok);
                        code[(size_t)p] = packet_code(p, ok);
                        pos = ok ? (link ? pos + 1 : 0) : -1;
                        mine += unit_cost(p, ok, pos);
                        prev_ok = ok;
                    }
                    D.s_units[(size_t)s] = mine;
                    units += mine;
                }
                part[c] = units;
            };
            if (parties > 1) host_failed |= !pool->run(count); else count(0);
            total_units = 0;
            for (int64_t v : part) total_units += v;
        }
        int R = reuse ? D.cut_hint_R : std::min(D.run_length_override, r_max);
        int64_t run_slots = reuse ? D.cut_hint_slots : 0;  // runs that fit the rounds R was chosen for
        bool single_round = false;  // every run has a resident wave slot of its own from the start of the launch
        if (R <= 0) {
            const int64_t slots = std::max(1, use_dual ? synth_dual_resident_slots(any_floor, ctx->num_cu)
                                              : D.big ? synth_big_resident_waves(any_floor, ctx->num_cu, D.size0, D.size1)
                                                      : synth_resident_waves(any_floor, ctx->num_cu, C, use_group));
            const int64_t work = (total_units + 7) / 8 * C;
            R = 4;
            int64_t best = -1;
            // The stereo fast path chains the runs of a workgroup (chain_runs: three of four recompute nothing), and the memory
            // system delivers more the shorter the runs are -- the launch's waves then sweep the batch round by round instead of
            // streaming through all of it at once (tools/io_shapes.hip, profiles/r5_io_shapes.txt: 5.05 / 5.5 / 5.7 TB/s for runs of
            // 32 / 16 / 8 frames with the arithmetic removed): the rounds whose runs come closest to kDualChainRun frames.
            const int chain_run = D.dual_run;
            // (batches cut by COST -- streams with short blocks -- keep their long runs: measured, shorter ones lose there, configs[2]
            // 0.205 -> 0.226 ms and configs[4]'s share 0.258 -> 0.302 at 8 frames, profiles/r5_ab_chain_batches.txt; their runs are
            // chained all the same)
            const bool chained = use_dual && compact && D.size1 == 2048 && !D.no_chain && !batches;
            // (chained runs are cut by length: the grid is known exactly -- four runs to a workgroup, one workgroup per channel pair
            // of a chunk on the pair route -- and a run length whose grid is ONE workgroup over k rounds costs a round: 8 192 frames
            // of 10 channels in runs of 10 are 1 025 workgroups on 1 024 places)
            const int64_t resident_wgs = std::max<int64_t>(1, slots / (2 * synth_dual_waves()));
            auto grid_of = [&](int64_t r) -> int64_t {
                int64_t n = 0;
                for (int st_i = 0; st_i < D.n_streams; ++st_i) n += (D.s_cnt[st_i] + r - 1) / r;
                return (n + synth_dual_waves() - 1) / synth_dual_waves() * (D.pairs ? C / 2 : 1);
            };
            for (int k = 1; k <= 64; ++k) {
                int64_t r = (work + k * slots - 1) / (k * slots);
                if (chained)
                    while (r <= r_max && grid_of(r) > k * resident_wgs) ++r;
                if (r > r_max) continue;
                if (r < 4) break;
                // (chained: four runs share one recomputed block, and a run length below the preferred one only adds prologues)
                const int64_t cost = chained ? (r >= chain_run ? 4 * r + 1 : 1000 + (chain_run - r)) : (int64_t)k * (r + 1);
                if (best < 0 || cost < best) { best = cost; R = (int)r; run_slots = k * slots / C; single_round = k == 1; }
            }
        }
        // (a run cut by cost holds at least R - 1 frames unless its stream ends: every frame costs at most a whole pass)
        // (a retry -- the reused cut hint did not fit, see the end of this function -- takes the first attempt's place
        // in the arena instead of a second allocation)
        if (runs_arena_mark == (size_t)-1) runs_arena_mark = A->used;
        A->used = runs_arena_mark;
        // (... and the lighter half of a skewed cut holds that much less: see THE SKEW below)
        static const int skew_cap_permille = [] { const char *e = getenv("VPZ_CUT_SKEW"); return e ? std::max(0, atoi(e)) : 25; }();
        const int min_run_frames = std::max(1, R - 1 - ((skew_cap_permille > 0 && R >= 16) ? R * skew_cap_permille / 1000 + 2 : 0));
        // (every segment -- a stream, or a piece of a long one -- ends with a partial run)
        runs_cap = (size_t)(total_frames / min_run_frames) + (size_t)std::max(n_segs, D.n_streams) + 1;
        runs = arena_alloc<RunDesc>(*A, runs_cap);  // (throws ArenaOverflow -> VPZ_E_NOMEM: open_arena's budget is R >= 4 runs)
        if (D.generic) return;
        int64_t target_units = 8 * (int64_t)R;
        // one run of a frame: as many frames from f0 on as the cost target (and the descriptor area) allow
        auto run_length = [&](int64_t base, int f0, int cnt, int64_t target, int64_t &units) -> int {
            int len = 0;
            units = 0;
            int pos = -1;
            bool prev_ok = false;
            const uint8_t *cd = code.data() + base + f0;
            while (f0 + len < cnt && len < r_max) {
                const uint8_t c8 = cd[len];
                const bool ok = c8 & 2;
                const bool link = ok && prev_ok && (c8 & 4);
                pos = ok ? (link ? pos + 1 : 0) : -1;
                const int u = (ok && (pos & 7) != 0) ? w_member : ((c8 & 1) ? w_short : 8);
                if (len > 0 && units + u > target) break;
                units += u;
                prev_ok = ok;
                ++len;
            }
            return len;
        };
        if (reuse) target_units = D.cut_hint_target;
        // THE SKEW.  With one round of runs, the first half of the grid's workgroups are the first to arrive on their CUs and
        // the second half join them as each CU's second workgroup -- and the waves of the second arrivals run slower for the
        // whole launch (measured per wave, tools/wave_times.sh: identical runs take 600 k cycles in wave slot 0 of their SIMD,
        // 647 k in slot 1; priorities set with s_setprio do not change it), so with equal work the early half idles at the end
        // while the late half finishes at half occupancy.  Runs that start in the first half of the batch's WORK -- they are
        // the first half of the grid -- are therefore cut 2 % heavier, the others as much lighter (tools/try_cut_skew.sh:
        // configs[4] 0.283 -> 0.278 ms at 20 per mille, worse again from 40 on).
        static const int skew_permille = [] { const char *e = getenv("VPZ_CUT_SKEW"); return e ? std::max(0, atoi(e)) : 25; }();
        int64_t heavy_work = reuse ? D.cut_hint_heavy : -1;
        if (batches && !reuse && single_round && use_dual && skew_permille > 0 && R >= 16) {  // (short runs: nothing to skew by)
            heavy_work = total_units * (1000 + skew_permille) / 2000;
            D.cut_prefix.resize((size_t)n_segs);
            int64_t acc = 0;
            for (int st_i = 0; st_i < n_segs; ++st_i) {
                D.cut_prefix[(size_t)st_i] = acc;
                acc += D.s_units[(size_t)st_i];
            }
        }
        if (heavy_work >= 0 && D.cut_prefix.size() != (size_t)n_segs) heavy_work = -1;
        // (all-long batches only: with short blocks in runs of equal length a frame more is not 3 % more -- configs[2] lost 2 %)
        // (group mode -- 6 channels, two workgroups of 8 waves per CU -- does not respond to it: configs[3] 0.323 either way)
        const bool skew_frames = !batches && !any_short && single_round && use_dual && skew_permille > 0 && R >= 24 &&
                                 R + 1 <= r_max && D.run_length_override <= 0;
        const int64_t heavy_frames = total_frames * (R + 1) / (2 * (int64_t)R);  // the first half of the frames' work at R + 1 each
        // the target of the run of stream `st_i` that starts `before` cost units into its stream
        auto target_at = [&](int st_i, int64_t before, int64_t target) -> int64_t {
            if (heavy_work < 0) return target;
            const int64_t sk = target * skew_permille / 1000;
            return D.cut_prefix[(size_t)st_i] + before < heavy_work ? target + sk : target - sk;
        };
        if (batches && run_slots > 0 && !reuse) {
            // runs of equal cost do not pack as evenly as runs of equal length (and every stream ends with a partial
            // one): a few more runs than the rounds hold would put a nearly empty round behind them -- count, and give
            // every run a little more until they fit
            auto runs_with = [&](int64_t target) -> int64_t {
                std::vector<int64_t> part(parties, 0);
                auto count = [&](int c) {
                    int lo, hi;
                    seg_range(c, lo, hi);
                    int64_t n = 0;
                    for (int s = lo; s < hi; ++s) {
                        int64_t before = 0, u = 0;
                        const int64_t seg_base = D.s_base[segs[s].stream] + segs[s].off;
                        for (int f0 = 0, cnt = segs[s].cnt; f0 < cnt; ++n, before += u)
                            f0 += run_length(seg_base, f0, cnt, target_at(s, before, target), u);
                    }
                    part[c] = n;
                };
                if (parties > 1) host_failed |= !pool->run(count); else count(0);
                int64_t n_total = 0;
                for (int64_t v : part) n_total += v;
                if (getenv("VPZ_HOST_PROFILE"))
                    fprintf(stderr, "[vpz host] run cutting: R %d, target %lld eighths, %lld runs for %lld slots\n", R,
                            (long long)target, (long long)n_total, (long long)run_slots);
                return n_total;
            };
            // in steps of half a pass until the runs fit, then back in eighths: the lightest runs that still fit (a launch
            // takes as long as its heaviest run; half a pass is 2 % of one)
            bool fits = false;
            for (int tries = 0; tries < 6 && target_units / 8 < r_max; ++tries) {
                if ((fits = runs_with(target_units) <= run_slots)) break;
                target_units += 4;
            }
            if (fits && target_units > 8 * (int64_t)R) {
                for (int64_t t = target_units - 3; t < target_units; ++t)
                    if (runs_with(t) <= run_slots) { target_units = t; break; }
            }
            D.cut_hint_R = R;
            D.cut_hint_slots = run_slots;
            D.cut_hint_target = target_units;
            D.cut_hint_heavy = heavy_work;
            D.cut_hint_frames = total_frames;
            D.cut_hint_streams = D.n_streams;
        }
        // the run of segment g (stream s, its packets from `base` on) that starts f0 frames into the segment and holds len frames
        auto make_run = [&](int g, int f0, int len) -> RunDesc {
            const int s = segs[g].stream, seg_off = segs[g].off;
            const int base = (int)D.s_base[s] + seg_off;
            const int stream_cnt = (int)D.s_cnt[s];
            RunDesc r{};
            r.first = base + f0;
            r.count = len;
            r.stream = s;
            if (seg_off + f0 == 0) {
                r.pre_kind = started_with_prev[s] ? kPreState : kPreNone;
                r.prev_long = started_prev_long[s];
            } else {
                r.pre_kind = kPreRecompute;
            }
            const bool last = seg_off + f0 + len >= stream_cnt;
            if (last) r.flags |= kRunSaveState;
            r.clip_epoch = D.states[s].clip_epoch;
            r.state_slot = D.states[s].state_slot;
            if (compact) {
                r.flags |= kRunCompact;
                const int64_t q = (int64_t)r.first + (r.pre_kind == kPreRecompute ? -1 : 0);  // first staged frame
                r.rec_base = (int32_t)(q * C);
                r.spec_base = packets[q].residue_offset;
                r.out_base = D.out_off_scratch[(size_t)r.first];
                if (q > 0 && packets[q - 1].stream == s) {
                    const PacketInfo &ppi = D.packet_info[packets[q - 1].flags & 7];
                    r.has_prev0 = 1;
                    r.prev_end0 = (uint16_t)ppi.right_start;
                    r.prev_stop0 = (uint16_t)ppi.right_end;
                } else {
                    const StreamState &S0 = D.states[s];
                    r.has_prev0 = S0.has_prev ? 1 : 0;
                    r.prev_end0 = (uint16_t)S0.prev_end;
                    r.prev_stop0 = (uint16_t)S0.prev_stop;
                }
                if (last && D.trim_out_count[s] >= 0) {  // the stream's last frame was cut by the EOS trim
                    r.flags |= kRunLastTrimmed;
                    r.last_out_count = (uint16_t)D.trim_out_count[s];
                    r.last_left_start = (uint16_t)D.trim_left_start[s];
                }
            }
            return r;
        };
        // Runs of equal LENGTH: the k-th run of a segment is its frames [k R, (k + 1) R) -- every run's place is known in advance,
        // so a large batch is filled in by the host pool, every thread its share of the run indices (thousands of runs of 8 frames
        // were 85 us on one thread: as long as the kernel takes for a third of them)
        const bool by_length_wide = !batches && !skew_frames && pool && pool->parties() > 1 && total_frames / std::max(1, R) >= 1024;
        if (by_length_wide) {
            std::vector<int64_t> &seg_first = D.cut_prefix;
            seg_first.assign((size_t)n_segs + 1, 0);
            for (int g = 0; g < n_segs; ++g) seg_first[(size_t)g + 1] = seg_first[(size_t)g] + (segs[g].cnt + R - 1) / R;
            const int64_t total_runs = seg_first[(size_t)n_segs];
            if ((size_t)total_runs > runs_cap) { host_failed = true; n_runs = 0; return; }
            const int P = pool->parties();
            host_failed |= !pool->run([&](int c) {
                const int64_t lo = total_runs * c / P, hi = total_runs * (c + 1) / P;
                int g = (int)(std::upper_bound(seg_first.begin(), seg_first.end(), lo) - seg_first.begin()) - 1;
                for (int64_t i = lo; i < hi; ++i) {
                    while (i >= seg_first[(size_t)g + 1]) ++g;
                    const int f0 = (int)(i - seg_first[(size_t)g]) * R;
                    runs[i] = make_run(g, f0, std::min(R, segs[g].cnt - f0));
                }
            });
            if (host_failed) { n_runs = 0; return; }
            n_runs = (size_t)total_runs;
        }
        std::vector<std::vector<RunDesc>> cut(by_length_wide ? 0 : parties);
        auto cut_streams = [&](int c) {
          int s_lo, s_hi;
          seg_range(c, s_lo, s_hi);
          std::vector<RunDesc> &mine = cut[c];
          if (parties > 1) {
              // (a stream without packets in this call keeps s_base = s_cnt = 0: count the range's packets, never
              // subtract bases)
              int64_t pk_in_range = 0;
              for (int g = s_lo; g < s_hi; ++g) pk_in_range += segs[g].cnt;
              mine.reserve((size_t)(pk_in_range / min_run_frames) + (size_t)(s_hi - s_lo) + 1);
          }
          for (int g = s_lo; g < s_hi; ++g) {
            const int s = segs[g].stream, seg_off = segs[g].off;
            const int cnt = segs[g].cnt, base = (int)D.s_base[s] + seg_off;  // (f0 below counts from the segment's start)
            if (reuse) {  // (the codes of this segment's packets, skipped with the counting pass)
                bool prev_ok = false;
                for (int64_t p = base, e = (int64_t)base + cnt; p < e; ++p) {
                    bool ok;
                    (void)joins_batch(p, prev_ok, ok);
                    code[(size_t)p] = packet_code(p, ok);
                    prev_ok = ok;
                }
            }
            int64_t before = 0, run_units = 0;
            for (int f0 = 0; f0 < cnt; before += run_units) {
                // (runs of equal LENGTH: the skew is a frame more in the first half of the frames, a frame less in the second)
                const int len = batches ? run_length(base, f0, cnt, target_at(g, before, target_units), run_units)
                                        : std::min(R + (skew_frames ? ((int64_t)base + f0 < heavy_frames ? 1 : -1) : 0), cnt - f0);
                const RunDesc r = make_run(g, f0, len);
                if (parties > 1) mine.push_back(r);
                else if (n_runs < runs_cap) runs[n_runs++] = r;
                else { host_failed = true; return; }
                f0 += len;
            }
          }
        };
        if (by_length_wide) {
            // (filled in above)
        } else if (parties > 1) {
            host_failed |= !pool->run(cut_streams);
            if (host_failed) { n_runs = 0; return; }
            for (const std::vector<RunDesc> &v : cut) {
                if (n_runs + v.size() > runs_cap) { host_failed = true; n_runs = 0; return; }  // (never silently into what follows)
                memcpy(runs + n_runs, v.data(), v.size() * sizeof(RunDesc));
                n_runs += v.size();
            }
        } else {
            cut_streams(0);
        }
        // this batch is not like the last one after all: more runs than the rounds hold, or -- a lighter mix of blocks, so
        // fewer and longer runs -- so few that part of the resident waves would idle through the launch
        if (reuse && ((int64_t)n_runs > run_slots || (int64_t)n_runs * 10 < D.cut_hint_runs * 9)) {
            D.cut_hint_frames = -1;
            n_runs = 0;
            cut_runs();
            return;
        }
        if (batches && !reuse) D.cut_hint_runs = (int64_t)n_runs;
        cut_R = R;
        chain_runs();
        if (getenv("VPZ_HOST_PROFILE"))
            fprintf(stderr, "[vpz host] cut: %s, by %s, R %d, target %lld eighths, %zu runs for %lld slots, %d segments on %d threads, heavy below %lld, "
                            "hint (frames %lld, runs %lld), chained %lld\n", reuse ? "hint reused" : "fitted", batches ? "cost" : "length", R,
                    (long long)target_units, n_runs, (long long)run_slots, n_segs, parties, (long long)heavy_work,
                    (long long)D.cut_hint_frames, (long long)D.cut_hint_runs, (long long)n_chained);
    }

    // The stereo fast path: runs r - 1 and r of one stream that land in ONE workgroup (the kernel takes run i in wave i mod
    // kDualWaves of workgroup i / kDualWaves) and meet in the steady state -- a 2048 block after a 2048 block, long windows on
    // both sides -- are CHAINED: the later one recomputes nothing, it overlaps its first frame with the tail the earlier one
    // leaves in LDS (kPreNeighbour, synth_desc.hpp).  With three runs of four chained, short runs cost a quarter of what their
    // recomputed blocks did, and short runs are what the memory system likes (tools/io_shapes.hip: the waves of a launch then
    // sweep a quarter or an eighth of the batch at a time instead of all of it).
    int64_t n_chained = 0;
    int cut_R = 0;  // the run length (or cost target, in passes) cut_runs settled on
    void chain_runs()
    {
        n_chained = 0;
        if (!use_dual || !compact || D.size1 != 2048 || D.no_chain) return;
        const size_t waves = (size_t)synth_dual_waves();
        // (a run looks at its predecessor's place and length only -- what the chaining never changes: any split of the runs works)
        auto chain_range = [&](size_t lo, size_t hi) -> int64_t {
            int64_t n = 0;
            for (size_t i = std::max<size_t>(lo, 1); i < hi; ++i) {
                RunDesc &r = runs[i];
                const RunDesc &pr = runs[i - 1];
                if (i % waves == 0 || r.pre_kind != kPreRecompute || r.stream != pr.stream || pr.count <= 0 ||
                    r.first != pr.first + pr.count || !(r.flags & kRunCompact) || r.count <= 0)
                    continue;
                const int64_t q = r.first;
                if (q <= 0 || packets[q - 1].stream != r.stream) continue;
                const uint8_t cf = cflags[q], pcf = cflags[q - 1];
                // frame q: long, long windows on both sides, taken; frame q - 1: long with a long window towards q, taken
                if ((cf & (7u | kCfSkip)) != 7u || (pcf & (1u | 4u | kCfSkip)) != 5u) continue;
                if (r.count == 1 && (r.flags & kRunLastTrimmed)) continue;  // (its only frame is the stream's EOS-trimmed last one)
                r.pre_kind = kPreNeighbour;
                r.rec_base = (int32_t)(q * C);
                r.spec_base = packets[q].residue_offset;
                const PacketInfo &ppi = D.packet_info[packets[q - 1].flags & 7];
                r.has_prev0 = 1;
                r.prev_end0 = (uint16_t)ppi.right_start;
                r.prev_stop0 = (uint16_t)ppi.right_end;
                ++n;
            }
            return n;
        };
        // the runs' flag bytes inline (SynthArgs.run_inline): one trip to the pinned arena per run instead of two
        // (only where runs are short enough to use them: the kernel takes the bytes of a run of up to 16 staged frames inline)
        const bool want_inline = cut_R > 0 && cut_R + 1 <= 16;
        run_inline = want_inline ? arena_alloc<uint8_t>(*A, 32 * n_runs + 32) : nullptr;
        auto inline_range = [&](size_t lo, size_t hi) {
            for (size_t i = lo; want_inline && i < hi; ++i) {
                const RunDesc &r = runs[i];
                uint8_t *dst = run_inline + 32 * i;
                memset(dst, 0, 32);
                if (!(r.flags & kRunCompact)) continue;
                const int64_t q = (int64_t)r.first + (r.pre_kind == kPreRecompute ? -1 : 0);
                const int n = std::min(16, r.count + (r.pre_kind == kPreRecompute ? 1 : 0));
                for (int j = 0; j < n; ++j) {
                    dst[j] = cflags[q + j];
                    dst[16 + j] = cmap[q + j];
                }
            }
        };
        // (one sweep does both: a run's bytes depend on its own record only, its chaining on its predecessor's place and length)
        HostPool *pool = static_cast<HostPool *>(ctx->host_pool);
        if (pool && pool->parties() > 1 && n_runs >= 1024) {  // (the pool's workers are still spinning from the cut's fork-join)
            const int P = pool->parties();
            std::vector<int64_t> part((size_t)P, 0);
            host_failed |= !pool->run([&](int c) {
                const size_t lo = n_runs * (size_t)c / P, hi = n_runs * (size_t)(c + 1) / P;
                part[(size_t)c] = chain_range(lo, hi);
                inline_range(lo, hi);
            });
            for (int64_t v : part) n_chained += v;
        } else {
            n_chained = chain_range(0, n_runs);
            inline_range(0, n_runs);
        }
    }

    // coupling packets: de-interleave + inverse coupling into a planar temp laid out in frame order
    void build_coupling_packets()
    {
        if (use_dual && D.pairs && !compact) {
            // explicit descriptors for the pair route: which steps a frame has depends on the PAIR that looks at it -- the frame
            // names its mapping (bits 16..23), and the kernel puts the pair's count and offset in (SynthArgs.map_bits)
            for (size_t fi = 0; fi < n_frames; ++fi) {
                FrameDesc &fd = frames[fi];
                fd.flags &= ~0x00FFFF00u;
                if (!(fd.flags & (kFrameDrain | kFrameNoFloor))) fd.flags |= (uint32_t)packets[fd.rec / C].mapping << kFrameStepsOffShift;
            }
        }
        if (!need_coupling || use_group || use_dual) return;
        // the separate pass hands planar, de-coupled spectra over: the frames lose their group-mode bits
        for (size_t fi = 0; fi < n_frames; ++fi) frames[fi].flags &= 0xFu | (kFrameSkipMask << kFrameSkipShift) | kFrameSteady;
        const size_t cps = coupling_packet_size();
        cpk = arena_alloc<uint8_t>(*A, cps * n_frames);
        for (size_t fi = 0; fi < n_frames; ++fi) {
            FrameDesc &fd = frames[fi];
            if (fd.flags & kFrameDrain) continue;
            const vpz_packet &pk = packets[fd.rec / C];
            const int half = (fd.flags & kFrameLong) ? half1 : half0;
            const bool couple = !(fd.flags & kFrameNoFloor) && D.mappings[pk.mapping].coupling_steps > 0;
            fill_coupling_packet(cpk + (size_t)n_cpk * cps, pk.residue_offset, temp_floats, half,
                                 couple ? D.mapping_steps_off[pk.mapping] : -1,
                                 couple ? D.mappings[pk.mapping].coupling_steps : 0,
                                 (pk.flags & VPZ_PKT_INTERLEAVED) ? 1 : 0);
            fd.spec_off = temp_floats;
            temp_floats += (int64_t)C * half;
            ++n_cpk;
        }
    }

    int build_floor0_records()
    {
        if (!any_floor0) return VPZ_OK;
        if (use_dual) {  // (the stereo fast path applies type-0 floors itself: floor0_curve_kernel works from the per-record info)
            if (!D.f0_amp || !D.f0_coeff || D.f0_stride < 1)
                return set_error(ctx, VPZ_E_INVALID_ARG,
                                 "vpz_decoder_synth: type-0 floors need vpz_decoder_set_floor0_data before the call");
            return VPZ_OK;
        }
        if (!D.f0_amp || !D.f0_coeff || D.f0_stride < 1)
            return set_error(ctx, VPZ_E_INVALID_ARG,
                             "vpz_decoder_synth: type-0 floors need vpz_decoder_set_floor0_data before the call");
        const size_t rs = floor0_rec_size();
        f0recs = arena_alloc<uint8_t>(*A, rs * n_frames * (size_t)C);
        for (size_t fi = 0; fi < n_frames; ++fi) {
            const FrameDesc &fd = frames[fi];
            if (fd.flags & (kFrameDrain | kFrameNoFloor)) continue;
            const vpz_mapping_config &mc = D.mappings[packets[fd.rec / C].mapping];
            const int half = (fd.flags & kFrameLong) ? half1 : half0;
            for (int ch = 0; ch < C; ++ch) {
                const int fl = mc.channel_floor[ch];
                if (D.floor_types[fl] != 0) continue;
                fill_floor0_rec(f0recs + (size_t)n_f0 * rs, fd.spec_off + (int64_t)ch * half, fd.rec + ch, fl, half,
                                (fd.flags & kFrameLong) ? 1 : 0);
                ++n_f0;
            }
        }
        return VPZ_OK;
    }

    int build_output_offsets()
    {
        if (!out_interleaved && channel_stride < stream_out_capacity && C > 1)
            return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_synth: channel_stride smaller than stream_out_capacity");
        offs = arena_alloc<int64_t>(*A, (size_t)D.n_streams);
        dev_channel_stride = channel_stride;
        if (mem_space == VPZ_MEM_HOST) {
            // A host-memory call mirrors its PCM on the device: the streams' areas BACK TO BACK there, whatever lies between them in
            // the caller's array (two short songs at either end of a library's PCM array are two short areas, not the span between
            // them), each on a 256-byte boundary so that every store of the kernels takes its widest form; copy_back knows both places
            const int64_t align = 64;
            auto up = [&](int64_t v) { return (v + align - 1) / align * align; };
            int64_t at = 0;
            if (out_interleaved) {
                for (int s = 0; s < D.n_streams; ++s) {
                    offs[s] = at;
                    at += up(D.out_count[s] * C);
                }
            } else {
                int64_t most = 0;
                for (int s = 0; s < D.n_streams; ++s) most = std::max(most, D.out_count[s]);
                dev_channel_stride = up(most);
                for (int s = 0; s < D.n_streams; ++s) {
                    offs[s] = at;
                    if (D.out_count[s] > 0) at += dev_channel_stride * C;
                }
            }
            mirror_elems = at;
        } else {
            for (int s = 0; s < D.n_streams; ++s) offs[s] = stream_out_offset ? stream_out_offset[s] : 0;
        }
        return VPZ_OK;
    }
    int64_t dev_channel_stride = 0;  // the channel stride the kernels use: the caller's, or the device mirror's (host-memory calls)
    int64_t mirror_elems = 0;        // PCM elements of a host-memory call's device mirror

    // any-block-size path: per-frame records + gather lists of the two exact-IMDCT launches
    void build_generic_lists()
    {
        if (!D.generic) return;
        gf = arena_alloc<GenericFrame>(*A, n_frames);
        src0 = arena_alloc<int64_t>(*A, n_frames * (size_t)C);
        dst0 = arena_alloc<int64_t>(*A, n_frames * (size_t)C);
        src1 = arena_alloc<int64_t>(*A, n_frames * (size_t)C);
        dst1 = arena_alloc<int64_t>(*A, n_frames * (size_t)C);
        save_list = arena_alloc<int32_t>(*A, (size_t)D.n_streams + 1);
        size_t fi = 0;
        for (int s = 0; s < D.n_streams; ++s) {
            const size_t cnt = (size_t)D.s_cnt[s];
            int64_t prev_y = started_with_prev[s] ? -1 : -2;
            int prev_n = started_prev_long[s] ? D.size1 : D.size0;
            long last_block = -1;
            for (size_t k = 0; k < cnt; ++k, ++fi) {
                const FrameDesc &fd = frames[fi];
                GenericFrame g{};
                g.spec_off = fd.spec_off;
                g.out_off = fd.out_off;
                g.rec = fd.rec;
                g.stream = s;
                g.left_start = fd.left_start;
                g.packet_len = fd.packet_len;
                g.prev_end = fd.prev_end;
                g.out_count = fd.out_count;
                g.flags = fd.flags;
                g.prev_y_off = prev_y;
                g.prev_n = prev_n;
                if (!(fd.flags & kFrameDrain)) {
                    g.n = (fd.flags & kFrameLong) ? D.size1 : D.size0;
                    g.y_off = y_floats;
                    for (int ch = 0; ch < C; ++ch) {
                        const int64_t so = fd.spec_off + (int64_t)ch * (g.n / 2), dofs = y_floats + (int64_t)ch * g.n;
                        if (fd.flags & kFrameLong) { src1[n1] = so; dst1[n1++] = dofs; }
                        else { src0[n0] = so; dst0[n0++] = dofs; }
                    }
                    y_floats += (int64_t)C * g.n;
                    prev_y = g.y_off;
                    prev_n = g.n;
                    last_block = (long)fi;
                }
                gf[fi] = g;
            }
            if (last_block >= 0) {
                gf[last_block].flags |= kFrameSaveState;
                save_list[n_save++] = (int32_t)last_block;
            }
        }
    }

    // Device side, part 1: VPZ_MEM_HOST inputs are staged, work buffers grown, the arena uploaded.
    // VPZ_MEM_HOST: the caller's residue (and floor records) start their way to the device BEFORE the host state machine runs
    // -- neither depends on it, and the link is what a host-memory call waits for: the pass over the packets (50 us for a
    // sub-batch of 16 real streams on an idle core, several times that while the cores decode) passes under the copy.  What
    // is copied: up to the highest residue a decodable packet names (>= what the state machine will use; within the extent
    // the caller stated, else nothing is started here and the call fails where it always did).  The caller's buffers are
    // read asynchronously from here on: every way out of the call synchronises (see EarlyUploadGuard).
    // the caller's host residue into D.b_in_res: as it is, or -- VPZ_RESIDUE_I16 -- at 2 bytes a value over the link and widened there
    int upload_residue(int64_t ext)
    {
        int rc;
        if (D.residue_format == VPZ_RESIDUE_I16) {
            // (the 16-bit values go over as they are; whether a kernel reads them in place or they are widened first is settled in
            // stage_inputs, once the state machine has said which kernels run)
            if ((rc = grow(ctx, D.b_in_res16, sizeof(int16_t) * (size_t)ext)) != VPZ_OK) return rc;
            VPZ_HIP_TRY(ctx, hipMemcpyAsync(D.b_in_res16.p, residue, sizeof(int16_t) * (size_t)ext, hipMemcpyHostToDevice, ctx->stream));
        } else {
            if ((rc = grow(ctx, D.b_in_res, sizeof(float) * (size_t)ext)) != VPZ_OK) return rc;
            VPZ_HIP_TRY(ctx, hipMemcpyAsync(D.b_in_res.p, residue, sizeof(float) * (size_t)ext, hipMemcpyHostToDevice, ctx->stream));
        }
        return VPZ_OK;
    }

    int stage_inputs_early(int64_t residue_floats, int64_t n_records)
    {
        if (mem_space != VPZ_MEM_HOST || !residue) return VPZ_OK;
        int64_t ext = 0;
        for (int64_t p = 0; p < n_packets; ++p) {
            const vpz_packet &pk = packets[p];
            if (pk.flags & VPZ_PKT_NOT_DECODED) continue;
            if (pk.residue_offset < 0) return VPZ_OK;
            ext = std::max(ext, pk.residue_offset + (int64_t)C * ((pk.flags & VPZ_PKT_BLOCK_FLAG) ? half1 : half0));
        }
        if (ext <= 0 || ext > residue_floats) return VPZ_OK;
        int rc;
        if ((rc = upload_residue(ext)) != VPZ_OK) return rc;
        early_residue = true;
        early_res_extent = ext;
        if (have_posts && n_records >= n_rec && n_rec > 0) {
            if ((rc = grow(ctx, D.b_in_posts, sizeof(int16_t) * 64 * (size_t)n_rec)) != VPZ_OK) return rc;
            if ((rc = grow(ctx, D.b_in_counts, (size_t)n_rec)) != VPZ_OK) return rc;
            VPZ_HIP_TRY(ctx, hipMemcpyAsync(D.b_in_posts.p, posts, sizeof(int16_t) * 64 * (size_t)n_rec, hipMemcpyHostToDevice,
                                            ctx->stream));
            VPZ_HIP_TRY(ctx, hipMemcpyAsync(D.b_in_counts.p, post_counts, (size_t)n_rec, hipMemcpyHostToDevice, ctx->stream));
            early_posts = true;
        }
        return VPZ_OK;
    }

    int stage_inputs()
    {
        int rc;
        d_res = residue;
        d_posts = posts;
        d_counts = post_counts;
        d_amp = D.f0_amp;
        d_coeff = D.f0_coeff;
        d_out = pcm_out;
        // VPZ_RESIDUE_I16 (ABI v5): the floored stereo fast path reads the 16-bit values in place and widens them in registers; every
        // other kernel reads float32 -- the values are widened into the decoder's staging buffer first (exact either way)
        const bool i16 = D.residue_format == VPZ_RESIDUE_I16;
        spec_i16 = i16 && use_dual && !D.pairs && any_floor && !D.no_direct_i16 &&
                   (mem_space == VPZ_MEM_HOST || (reinterpret_cast<uintptr_t>(residue) & 7) == 0);
        if (mem_space != VPZ_MEM_HOST && i16 && !spec_i16) {  // device-resident int16 values: widened into the staging buffer
            if ((rc = grow(ctx, D.b_in_res, sizeof(float) * (size_t)res_extent)) != VPZ_OK) return rc;
            VPZ_HIP_TRY(ctx, launch_widen_i16(residue, static_cast<float *>(D.b_in_res.p), res_extent, ctx->num_cu, ctx->stream));
            d_res = static_cast<const float *>(D.b_in_res.p);
        }
        if (mem_space == VPZ_MEM_HOST) {
            if (!(early_residue && early_res_extent >= res_extent) && (rc = upload_residue(res_extent)) != VPZ_OK) return rc;
            if (i16 && !spec_i16) {
                if ((rc = grow(ctx, D.b_in_res, sizeof(float) * (size_t)res_extent)) != VPZ_OK) return rc;
                VPZ_HIP_TRY(ctx, launch_widen_i16(D.b_in_res16.p, static_cast<float *>(D.b_in_res.p), res_extent, ctx->num_cu, ctx->stream));
            }
            d_res = spec_i16 ? static_cast<const float *>(D.b_in_res16.p) : static_cast<const float *>(D.b_in_res.p);
            if (any_floor) {
                if (!early_posts) {
                    if ((rc = grow(ctx, D.b_in_posts, sizeof(int16_t) * 64 * (size_t)n_rec)) != VPZ_OK) return rc;
                    if ((rc = grow(ctx, D.b_in_counts, (size_t)n_rec)) != VPZ_OK) return rc;
                    VPZ_HIP_TRY(ctx, hipMemcpyAsync(D.b_in_posts.p, posts, sizeof(int16_t) * 64 * (size_t)n_rec,
                                                    hipMemcpyHostToDevice, ctx->stream));
                    VPZ_HIP_TRY(ctx, hipMemcpyAsync(D.b_in_counts.p, post_counts, (size_t)n_rec, hipMemcpyHostToDevice,
                                                    ctx->stream));
                }
                d_posts = static_cast<const int16_t *>(D.b_in_posts.p);
                d_counts = static_cast<const uint8_t *>(D.b_in_counts.p);
            }
            if (any_floor0) {
                if ((rc = grow(ctx, D.b_in_amp, sizeof(float) * (size_t)n_rec)) != VPZ_OK) return rc;
                if ((rc = grow(ctx, D.b_in_coeff, sizeof(float) * (size_t)n_rec * D.f0_stride)) != VPZ_OK) return rc;
                VPZ_HIP_TRY(ctx, hipMemcpyAsync(D.b_in_amp.p, D.f0_amp, sizeof(float) * (size_t)n_rec,
                                                hipMemcpyHostToDevice, ctx->stream));
                VPZ_HIP_TRY(ctx, hipMemcpyAsync(D.b_in_coeff.p, D.f0_coeff, sizeof(float) * (size_t)n_rec * D.f0_stride,
                                                hipMemcpyHostToDevice, ctx->stream));
                d_amp = static_cast<const float *>(D.b_in_amp.p);
                d_coeff = static_cast<const float *>(D.b_in_coeff.p);
            }
            if ((rc = grow(ctx, D.b_out, out_elem * (size_t)mirror_elems + 16)) != VPZ_OK) return rc;
            d_out = D.b_out.p;
        }
        if (need_coupling && !use_group && !use_dual && (rc = grow(ctx, D.b_temp, sizeof(float) * (size_t)temp_floats)) != VPZ_OK)
            return rc;
        if (any_floor0 && use_dual && (rc = grow(ctx, D.b_f0curve, sizeof(float) * (size_t)n_rec * (size_t)D.f0_k)) != VPZ_OK) return rc;
        if (any_floor) {
            if ((rc = grow(ctx, D.b_cposts, sizeof(int32_t) * 64 * (size_t)n_rec)) != VPZ_OK) return rc;
            if ((rc = grow(ctx, D.b_ccount, (size_t)n_rec)) != VPZ_OK) return rc;
            if (D.generic && (rc = grow(ctx, D.b_curve, (size_t)n_rec * (size_t)half1)) != VPZ_OK) return rc;
        }
        if (D.generic && (rc = grow(ctx, D.b_ybuf, sizeof(float) * (size_t)std::max<int64_t>(y_floats, 1))) != VPZ_OK)
            return rc;
        // (A second stream for this upload, ordered with events so that it overlaps the previous call's kernels, was
        // measured 2-3x SLOWER per call on MI355X / ROCm 7.2: cross-stream event waits cost more than the copy.)
        // A small arena is not copied at all: the kernels read the few hundred KiB of descriptors straight from the
        // pinned host memory (each wave fetches its 64-byte run record and two bytes per frame once, at its start).
        // That takes a DMA command and its two command-processor gaps (~25 us) off every call; the arena stays
        // untouched until the kernels are done (`uploaded` is recorded after the launches in that case).
        zero_copy = A->mapped != nullptr && A->used <= D.zero_copy_max && !D.generic;
        if (!zero_copy) {
            VPZ_HIP_TRY(ctx, hipMemcpyAsync(A->dev.p, A->base, A->used, hipMemcpyHostToDevice, ctx->stream));
            VPZ_HIP_TRY(ctx, hipEventRecord(A->uploaded, ctx->stream));
            A->pending = true;
        }
        return VPZ_OK;
    }

    // Device side, part 2: the kernels, all asynchronous on the context's stream.
    int launch()
    {
        const float *d_spec = d_res;
        const int64_t *d_outoff = (stream_out_offset || mem_space == VPZ_MEM_HOST) ? static_cast<const int64_t *>(dev(offs)) : nullptr;
        if (need_coupling && !use_group && !use_dual) {
            hipError_t e = launch_coupling(dev(cpk), n_cpk, D.d_steps, C, d_res, static_cast<float *>(D.b_temp.p), half1,
                                           ctx->stream);
            if (e != hipSuccess) return set_error(ctx, VPZ_E_HIP, "coupling kernel launch", e);
            d_spec = static_cast<const float *>(D.b_temp.p);
        }
        const int32_t *d_cposts = static_cast<const int32_t *>(D.b_cposts.p);
        const uint8_t *d_ccount = static_cast<const uint8_t *>(D.b_ccount.p);
        if (any_floor) {  // Floor1.UnwrapPosts and the choice of the posts a line is drawn to, per channel record
            const bool f0_fused = any_floor0 && use_dual;
            // (a decoder whose floors are all of type 0 has nothing to unwrap: the curve kernel leaves the records' markers itself)
            hipError_t e = hipSuccess;
            if (!(f0_fused && !D.has_floor1))
                e = launch_floor1_unwrap((int)n_rec, d_posts, d_counts, static_cast<uint8_t *>(dev(rec_floor)), D.d_floors,
                                         (int)D.floors.size(), static_cast<int32_t *>(D.b_cposts.p),
                                         static_cast<uint8_t *>(D.b_ccount.p), nullptr, nullptr, ctx->stream, f0_fused ? 1 : 0);
            if (e == hipSuccess && f0_fused) {  // the records' Floor0 curves over their bark indices (Floor0.cs:188-219)
                e = launch_floor0_curves((int)n_rec, static_cast<uint8_t *>(dev(rec_floor)), D.d_floors0, d_amp, d_coeff, D.f0_stride,
                                         D.f0_k, D.d_f0_w, static_cast<float *>(D.b_f0curve.p), d_counts,
                                         static_cast<uint8_t *>(D.b_ccount.p), static_cast<int32_t *>(D.b_cposts.p), ctx->stream);
                D.f0_amp = D.f0_coeff = nullptr;  // consumed
            }
            if (e == hipSuccess && D.generic)  // the three-pass path reads the curve from memory
                e = launch_floor1_render((int)n_rec, d_cposts, d_ccount, static_cast<uint8_t *>(dev(rec_floor)), half0,
                                         half1, static_cast<uint8_t *>(D.b_curve.p), ctx->stream);
            if (e != hipSuccess) return set_error(ctx, VPZ_E_HIP, "floor1 unwrap kernel launch", e);
        }
        if (any_floor0 && !use_dual) {  // Floor0.Apply in place on the temp (rare; Floor0.cs:164-225)
            hipError_t e = launch_floor0_apply(dev(f0recs), n_f0, D.d_floors0, D.d_bark_maps, d_amp, d_coeff, D.f0_stride,
                                               static_cast<float *>(D.b_temp.p), ctx->stream);
            if (e != hipSuccess) return set_error(ctx, VPZ_E_HIP, "floor0 kernel launch", e);
            D.f0_amp = D.f0_coeff = nullptr;  // consumed
        }
        if (D.generic) {
            // any-block-size path: floor pass, exact IMDCT per size, OLA pass, state save
            const GenericFrame *d_gf = static_cast<const GenericFrame *>(dev(gf));
            float *d_temp = static_cast<float *>(D.b_temp.p);
            float *d_y = static_cast<float *>(D.b_ybuf.p);
            hipError_t e = hipSuccess;
            if (any_floor)
                e = launch_generic_floor(d_gf, (int)n_frames, C, half1, d_temp, d_ccount,
                                         static_cast<const uint8_t *>(D.b_curve.p), ctx->d_inv_db, ctx->stream);
            // per block size: the gathered FAST transform (every size from 256 up), else the reference's own
            // schedule (64 and 128 must take it: quirk q1)
            auto imdct_gathered = [&](int n, BlockTables *t, int64_t cnt, const int64_t *so, const int64_t *dof) {
                if (n == 4096 && t->d_fast)
                    return launch_imdct_fast_4096(d_temp, d_y, cnt, t->d_fast, ctx, ctx->stream, so, dof);
                if (n == 8192 && t->d_fast)
                    return launch_imdct_fast_8192(d_temp, d_y, cnt, t->d_fast, ctx, ctx->stream, so, dof);
                if (n == 2048 && t->d_fast)
                    return launch_imdct_fast_2048(d_temp, d_y, cnt, t->d_fast, ctx, ctx->stream, so, dof);
                if (n == 256 && t->d_fast)
                    return launch_imdct_fast_256(d_temp, d_y, cnt, t->d_fast, ctx, ctx->stream, so, dof);
                if ((n == 512 || n == 1024) && t->d_fast)
                    return launch_imdct_fast_mid(n, d_temp, d_y, cnt, t->d_fast, ctx, ctx->stream, so, dof);
                return launch_imdct_exact(n, t->ld, d_temp, d_y, cnt, t->d_A, t->d_B, t->d_C, t->d_bitrev, ctx->num_cu,
                                          ctx->stream, so, dof);
            };
            if (e == hipSuccess && n0)
                e = imdct_gathered(D.size0, D.t0, (int64_t)n0, static_cast<const int64_t *>(dev(src0)),
                                   static_cast<const int64_t *>(dev(dst0)));
            if (e == hipSuccess && n1)
                e = imdct_gathered(D.size1, D.t1, (int64_t)n1, static_cast<const int64_t *>(dev(src1)),
                                   static_cast<const int64_t *>(dev(dst1)));
            if (e == hipSuccess)
                e = launch_generic_ola(d_gf, (int)n_frames, C, D.size0, D.size1, d_y, D.d_state_h, D.t0->d_slope,
                                       D.t1->d_slope, static_cast<float *>(d_out), d_outoff, dev_channel_stride, out_interleaved,
                                       D.clip, D.d_clipped, out_s16 ? 1 : 0, ctx->stream);
            if (e == hipSuccess)
                e = launch_generic_save_state(d_gf, static_cast<const int32_t *>(dev(save_list)), (int)n_save, C, D.size1,
                                              d_y, D.d_state_h, ctx->stream);
            if (e != hipSuccess) return set_error(ctx, VPZ_E_HIP, "generic synthesis kernel launch", e);
            return VPZ_OK;
        }
        SynthArgs a{};
        a.frames = static_cast<const FrameDesc *>(dev(frames));
        a.cflags = static_cast<const uint8_t *>(dev(cflags));
        a.cmap = static_cast<const uint8_t *>(dev(cmap));
        a.run_inline = (use_dual && !D.no_run_inline) ? static_cast<const uint8_t *>(dev(run_inline)) : nullptr;
        a.map_bits = D.d_map_bits;
        a.pair_ch = D.d_pair_ch;
        a.n_pairs = C / 2;
        a.n_mappings = (int32_t)std::max<size_t>(1, D.mappings.size());
        for (int f = 0; f < 8; ++f) {
            const PacketInfo &pi = D.packet_info[f];
            a.geom[f] = PacketGeom{(uint16_t)pi.left_start, (uint16_t)pi.right_start, (uint16_t)pi.right_end,
                                   (uint16_t)pi.left_use_size1};
        }
        a.runs = static_cast<const RunDesc *>(dev(runs));
        a.n_runs = (int32_t)n_runs;
        a.channels = C;
        a.size0 = D.size0;
        a.size1 = D.size1;
        a.spec = d_spec;
        a.ccount = any_floor ? d_ccount : nullptr;
        a.cposts = any_floor ? d_cposts : nullptr;
        a.steps = D.d_steps_lvl;
        a.n_step_pairs = D.n_step_pairs;
        a.max_steps = D.max_steps;
        if (use_dual && D.pairs) {  // the pairs' own step lists and per-(pair, mapping) words
            a.map_bits = D.d_pair_map_bits;
            a.steps = D.d_pair_steps;
            a.n_step_pairs = D.n_pair_step_pairs;
        }
        a.group = use_group ? 1 : 0;
        a.group_dma = (use_group && D.group_dma) ? 1 : 0;
        a.inv_db = ctx->d_inv_db;
        a.f0_curve = static_cast<const float *>(D.b_f0curve.p);
        a.f0_bark = D.d_f0_bark;
        a.f0_stride = D.f0_k;
        a.spec_i16 = spec_i16 ? 1 : 0;
        a.state_h = D.d_state_h;
        if (D.big && !use_dual) {
            const int64_t tf = synth_big_tail_floats(D.size1, (int64_t)n_runs * C);
            if (tf > 0) {
                const int grc = grow(ctx, D.b_bigtail, sizeof(float) * (size_t)tf);
                if (grc != VPZ_OK) return grc;
            }
            a.big_tail = static_cast<float *>(D.b_bigtail.p);
        }
        a.state_slot_floats = (int64_t)D.n_streams * C * half1;
        a.tw_long = D.t1->d_fast;
        a.tw_short = D.t0->d_fast;
        a.slope0 = D.t0->d_slope;
        a.slope1 = D.t1->d_slope;
        a.out = static_cast<float *>(d_out);
        a.stream_out_off = d_outoff;
        a.channel_stride = dev_channel_stride;
        a.interleaved = out_interleaved;
        a.s16 = out_s16 ? 1 : 0;
        a.clip = D.clip;
        a.clipped = D.d_clipped;
        a.no_batch = ((D.ablate & 128) || !cut_by_cost || any_floor0) ? 1 : 0;
        a.ablate = D.ablate;
        a.stamps = nullptr;
#if defined(VPZ_STAMPS) || defined(VPZ_WAVE_TIMES)
        static unsigned long long *d_stamps = nullptr;
        constexpr size_t kStampWaves = 1 << 16;  // per-wave records behind the 16 sums: [16 + 16 * run]
        if (!d_stamps) (void)hipMalloc(&d_stamps, (16 + 16 * kStampWaves) * sizeof(unsigned long long));
        (void)hipMemsetAsync(d_stamps, 0, (16 + 16 * std::min<size_t>(kStampWaves, n_runs)) * sizeof(unsigned long long), ctx->stream);
        a.stamps = d_stamps;
#endif
        hipError_t e = use_dual ? (D.pairs ? launch_synth_pairs(a, any_floor, ilv_seen, ctx->stream) : launch_synth_dual(a, any_floor, ilv_seen, ctx->stream))
                                : D.big ? launch_synth_big(a, any_floor, ctx->stream) : launch_synth(a, any_floor, ctx->stream);
        if (e != hipSuccess) return set_error(ctx, VPZ_E_HIP, "synth kernel launch", e);
#if defined(VPZ_STAMPS) || defined(VPZ_WAVE_TIMES)
        {
            unsigned long long h[16];
            (void)hipMemcpyAsync(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
            static const char *names_one[9] = {"desc+prefetch", "barrier0", "stage+barrier", "coupling", "pickup+curve",
                                               "floor*+imdct", "wait next", "ola+stores", "tail"};
            static const char *names_dual[9] = {"desc+prefetch", "unpack+coupling", "curves", "floor*+imdct x2", "wait next",
                                                "ola+stores", "tail", "-", "-"};
            const char **names = use_dual ? names_dual : names_one;
            unsigned long long tot = 0;
            for (int k = 0; k < 9; ++k) tot += h[k];
            fprintf(stderr, "[stamps] %llu waves, %.0f cycles per wave:", h[15], h[15] ? (double)tot / h[15] : 0.0);
            for (int k = 0; k < 9; ++k) fprintf(stderr, " %s %.1f%%", names[k], tot ? 100.0 * h[k] / tot : 0.0);
            if (h[14] && h[15]) {
                const double mean = (double)tot / h[15], var = (double)h[13] * 1e6 / h[15] - mean * mean;
                fprintf(stderr, " | slowest wave %.0f cycles = %.2f x mean, sigma %.2f x mean, %.1f passes per wave, %d runs", (double)h[14],
                        (double)h[14] / mean, var > 0 ? sqrt(var) / mean : 0.0, (double)h[12] / h[15], (int)n_runs);
            }
            fprintf(stderr, "\n");
            if (const char *dump = getenv("VPZ_STAMPS_DUMP")) {  // per-wave records: run, frames, passes, cycles per phase
                const size_t nw = std::min<size_t>(kStampWaves, n_runs);
                std::vector<unsigned long long> w(16 * nw);
                (void)hipMemcpy(w.data(), d_stamps + 16, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                if (FILE *f = fopen(dump, "w")) {
                    fprintf(f, "run,stream,frames,passes,long_frames,c0,c1,c2,c3,c4,c5,c6\n");
                    for (size_t r = 0; r < nw; ++r) {
                        fprintf(f, "%zu,%d,%d,%llu,%llu", r, runs[r].stream, runs[r].count, w[16 * r + 9], w[16 * r + 10]);
                        for (int k = 0; k < 7; ++k) fprintf(f, ",%llu", w[16 * r + k]);
                        fprintf(f, "\n");
                    }
                    fclose(f);
                }
            }
        }
#endif
        return VPZ_OK;
    }

    // VPZ_MEM_HOST: PCM back to the caller's buffer, synchronously
    int copy_back()
    {
        if (mem_space != VPZ_MEM_HOST) return VPZ_OK;
        for (int s = 0; s < D.n_streams; ++s) {
            if (D.out_count[s] <= 0) continue;
            char *h = static_cast<char *>(pcm_out);
            const char *dv = static_cast<const char *>(d_out);
            const int64_t host_off = stream_out_offset ? stream_out_offset[s] : 0;  // (the device mirror packs the areas: offs[s])
            if (out_interleaved) {
                VPZ_HIP_TRY(ctx, hipMemcpyAsync(h + out_elem * (size_t)host_off, dv + out_elem * (size_t)offs[s],
                                                out_elem * (size_t)(D.out_count[s] * C), hipMemcpyDeviceToHost,
                                                ctx->stream));
            } else {
                for (int ch = 0; ch < C; ++ch) {
                    const size_t at_h = out_elem * (size_t)(host_off + (int64_t)ch * channel_stride);
                    const size_t at_d = out_elem * (size_t)(offs[s] + (int64_t)ch * dev_channel_stride);
                    VPZ_HIP_TRY(ctx, hipMemcpyAsync(h + at_h, dv + at_d, out_elem * (size_t)D.out_count[s],
                                                    hipMemcpyDeviceToHost, ctx->stream));
                }
            }
        }
        VPZ_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return VPZ_OK;
    }
};

}  // namespace

static int synth_impl(vpz_decoder *d, int64_t n_packets, const vpz_packet *packets, const float *residue,
                      int64_t residue_floats, const int16_t *posts, const uint8_t *post_counts, int64_t n_records,
                      int mem_space, void *pcm_out, const int64_t *stream_out_offset, int64_t stream_out_capacity,
                      int out_layout, int64_t channel_stride, int64_t *samples_written)
{
    Decoder &D = d->impl;
    Context *ctx = D.ctx;
    D.mismatch_packets.clear();
    if (n_packets < 0 || (n_packets > 0 && (!packets || !residue || !pcm_out)) || !samples_written)
        return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_synth: null argument");
    // the extents of the caller's buffers are part of the call (the reference's arguments are Span<T>, bounds-checked:
    // Mapping.cs:98): a packet that addresses residue beyond them, or a batch with fewer post records than
    // packets * channels, is refused here -- it would be an out-of-bounds read on the device otherwise
    if (residue_floats < 0 || n_records < 0)
        return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_synth: negative buffer extent");
    if (posts && post_counts && n_records < n_packets * (int64_t)D.channels)
        return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_synth: fewer post records than packets * channels");
    if (mem_space != VPZ_MEM_HOST && mem_space != VPZ_MEM_DEVICE)
        return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_synth: bad mem_space");
    if (out_layout < VPZ_OUT_INTERLEAVED || out_layout > VPZ_OUT_PLANAR_S16)
        return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_synth: bad out_layout");
    if (n_packets > (int64_t)0x7fffffff / std::max(1, D.channels))
        return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_synth: batch too large");
    VPZ_HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int s = 0; s < D.n_streams; ++s) samples_written[s] = 0;
    D.packet_samples.assign((size_t)n_packets, 0);
    if (n_packets == 0) return VPZ_OK;

    const bool host_profile = getenv("VPZ_HOST_PROFILE") != nullptr;  // (per call: the tests switch it on for single calls)
    auto tick = [] { return std::chrono::steady_clock::now(); };
    const auto t_begin = tick();
    SynthCall call(D, n_packets, packets, residue, posts, post_counts, mem_space, pcm_out, stream_out_offset,
                   stream_out_capacity, out_layout, channel_stride);
    int rc;
    // (whatever the way out: no copy may still be reading the caller's buffers when the call returns)
    struct EarlyUploadGuard {
        Context *ctx;
        SynthCall &call;
        bool completed = false;
        ~EarlyUploadGuard()
        {
            if ((call.early_residue || call.early_posts) && !completed) (void)hipStreamSynchronize(ctx->stream);
        }
    } early_guard{ctx, call};
    if (!D.no_early_upload && (rc = call.stage_inputs_early(residue_floats, n_records)) != VPZ_OK) return rc;
    if ((rc = call.open_arena()) != VPZ_OK) return rc;
    const auto t_arena = tick();
    rc = call.run_state_machine_parallel(samples_written);
    if (rc < 0) return rc;
    const bool was_parallel = rc == 1;
    if (rc == 0 && (rc = call.run_state_machine(samples_written)) != VPZ_OK) return rc;
    // group mode of the fused kernel (de-interleave and inverse coupling in LDS) when the batch needs either and
    // its packets can be read in 16-byte pieces; otherwise the separate pass through a planar temp
    // ... and for interleaved output of more than two channels, which only a packet's waves together can write densely
    const bool wants_group = call.need_coupling || (call.out_interleaved && D.channels > 2);
    call.use_dual = call.dual_usable();
    call.use_group = !call.use_dual && D.group_ok && wants_group && !call.any_floor0 && call.group_align_ok &&
                     (mem_space == VPZ_MEM_HOST || (reinterpret_cast<uintptr_t>(residue) & 15) == 0);
    if (call.res_extent > residue_floats)
        return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_decoder_synth: a packet's residue lies beyond residue_floats");
    const auto t_pass1 = tick();
    if (call.n_frames == 0) {
        D.states = call.st;
        return VPZ_OK;  // (window mismatches are per-packet conditions: vpz_decoder_last_packet_status)
    }
    call.cut_runs();
    if (call.host_failed) return set_error(ctx, VPZ_E_NOMEM, "vpz_decoder_synth: run cutting failed (allocation)");
    call.build_coupling_packets();
    if ((rc = call.build_floor0_records()) != VPZ_OK) return rc;
    if ((rc = call.build_output_offsets()) != VPZ_OK) return rc;
    call.build_generic_lists();
    const auto t_pass2 = tick();
    if ((rc = call.stage_inputs()) != VPZ_OK) return rc;
    if ((rc = call.launch()) != VPZ_OK) return rc;
    if (call.zero_copy) {  // the kernels read the arena itself: it is free again when they are done
#ifdef VPZ_TUNING  // (timing experiments only, WRONG in general: no event behind the call -- what does the queue's barrier packet cost?)
        static const bool no_event = getenv("VPZ_UNSAFE_NO_ARENA_EVENT") != nullptr;
        if (!no_event)
#endif
        {
            VPZ_HIP_TRY(ctx, hipEventRecord(call.A->uploaded, ctx->stream));
            call.A->pending = true;
        }
    }
    if ((rc = call.copy_back()) != VPZ_OK) return rc;
    early_guard.completed = mem_space == VPZ_MEM_HOST;  // (copy_back has waited for the stream)
    if (!D.generic)  // every stream with frames in this batch has had its state written to the other copy
        for (int s = 0; s < D.n_streams; ++s)
            if (D.s_cnt[s] > 0) call.st[s].state_slot ^= 1;
    D.states = call.st;
    if (host_profile) {
        const auto t_end = tick();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        // (route: which kernel synthesises -- the tests read it to know that the route they ask for is the one that ran)
        const char *route = D.generic ? "generic" : call.use_dual ? (D.pairs ? "pairs" : "stereo") : D.big ? "big" : call.use_group ? "group" : "separate";
        fprintf(stderr, "[vpz host] packets %lld: route %s, arena wait %.1f us, pass1 %.1f us (%s), runs %.1f us, uploads+launch %.1f us\n",
                (long long)n_packets, route, us(t_begin, t_arena), us(t_arena, t_pass1), was_parallel ? "parallel" : "serial",
                us(t_pass1, t_pass2), us(t_pass2, t_end));
    }
    return VPZ_OK;  // (window mismatches are per-packet conditions: vpz_decoder_last_packet_status)
}

int vpz_decoder_synth(vpz_decoder *d, int64_t n_packets, const vpz_packet *packets, const float *residue,
                      int64_t residue_floats, const int16_t *posts, const uint8_t *post_counts, int64_t n_records,
                      int mem_space, void *pcm_out, const int64_t *stream_out_offset, int64_t stream_out_capacity,
                      int out_layout, int64_t channel_stride, int64_t *samples_written)
{
    if (!d) return VPZ_E_INVALID_ARG;
    // "never throws": the host half allocates (vectors, the descriptor arena); an allocation failure becomes a status
    try {
        return synth_impl(d, n_packets, packets, residue, residue_floats, posts, post_counts, n_records, mem_space, pcm_out,
                          stream_out_offset, stream_out_capacity, out_layout, channel_stride, samples_written);
    } catch (const ArenaOverflow &) {
        return set_error(d->impl.ctx, VPZ_E_NOMEM, "vpz_decoder_synth: the descriptor arena is too small for this batch");
    } catch (const std::bad_alloc &) {
        return set_error(d->impl.ctx, VPZ_E_NOMEM, "vpz_decoder_synth: host allocation failed");
    } catch (...) {
        return set_error(d->impl.ctx, VPZ_E_NOMEM, "vpz_decoder_synth: host pass failed");
    }
}

int vpz_decoder_last_packet_status(vpz_decoder *d, int32_t *out, int64_t capacity, int64_t *n_not_ok)
{
    if (!d || capacity < 0 || (capacity > 0 && !out)) return VPZ_E_INVALID_ARG;
    Decoder &D = d->impl;
    const int64_t n = std::min<int64_t>(capacity, (int64_t)D.packet_samples.size());
    for (int64_t i = 0; i < n; ++i) out[i] = VPZ_OK;
    for (int64_t p : D.mismatch_packets)
        if (p < n) out[p] = VPZ_E_WINDOW_MISMATCH;
    if (n_not_ok) *n_not_ok = (int64_t)D.mismatch_packets.size();
    return VPZ_OK;
}

// test-only: the integers of the Floor1 device path (include/vorbispizza_synth_debug.h)
int vpz_debug_floor1_indices(vpz_decoder *d, int64_t n_records, const int16_t *posts, const uint8_t *post_counts,
                             const uint8_t *record_floor, const uint8_t *record_long, uint8_t *curve_out,
                             int16_t *final_y_out, uint8_t *step_flags_out, uint8_t *active_count_out)
{
    if (!d) return VPZ_E_INVALID_ARG;
    Decoder &D = d->impl;
    Context *ctx = D.ctx;
    if (n_records < 0 || n_records > 0x7fffffff / 64 || (n_records > 0 && (!posts || !post_counts || !record_floor || !record_long)))
        return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_debug_floor1_indices: bad arguments");
    if (n_records == 0) return VPZ_OK;
    const size_t n = (size_t)n_records;
    const int half0 = D.size0 / 2, half1 = D.size1 / 2;
    std::vector<uint8_t> info(n);
    for (size_t r = 0; r < n; ++r) {
        if (record_floor[r] >= D.floors.size() || D.floor_types[record_floor[r]] != 1)
            return set_error(ctx, VPZ_E_INVALID_ARG, "vpz_debug_floor1_indices: record_floor is not a type-1 floor");
        info[r] = (uint8_t)(record_floor[r] | (record_long[r] ? 0x80 : 0));
    }
    VPZ_HIP_TRY(ctx, hipSetDevice(ctx->device));
    struct Bufs {
        void *p[7] = {};
        ~Bufs() { for (void *q : p) if (q) (void)hipFree(q); }
    } B;
    const size_t bytes[7] = {n * 128, n, n, n * 256, n, n * (size_t)half1, n * 64 * 3};
    for (int i = 0; i < 7; ++i) VPZ_HIP_TRY(ctx, hipMalloc(&B.p[i], bytes[i]));
    int16_t *d_posts = static_cast<int16_t *>(B.p[0]);
    uint8_t *d_counts = static_cast<uint8_t *>(B.p[1]), *d_info = static_cast<uint8_t *>(B.p[2]);
    int32_t *d_cposts = static_cast<int32_t *>(B.p[3]);
    uint8_t *d_ccount = static_cast<uint8_t *>(B.p[4]), *d_curve = static_cast<uint8_t *>(B.p[5]);
    int16_t *d_y = static_cast<int16_t *>(B.p[6]);
    uint8_t *d_f = static_cast<uint8_t *>(B.p[6]) + n * 128;
    VPZ_HIP_TRY(ctx, hipMemcpyAsync(d_posts, posts, bytes[0], hipMemcpyHostToDevice, ctx->stream));
    VPZ_HIP_TRY(ctx, hipMemcpyAsync(d_counts, post_counts, n, hipMemcpyHostToDevice, ctx->stream));
    VPZ_HIP_TRY(ctx, hipMemcpyAsync(d_info, info.data(), n, hipMemcpyHostToDevice, ctx->stream));
    if (curve_out) VPZ_HIP_TRY(ctx, hipMemcpyAsync(d_curve, curve_out, bytes[5], hipMemcpyHostToDevice, ctx->stream));
    VPZ_HIP_TRY(ctx, hipMemsetAsync(B.p[6], 0, bytes[6], ctx->stream));
    hipError_t e = launch_floor1_unwrap((int)n, d_posts, d_counts, d_info, D.d_floors, (int)D.floors.size(), d_cposts,
                                        d_ccount, d_y, d_f, ctx->stream);
    if (e == hipSuccess) e = launch_floor1_render((int)n, d_cposts, d_ccount, d_info, half0, half1, d_curve, ctx->stream);
    if (e != hipSuccess) return set_error(ctx, VPZ_E_HIP, "floor1 debug kernels", e);
    if (curve_out) VPZ_HIP_TRY(ctx, hipMemcpyAsync(curve_out, d_curve, bytes[5], hipMemcpyDeviceToHost, ctx->stream));
    if (final_y_out) VPZ_HIP_TRY(ctx, hipMemcpyAsync(final_y_out, d_y, n * 128, hipMemcpyDeviceToHost, ctx->stream));
    if (step_flags_out) VPZ_HIP_TRY(ctx, hipMemcpyAsync(step_flags_out, d_f, n * 64, hipMemcpyDeviceToHost, ctx->stream));
    if (active_count_out) VPZ_HIP_TRY(ctx, hipMemcpyAsync(active_count_out, d_ccount, n, hipMemcpyDeviceToHost, ctx->stream));
    VPZ_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VPZ_OK;
}

int vpz_decoder_set_floor0_data(vpz_decoder *d, const float *amp, const float *coeff, int32_t coeff_stride)
{
    if (!d) return VPZ_E_INVALID_ARG;
    Decoder &D = d->impl;
    if ((amp == nullptr) != (coeff == nullptr) || (amp && coeff_stride < 1))
        return set_error(D.ctx, VPZ_E_INVALID_ARG, "vpz_decoder_set_floor0_data: bad arguments");
    D.f0_amp = amp;
    D.f0_coeff = coeff;
    D.f0_stride = coeff_stride;
    return VPZ_OK;
}

int vpz_decoder_last_packet_samples(vpz_decoder *d, int32_t *out, int64_t capacity)
{
    if (!d || (!out && capacity > 0) || capacity < 0) return VPZ_E_INVALID_ARG;
    const Decoder &D = d->impl;
    const int64_t n = std::min<int64_t>(capacity, (int64_t)D.packet_samples.size());
    if (n > 0) memcpy(out, D.packet_samples.data(), sizeof(int32_t) * (size_t)n);
    return VPZ_OK;
}

int vpz_decoder_has_clipped(vpz_decoder *d, int32_t stream, int32_t *has_clipped)
{
    if (!d || !has_clipped) return VPZ_E_INVALID_ARG;
    Decoder &D = d->impl;
    if (stream < 0 || stream >= D.n_streams) return set_error(D.ctx, VPZ_E_INVALID_ARG, "bad stream index");
    int32_t v = 0;
    VPZ_HIP_TRY(D.ctx, hipMemcpyAsync(&v, D.d_clipped + stream, sizeof v, hipMemcpyDeviceToHost, D.ctx->stream));
    VPZ_HIP_TRY(D.ctx, hipStreamSynchronize(D.ctx->stream));
    *has_clipped = D.generic ? v != 0 : v == D.states[stream].clip_epoch;
    return VPZ_OK;
}

int vpz_decoder_set_position(vpz_decoder *d, int32_t stream, int64_t sample_position)
{
    if (!d) return VPZ_E_INVALID_ARG;
    Decoder &D = d->impl;
    if (stream < 0 || stream >= D.n_streams) return set_error(D.ctx, VPZ_E_INVALID_ARG, "vpz_decoder_set_position: bad stream");
    D.states[stream].current_position = sample_position;
    D.states[stream].has_position = true;
    return VPZ_OK;
}

int vpz_decoder_set_residue_format(vpz_decoder *d, int32_t format)
{
    if (!d) return VPZ_E_INVALID_ARG;
    Decoder &D = d->impl;
    if (format != VPZ_RESIDUE_F32 && format != VPZ_RESIDUE_I16)
        return set_error(D.ctx, VPZ_E_INVALID_ARG, "vpz_decoder_set_residue_format: neither VPZ_RESIDUE_F32 nor VPZ_RESIDUE_I16");
    D.residue_format = format;
    return VPZ_OK;
}

int vpz_decoder_set_host_threads(vpz_decoder *d, int32_t n)
{
    if (!d) return VPZ_E_INVALID_ARG;
    Decoder &D = d->impl;
    if (n < 0) return set_error(D.ctx, VPZ_E_INVALID_ARG, "vpz_decoder_set_host_threads: negative thread count");
    D.host_threads = std::min<int32_t>(n, 16);
    return VPZ_OK;
}

int vpz_decoder_set_stream_capacities(vpz_decoder *d, const int64_t *capacity, int32_t n)
{
    if (!d) return VPZ_E_INVALID_ARG;
    Decoder &D = d->impl;
    if (!capacity && n == 0) {
        D.stream_caps.clear();
        return VPZ_OK;
    }
    if (!capacity || n != D.n_streams)
        return set_error(D.ctx, VPZ_E_INVALID_ARG, "vpz_decoder_set_stream_capacities: one capacity per stream of the decoder, or (NULL, 0)");
    for (int32_t s = 0; s < n; ++s)
        if (capacity[s] < 0) return set_error(D.ctx, VPZ_E_INVALID_ARG, "vpz_decoder_set_stream_capacities: negative capacity");
    try {
        D.stream_caps.assign(capacity, capacity + n);
    } catch (const std::bad_alloc &) {
        return set_error(D.ctx, VPZ_E_NOMEM, "vpz_decoder_set_stream_capacities: out of host memory");
    }
    return VPZ_OK;
}

int vpz_decoder_position(vpz_decoder *d, int32_t stream, int64_t *sample_position)
{
    if (!d || !sample_position) return VPZ_E_INVALID_ARG;
    Decoder &D = d->impl;
    if (stream < 0 || stream >= D.n_streams) return set_error(D.ctx, VPZ_E_INVALID_ARG, "bad stream index");
    *sample_position = D.states[stream].current_position;
    return VPZ_OK;
}

}  // extern "C"
