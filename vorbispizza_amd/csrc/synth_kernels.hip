// Fused PCM-synthesis kernels behind vpz_decoder_synth:
//
//   floor1_prepare_kernel  Floor1.UnwrapPosts (Floor1.cs:270-353, one LANE per channel-record) and the
//                          curve render (Floor1.cs:236-262, 372-397) as one table index per bin;
//                          integer only, bit-exact.
//   coupling_kernel        Residue2 de-interleave (Residue2.cs:42-51) + inverse square-polar
//                          coupling (Mapping.cs:166-172, 198-269), element-wise, bit-exact.
//   synth_kernel           per channel-block: inverse dB table lookup x residue (Floor1.cs:383,395)
//                          -> inverse MDCT (Mdct.cs) -> window + overlap-add with the
//                          previous block (StreamDecoder.cs:764-791) -> clip (Utils.cs:44-58) ->
//                          interleaved / planar store (StreamDecoder.cs:515-638).
//
// synth_kernel gives one wavefront a RUN of consecutive blocks of one channel of one stream, so
// the previous block's output stays in LDS and the only HBM traffic is the spectrum in and the
// PCM out (8 B per sample for long blocks).  The block before a run's first is recomputed (one
// redundant IMDCT per run) or taken from the decoder's saved state.
#include <cstdlib>
#include <type_traits>

#include "imdct_core.hpp"
#include "synth_common.hpp"
#include "synth_desc.hpp"
#include "vpz_internal.hpp"

namespace vpz {

constexpr int kSynthWaves = 8;
constexpr int kSynthThreads = 64 * kSynthWaves;

// ---------------------------------------------------------------------------------------------
// De-interleave + inverse coupling into a planar temp.  One thread per (packet, bin).
// ---------------------------------------------------------------------------------------------
struct CouplingPacket {
    int64_t src_off;     // float offset in the caller's residue buffer
    int64_t dst_off;     // float offset in the planar temp (== FrameDesc.spec_off)
    int32_t half;        // blocksize / 2
    int32_t steps_off;   // offset into the steps array (pairs mag, ang), -1: no coupling
    int32_t steps;
    int32_t interleaved;
};


// Tile kernel (channels <= kCouplingTileChannels): one workgroup takes `width` bins (256 ... 1024, more for
// fewer channels so that every thread keeps several loads in flight) of one packet, stages all channels in
// LDS as [channel][width + 1] (coalesced global reads in either source layout), runs the coupling steps
// column by column, and writes every planar row once.  HBM traffic: each value read once, written once.
constexpr int kCouplingTileChannels = 32;
__host__ __device__ inline int coupling_tile_width(int channels) { return channels <= 4 ? 1024 : (channels <= 8 ? 512 : 256); }

__global__ __launch_bounds__(256) void coupling_tile_kernel(const CouplingPacket *__restrict__ pkts,
                                                           const uint8_t *__restrict__ steps, int channels,
                                                           const float *__restrict__ residue, float *__restrict__ temp)
{
    extern __shared__ float s_tile[];  // [channels][width + 1]
    const CouplingPacket pk = pkts[blockIdx.y];
    const int width = coupling_tile_width(channels), ld = width + 1;
    const int bin0 = blockIdx.x * width;
    if (bin0 >= pk.half) return;
    const int nb = min(width, pk.half - bin0);
    const int t = threadIdx.x;
    const float *src = residue + pk.src_off;
    if (pk.interleaved) {
        const float *s = src + (size_t)bin0 * channels;
        const int total = nb * channels;
        // element i = (bin, channel) = (i / channels, i % channels); i advances by 256 per step
        int bin = t / channels, c = t - bin * channels;
        const int dbin = 256 / channels, dc = 256 - dbin * channels;
#pragma unroll 8
        for (int i = t; i < total; i += 256) {
            s_tile[c * ld + bin] = s[i];
            bin += dbin;
            c += dc;
            if (c >= channels) { c -= channels; ++bin; }
        }
        __syncthreads();
    } else {
        for (int c = 0; c < channels; ++c)
#pragma unroll 4
            for (int b = t; b < nb; b += 256) s_tile[c * ld + b] = src[(size_t)c * pk.half + bin0 + b];
    }
    if (pk.steps_off >= 0) {
        const uint8_t *st = steps + pk.steps_off;
        for (int i = pk.steps - 1; i >= 0; --i) {  // reverse order, Mapping.cs:166
            float *pm = s_tile + st[2 * i] * ld, *pa = s_tile + st[2 * i + 1] * ld;
            for (int b = t; b < nb; b += 256) {
                float m = pm[b], a = pa[b];
                couple(m, a);
                pm[b] = m;
                pa[b] = a;
            }
        }
    }
    float *dst = temp + pk.dst_off + bin0;
    for (int c = 0; c < channels; ++c)
#pragma unroll 4
        for (int b = t; b < nb; b += 256) dst[(size_t)c * pk.half + b] = s_tile[c * ld + b];
}

// Fallback for more channels than the tile holds: one thread per (packet, bin), through global memory.
__global__ __launch_bounds__(256) void coupling_kernel(const CouplingPacket *__restrict__ pkts,
                                                      const uint8_t *__restrict__ steps,
                                                      int channels, const float *__restrict__ residue,
                                                      float *__restrict__ temp, int max_half)
{
    const CouplingPacket pk = pkts[blockIdx.y];
    const int bin = blockIdx.x * 256 + threadIdx.x;
    if (bin >= pk.half) return;
    const float *src = residue + pk.src_off;
    float *dst = temp + pk.dst_off;
    for (int c = 0; c < channels; ++c) {
        float v = pk.interleaved ? src[(size_t)bin * channels + c] : src[(size_t)c * pk.half + bin];
        dst[(size_t)c * pk.half + bin] = v;
    }
    if (pk.steps_off < 0) return;
    const uint8_t *st = steps + pk.steps_off;
    for (int i = pk.steps - 1; i >= 0; --i) {  // reverse order, Mapping.cs:166
        float *pm = dst + (size_t)st[2 * i] * pk.half + bin;
        float *pa = dst + (size_t)st[2 * i + 1] * pk.half + bin;
        float m = *pm, a = *pa;
        couple(m, a);
        *pm = m;
        *pa = a;
    }
}


// ---------------------------------------------------------------------------------------------
// floor1_unwrap_kernel: everything serial about Floor1, one LANE per channel-record (kUnwrapRecs records per wavefront):
// Floor1.UnwrapPosts (Floor1.cs:270-353) and the walk over the posts in X order that picks the ones a line is drawn
// to (Floor1.cs:236-252: post 0 and every post whose step flag is set).  Output per record: the active posts in X
// order as x | (finalY * multiplier) << 16 (Apply's `* _multiplier`, Floor1.cs:237,245) and their count -- 0 when
// ExecuteChannel is false.  Integers only, bit-exact; the curve itself is rendered by whoever consumes the posts
// (synth_kernel in LDS, floor1_render_kernel into memory).
// rec_info[rec]: floor index in bits 0..5, bit 6 = type-0 floor, bit 7 = long block.  A type-0 record gets the
// one-post curve "index 255 everywhere" (table[255] == 1.0): floor0_apply_kernel has multiplied its spectrum already.
// dbg_y / dbg_f (test entry only): finalY * multiplier and the step flags per post, [rec][64].
// ---------------------------------------------------------------------------------------------
constexpr int kPrepFloorsInLds = 4;


// kFloorsInLds: the decoder's floors all fit the LDS copy (the usual case: a stream has one or two); otherwise they are
// read from memory.  Two instantiations rather than one pointer that may point either way: that would be a FLAT
// access, slow and -- in a chain of dependent steps -- waited for at every step.
// kUnwrapRecs records per wavefront, one lane each (the other lanes only help to move data): the kernel is a single
// round of latency -- staging, the walk, copy-out -- so a smaller tile means less of the first and the last per wave,
// and twice the wavefronts fit (10 KB of LDS each).
constexpr int kUnwrapRecs = 64;
constexpr int kUnwrapWaves = 4;  // wavefronts per workgroup: the dispatcher's workgroup rate is a visible part of so short a kernel
template <bool kFloorsInLds>
__global__ __launch_bounds__(64 * kUnwrapWaves) void floor1_unwrap_kernel(int n_rec, const int16_t *__restrict__ posts,
                                                          const uint8_t *__restrict__ post_counts,
                                                          const uint8_t *__restrict__ rec_info,
                                                          const FloorDev *__restrict__ g_floors, int n_floors,
                                                          int32_t *__restrict__ cposts, uint8_t *__restrict__ ccount,
                                                          int16_t *__restrict__ dbg_y, uint8_t *__restrict__ dbg_f,
                                                          unsigned long long *stamps, int f0_fused)
{
#ifdef VPZ_STAMPS
    unsigned long long t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
    constexpr int R = kUnwrapRecs;
    // finalY of the tile's records, [post][record]; 16 bits hold every value a valid packet can produce (below
    // 2 * range) -- beyond that the reference indexes outside its dB table anyway
    __shared__ int16_t s_y_all[kUnwrapWaves][64][R + 2];
    __shared__ int16_t s_posts_all[kUnwrapWaves][R][66];  // raw posts, staged with coalesced loads (a row is 33 words)
    __shared__ FloorDev s_floors[kFloorsInLds ? kPrepFloorsInLds : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int16_t (*s_y)[R + 2] = s_y_all[wave];
    int16_t (*s_posts)[66] = s_posts_all[wave];
    const int first = (blockIdx.x * kUnwrapWaves + wave) * R;
    const int rec = first + lane;
    const bool mine = lane < R && rec < n_rec;
    // (asked for up front: read after the staging they would be two more memory round trips in a row)
    const int my_count = mine ? post_counts[rec] : 0;
    const uint8_t info = mine ? rec_info[rec] : 0;
    // stage the inputs: a per-lane walk over global memory would put one DRAM round trip on every post.  Every load
    // -- the floors' tables included -- is in flight before the first LDS store.
    {
        constexpr int kFloorWords = (int)(sizeof(FloorDev) / 4);
        constexpr int kFloorLoads = kFloorsInLds ? (kPrepFloorsInLds * kFloorWords + 63) / 64 : 1;
        constexpr int kPostLoads = R * 32 / 64;
        uint32_t fw[kFloorLoads];
        const int words = kFloorsInLds ? n_floors * kFloorWords : 0;
#pragma unroll
        for (int j = 0; j < kFloorLoads; ++j) {  // (every wave loads and stores the same words: simpler than a split)
            const int i = lane + 64 * j;
            fw[j] = i < words ? reinterpret_cast<const uint32_t *>(g_floors)[i] : 0u;
        }
        uint32_t v[kPostLoads];
#pragma unroll
        for (int it = 0; it < kPostLoads; ++it) {
            const int i = lane + it * 64;
            const int r = i >> 5, w = i & 31;
            const int rr = first + r < n_rec ? first + r : n_rec - 1;  // (the last tile re-reads its last record)
            v[it] = reinterpret_cast<const uint32_t *>(posts + (size_t)rr * 64)[w];
        }
#pragma unroll
        for (int j = 0; j < kFloorLoads; ++j) {
            const int i = lane + 64 * j;
            if (i < words) reinterpret_cast<uint32_t *>(s_floors)[i] = fw[j];
        }
#pragma unroll
        for (int it = 0; it < kPostLoads; ++it) {
            const int i = lane + it * 64;
            const int r = i >> 5, w = i & 31;
            reinterpret_cast<uint32_t *>(&s_posts[r][0])[w] = v[it];
        }
    }
    __syncthreads();
    VPZ_STAMP(0);
    // The active posts leave through LDS (the raw posts' area, free once the walk is done; 33 words per record, the first
    // 32 staged): written by their lane one at a time straight to memory they are scattered 4-byte stores, and the
    // address unit takes those one lane per cycle -- that was a third of this kernel's time.
    int32_t *s_out = reinterpret_cast<int32_t *>(&s_posts[0][0]) + (lane < R ? lane : 0) * 33;
    int count = 0;
    if (my_count != 0) {
        int32_t *row = cposts + (size_t)rec * 64;  // this lane's record: one 256-byte row, front to back
        if (info & 0x40) {
            if (f0_fused) {  // type-0 floor applied by the stereo fast path itself: the marker, and which floor it is
                s_out[0] = info & 0x3F;
                count = kFloor0Marker;
            } else {  // type-0 floor already applied (floor0_apply_kernel on the planar temp): the curve is 1.0
                s_out[0] = 255 << 16;
                count = 1;
            }
        } else {
            const int fi = info & 0x3F;
            const FloorDev &f = kFloorsInLds ? s_floors[fi] : g_floors[fi];  // (folds to one address space)
            const int pc = f.x_count;  // Unpack leaves PostCount == xList.Length or 0 (Floor1.cs:173-218)
            const int range = f.range, mult = f.multiplier;
            const int16_t *p = s_posts[lane];
            auto clamp16 = [](int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); };
            unsigned long long flags = 3ull;  // stepFlags, one bit per post: posts 0 and 1 are always set (:283-284)
            s_y[0][lane] = p[0];
            s_y[1][lane] = p[1];
            // the walk is a chain of dependent LDS round trips (a step reads what earlier steps wrote): everything
            // that does not depend on the packet -- neighbours, distances -- and the raw post are fetched a step ahead,
            // and a step has no branch (Floor1.cs:286-352, every arm computed, the result selected)
            uint32_t st0 = f.step[2][0], st1 = f.step[2][1];
            int val_next = p[2];
            for (int i = 2; i < pc; ++i) {
                const uint32_t c0 = st0, c1 = st1;
                const int val = val_next;
                const int nx = i + 1 < pc ? i + 1 : i;
                st0 = f.step[nx][0];
                st1 = f.step[nx][1];
                val_next = p[nx];
                const int lo = c0 & 0xFF, hi = (c0 >> 8) & 0xFF;
                // Floor1.RenderPoint (Floor1.cs:355-370) with its two x differences tabulated
                const int y0 = s_y[lo][lane], y1 = s_y[hi][lane];
                const int dy = y1 - y0;
                const int off = div_floor_small(iabs(dy) * (int)(c0 >> 16), (int)(int16_t)(c1 & 0xFFFF));
                const int predicted = dy < 0 ? y0 - off : y0 + off;
                const int highroom = range - predicted;
                const int lowroom = predicted;
                const int room = (highroom < lowroom ? highroom : lowroom) * 2;
                const int big = (highroom > lowroom) ? val - lowroom + predicted : predicted - val + highroom - 1;
                const int small = ((val % 2) == 1) ? predicted - ((val + 1) / 2) : predicted + (val / 2);
                const int result = val != 0 ? (val >= room ? big : small) : predicted;  // (val == 0: flag stays clear, :344-347)
                const unsigned long long touched = (1ull << lo) | (1ull << hi) | (1ull << i);
                flags |= val != 0 ? touched : 0ull;
                s_y[i][lane] = (int16_t)clamp16(result);
            }
            VPZ_STAMP(1);
            // Floor1.cs:236-252: post 0, then every flagged post in X order; Apply multiplies by _multiplier (:237,245).
            // Without a branch: a post that is not drawn to is written to where the next one goes (slot 32 of the 33:
            // nobody reads it), so the LDS reads of several posts can be in flight together.
#pragma unroll 8
            for (int i = 0; i < pc; ++i) {
                const uint32_t so = f.sorted[i];
                const int idx = so & 0xFF;
                const bool active = i == 0 || ((flags >> idx) & 1ull);
                const int v = (int)(so >> 16) | (clamp16(s_y[idx][lane] * mult) << 16);
                s_out[count < 32 ? count : 32] = v;
                if (active && count >= 32) row[count] = v;  // (more than 32 active posts: the rest goes out directly)
                count += active ? 1 : 0;
            }
            if (dbg_y) {
                for (int i = 0; i < 64; ++i) {
                    dbg_y[(size_t)rec * 64 + i] = i < pc ? (int16_t)clamp16(s_y[i][lane] * mult) : (int16_t)0;
                    dbg_f[(size_t)rec * 64 + i] = i < pc ? (uint8_t)((flags >> i) & 1ull) : (uint8_t)0;
                }
            }
        }
    }
    VPZ_STAMP(2);
    if (mine) ccount[rec] = (uint8_t)count;
    // two records per pass: lanes 0..31 write the staged posts of one, lanes 32..63 of the next -- 128 contiguous bytes
    // each; the LDS reads of eight passes are issued together
    uint8_t *s_cnt = reinterpret_cast<uint8_t *>(&s_y[0][0]);  // (the walk's area is free now)
    __builtin_amdgcn_wave_barrier();
    if (lane < R) s_cnt[lane] = (uint8_t)(count < 32 ? count : 32);
    __builtin_amdgcn_wave_barrier();
    const int half = lane >> 5, c = lane & 31;
    for (int it0 = 0; it0 < R / 2; it0 += 8) {
        int32_t v[8];
        int cn[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = 2 * (it0 + u) + half;
            v[u] = reinterpret_cast<const int32_t *>(&s_posts[0][0])[r * 33 + c];
            cn[u] = s_cnt[r];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = 2 * (it0 + u) + half;
            if (c < cn[u]) cposts[(size_t)(first + r) * 64 + c] = v[u];
        }
    }
    VPZ_STAMP(3);
#ifdef VPZ_STAMPS
    if (stamps && lane == 0) {
        for (int k = 0; k < 4; ++k) atomicAdd(&stamps[k], t_acc[k]);
        atomicAdd(&stamps[15], 1ull);
    }
#endif
}

// floor1_render_kernel: the curve of every record as one table index per bin in memory, curve_y[rec][half1] -- for
// the any-block-size path (generic_floor_kernel reads it) and for the test entry that reads the integers back.
// One wavefront per record, the same render_floor_indices the fused kernel runs in LDS.
constexpr int kRenderWaves = 4;
__global__ __launch_bounds__(64 * kRenderWaves) void floor1_render_kernel(int n_rec, const int32_t *__restrict__ cposts,
                                                                         const uint8_t *__restrict__ ccount,
                                                                         const uint8_t *__restrict__ rec_info,
                                                                         int half0, int half1,
                                                                         uint8_t *__restrict__ curve_y)
{
    // (the closed-form render works in LDS: 256 ints of curve, then its tables; the walk writes to memory directly)
    constexpr int kAuxInts = (render_aux_ints(128) > 256 + kRenderFastAuxInts ? render_aux_ints(128) : 256 + kRenderFastAuxInts) + 3 & ~3;
    __shared__ __attribute__((aligned(16))) int s_aux[kRenderWaves][kAuxInts];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rec = blockIdx.x * kRenderWaves + wave;
    if (rec >= n_rec) return;
    const int m = ccount[rec];
    if (m == 0) return;
    const int cp = lane < m ? cposts[(size_t)rec * 64 + lane] : 0;
    const int n = (rec_info[rec] & 0x80) ? half1 : half0;
    uint8_t *row = curve_y + (size_t)rec * half1;
    if (n <= 1024) {
        if (render_floor_indices_fast(reinterpret_cast<uint8_t *>(s_aux[wave]), s_aux[wave] + 256, n, n, cp, m, lane)) {
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < (n >> 2); i += 64) reinterpret_cast<uint32_t *>(row)[i] = (uint32_t)s_aux[wave][i];
        } else {
            render_floor_indices<32>(row, s_aux[wave], n, n, cp, m, lane);
        }
    } else {
        render_floor_indices<128>(row, s_aux[wave], n, n, cp, m, lane);
    }
}


// Group mode: every packet goes through the group's LDS rows, so the way it comes out of HBM is free to choose --
// 16-byte pieces, four per lane, kept in the registers a spectrum would occupy (x[2j], x[2j+1]):
//   Residue2-interleaved packet ([half][C] floats, Residue2.cs:31-34): the C waves split the vector, piece
//     q = lane + 64*w + 64*C*j (w: wave in the group) -- every wave-load is one contiguous 1 KiB span;
//   planar packet: the wave takes its own channel's row, piece q = lane + 64*j.
// The loads are UNCONDITIONAL (pieces past the end re-read the last one; staging drops them): a load under a lane
// condition needs a select on its result, and that select would make the wave wait for the data right here --
// which is the one place where it must not.
__device__ __forceinline__ void load_group_share(float2 (&x)[8], const VPZ_GLOBAL float *src, int first, int step, int limit,
                                                 int lane)
{
    const VPZ_GLOBAL float4 *s4 = reinterpret_cast<const VPZ_GLOBAL float4 *>(src);
    asm volatile("" : "+v"(lane));  // the piece numbers are frame-invariant: keep them out of long-lived registers
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = min(lane + first + step * j, limit - 1);
        const float4 v = s4[q];
        x[2 * j] = make_float2(v.x, v.y);
        x[2 * j + 1] = make_float2(v.z, v.w);
    }
}

// ... and the de-interleave (Residue2.cs:42-51) into the group's LDS rows: element e = bin * C + channel goes to
// rows[channel][bin].  `magic` = ceil(2^18 / C): e / C == (e * magic) >> 18 for every e < 8192, C <= 8.
// upper == false (a 2048 block whose residue's support ends in the lower half, ABI v4): only the first two pieces of every lane
// hold data -- pieces 64 w + 64 C j, j < 2, are exactly the lower half of every channel's row
// `pre`: for a 2048 block, the four row positions (floats from `rows`) of a lane's elements 4 q0 + i -- they do not depend on the
// frame, the kernel computes them once per run (stage_positions) and keeps exactly these four registers across the loop
__device__ __forceinline__ void stage_positions(int (&pre)[4], int C, uint32_t magic, int w, int lane)
{
    const int q0 = lane + 64 * w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t e = 4u * (uint32_t)q0 + (uint32_t)i;
        const uint32_t bin = __umul24(e, magic) >> 18;  // (24-bit multiplies are full rate, 32-bit ones a quarter)
        const uint32_t c = e - __umul24(bin, (uint32_t)C);
        pre[i] = (int)(__umul24(c, (uint32_t)kWaveBufFloats) + bin);
    }
}
__device__ __forceinline__ void stage_interleaved(const float2 (&x)[8], float *rows, int C, uint32_t magic, int half,
                                                  int w, int lane, bool upper, const int (&pre)[4])
{
    const int total4 = (C * half) >> 2;
    // The LDS addresses below do not depend on the frame: left alone, the compiler computes them once before the
    // frame loop and then has to park them in scratch memory.  An opaque lane id keeps them inside the loop.
    asm volatile("" : "+v"(lane));
    const int q0 = lane + 64 * w;
    // a 2048-sample block has all four pieces of every lane (4 * 64 * C pieces): no lane condition per store -- sixteen
    // `saveexec / branch / restore` sequences were most of what staging cost
    const bool full = half == 1024;
    if (C == 2) {  // (L R L R): two bins of each channel, 8-byte stores
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + 128 * j;
            if ((full && (upper || j < 2)) || (!full && q < total4)) {
                reinterpret_cast<float2 *>(rows)[q] = make_float2(x[2 * j].x, x[2 * j + 1].x);
                reinterpret_cast<float2 *>(rows + kWaveBufFloats)[q] = make_float2(x[2 * j].y, x[2 * j + 1].y);
            }
        }
        return;
    }
    // piece j holds elements 4*q0 + 256*C*j + i: the SAME channel as element 4*q0 + i, 256*j bins further on -- one
    // division per i, and the four pieces differ by a constant offset
    if (full) {
        float *dst[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = rows + pre[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 2 || upper) {
                dst[0][256 * j] = x[2 * j].x;
                dst[1][256 * j] = x[2 * j].y;
                dst[2][256 * j] = x[2 * j + 1].x;
                dst[3][256 * j] = x[2 * j + 1].y;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t e = 4u * (uint32_t)q0 + (uint32_t)i;
        const uint32_t bin = __umul24(e, magic) >> 18;
        const uint32_t c = e - __umul24(bin, (uint32_t)C);
        float *dst = rows + __umul24(c, (uint32_t)kWaveBufFloats) + bin;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = (i & 2) ? ((i & 1) ? x[2 * j + 1].y : x[2 * j + 1].x) : ((i & 1) ? x[2 * j].y : x[2 * j].x);
            if (q0 + 64 * C * j < total4) dst[256 * j] = v;
        }
    }
}

// ... or, for a planar packet, the wave's own channel straight into its row
// Group mode, SynthArgs.group_dma: the packet lies in the group's rows as it came -- [bin][C], C even, i.e. C/2 float2 per bin
// (stage_by_lds_dma below) -- and wave w takes the lane's 8 points of ITS channel out of it: the float2 that holds the channel,
// and the one that holds the other channel of w's coupling step (`partner`, -1: none; usually the same float2), with the step
// applied to the values in registers (Mapping.cs:166-225; a channel is in one step at most, so there is no order to keep).
// Point kk = kk0 + kst * m is bins 2kk, 2kk + 1.  No ds_write, no coupling pass over LDS -- but 8-byte reads at a stride of C
// dwords (2-way bank conflicts for C = 6) and every wave of a pair runs the step: measured SLOWER than the product's staging
// (configs[3] 0.353 against 0.321 ms), hence opt-in only (VPZ_GROUP_DMA=1).
__device__ __forceinline__ void pickup_interleaved(float2 (&x)[8], const float *rows, int C, int w, int partner, bool is_mag,
                                                   int kk0, int kst)
{
    const float2 *v2 = reinterpret_cast<const float2 *>(rows);
    const int hc = C >> 1;
    const bool odd = w & 1, podd = partner & 1;
    const float2 *own = v2 + (w >> 1) + 2 * kk0 * hc;
    const float2 *oth = v2 + (partner >= 0 ? partner >> 1 : 0) + 2 * kk0 * hc;
    const bool same = partner >= 0 && (partner >> 1) == (w >> 1);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int o = 2 * kst * m * hc;
        const float2 a0 = own[o], a1 = own[o + hc];
        float s0 = odd ? a0.y : a0.x, s1 = odd ? a1.y : a1.x;
        if (partner >= 0) {  // (wave-uniform)
            float p0, p1;
            if (same) {
                p0 = odd ? a0.x : a0.y;
                p1 = odd ? a1.x : a1.y;
            } else {
                const float2 b0 = oth[o], b1 = oth[o + hc];
                p0 = podd ? b0.y : b0.x;
                p1 = podd ? b1.y : b1.x;
            }
            if (is_mag) { couple(s0, p0); couple(s1, p1); }
            else { couple(p0, s0); couple(p1, s1); }
        }
        x[m] = make_float2(s0, s1);
    }
}

// ... landed there by LDS-DMA (`global_load_lds_dwordx4`: 16 bytes per lane straight from memory into LDS, a wave-load's 64
// pieces back to back; no registers, no ds_write), wave w of the C taking pieces 64 w + lane + 64 C j.  The rows must have
// been given up by the previous frame (they are one landing area: a piece lands in whichever row it falls into); the wave
// waits for its pieces here -- nothing is in flight a frame ahead.  Also the timing experiment of VPZ_SYNTH_ABLATE bit 2048
// (product pick-up on the landed vector: wrong results).
__device__ __forceinline__ void stage_by_lds_dma(const float *src, float *rows, int C, int half, int w, int lane)
{
    const int total4 = (C * half) >> 2;  // 16-byte pieces of the packet
    typedef __attribute__((address_space(1))) const void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    asm volatile("" : "+v"(lane));
    for (int g = 64 * w; g < total4; g += 64 * C)
        __builtin_amdgcn_global_load_lds((gvoid *)(reinterpret_cast<const float4 *>(src) + g + lane), (lvoid *)(rows + 4 * g), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ... and as a GATHER (bit 4096; results are right): wave w asks for ITS channel's bins one dword per lane -- lane i of piece m
// reads element (64 m + i) * C + w -- and LDS-DMA lands a wave-load's 64 dwords back to back: the row comes out de-interleaved,
// no register, no ds_write, no strided LDS read.
__device__ __forceinline__ void stage_by_lds_dma_gather(const float *src, float *row, int C, int half, int w, int lane)
{
    typedef __attribute__((address_space(1))) const void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    asm volatile("" : "+v"(lane));
    const float *p = src + (size_t)lane * C + w;
    for (int b = 0; b < half; b += 64) {
        if (b + lane < half) __builtin_amdgcn_global_load_lds((gvoid *)(p + (size_t)b * C), (lvoid *)(row + b), 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void stage_planar(const float2 (&x)[8], float *row, int half, int lane, bool upper = true)
{
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = lane + 64 * j;
        if ((half == 1024 && (upper || j < 2)) || (half != 1024 && q < (half >> 2)))  // (a long block has every piece: one wave-uniform test, no lane mask)
            reinterpret_cast<float4 *>(row)[q] = make_float4(x[2 * j].x, x[2 * j].y, x[2 * j + 1].x, x[2 * j + 1].y);
    }
}

// Group mode, interleaved output of more than two channels, long block after long block: the C waves have left the
// packet's 1024 output samples per channel in their rows; together they write the packet's [1024][C] block as dense
// 16-byte pieces, piece q = lane + 64*w + 64*C*j -- the de-interleave of stage_interleaved run backwards (element
// e = 4q + i is sample e / C of channel e % C, and piece j + 1 is the same channel 256 samples on).  A wave on its own
// can only scatter 4-byte stores at a stride of C samples: 6 channels took 1.7 x the time of planar output that way.
template <bool kS16>
__device__ __forceinline__ void emit_interleaved_rows(const float *rows, void *out, int C, uint32_t magic, int w, int lane,
                                                      int samples = 1024)
{
    asm volatile("" : "+v"(lane));
    const int q0 = lane + 64 * w;
    const bool full = samples == 1024;  // (a batch of short blocks holds 128 samples per block: its tail pieces are tested)
    if (!kS16) {
        const int total4 = (samples * C) >> 2;
        float v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t e = 4u * (uint32_t)q0 + (uint32_t)i;
            const uint32_t smp = __umul24(e, magic) >> 18;
            const uint32_t c = e - __umul24(smp, (uint32_t)C);
            const float *src = rows + __umul24(c, (uint32_t)kWaveBufFloats) + smp;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j][i] = (full || q0 + 64 * C * j < total4) ? src[256 * j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (full || q0 + 64 * C * j < total4)
                store_pcm4(reinterpret_cast<float4 *>(out) + q0 + 64 * C * j, make_float4(v[j][0], v[j][1], v[j][2], v[j][3]));
    } else {
        const int total8 = (samples * C) >> 3;
        float v[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t e = 8u * (uint32_t)q0 + (uint32_t)i;
            const uint32_t smp = __umul24(e, magic) >> 18;
            const uint32_t c = e - __umul24(smp, (uint32_t)C);
            const float *src = rows + __umul24(c, (uint32_t)kWaveBufFloats) + smp;
#pragma unroll
            for (int j = 0; j < 2; ++j) v[j][i] = (full || q0 + 64 * C * j < total8) ? src[512 * j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (full || q0 + 64 * C * j < total8)
                store_nt(reinterpret_cast<uint4 *>(out) + q0 + 64 * C * j, pack_s16(v[j][0], v[j][1]), pack_s16(v[j][2], v[j][3]),
                         pack_s16(v[j][4], v[j][5]), pack_s16(v[j][6], v[j][7]));
    }
}

// kOut: 0 planar output, 1 interleaved (any channel count: every wave scatters its own channel),
//       2 interleaved stereo: the two waves of a stream (channels 0 / 1, adjacent in the workgroup) build
//         their blocks, meet at a workgroup barrier, and each writes HALF of the packet's samples for BOTH
//         channels -- dense 32-byte (L R L R | L R L R) stores instead of 4-byte stores at an 8-byte stride.
// kGeneral: block sizes from {256, 512, 1024, 2048} in any combination (speech-rate and low-bitrate streams use
//       512 / 1024); the plain variant is the 256 / 2048 kernel with its geometry folded at compile time.
// kGroup: the waves of one run's channels (<= 8) sit in one workgroup and work on the same packet at the same
//       time: the packet -- planar or the Residue2-interleaved vector -- is staged in their LDS rows, de-interleaved
//       on the way (Residue2.cs:42-51), the coupling steps run there in reverse order (Mapping.cs:166-172), and
//       every wave then picks its own channel up.  No de-interleaved / de-coupled copy of the residue exists in HBM.
// Floor: the wave renders its channel's curve (Floor1.cs:236-262, 372-397) from the record's active posts into its
//       LDS row as one table index per bin, right before the row is needed for anything else.
// kS16: PCM leaves as 16-bit samples (to_s16) instead of float32; offsets and strides count samples either way.
// The steady state of a stream (2048 after 2048, long windows): tested field by field here.  The descriptor carries it as ONE
// bit too (kFrameSteady, which the stereo fast path tests) -- measured in this kernel, the bit test is SLOWER: configs[3] 0.265 ms
// with the eight field tests, 0.274 with the bit (three alternating rounds on one box, profiles/r4_ab_steady_bit.txt): the
// compiler folds the geometry it has just compared into the branch's address arithmetic.  -DVPZ_GROUP_STEADY_BIT: the bit.
#ifdef VPZ_GROUP_STEADY_BIT
#define VPZ_STEADY(fd) ((fd).flags & kFrameSteady)
#else
#define VPZ_STEADY(fd) (is_long && ((fd).flags & kFrameSlope1) && (fd).left_start == 0 && plen == 1024 && (fd).prev_end == 1024 && (fd).out_count == 1024 && k_size1 == 2048)
#endif
template <bool kHasFloor, int kOut, bool kGeneral, bool kGroup, bool kS16>
__global__ __launch_bounds__(kSynthThreads, 4) void synth_kernel(SynthArgs a)
{
    using out_t = typename std::conditional<kS16, int16_t, float>::type;
    constexpr bool kInterleaved = kOut != 0;
    constexpr bool kPair = kOut == 2;
    constexpr bool kSync = kPair || kGroup;  // the workgroup's waves run their loops in lock step
    constexpr bool kBatchShort = kGroup && !kGeneral && kHasFloor;  // batches of short blocks (see the run builder)
    constexpr int kRunMax = kGeneral ? kMaxRunLengthGeneral : kMaxRunLength;
    __shared__ int s_iters;
    // tables: the plain variant keeps exactly what 2048 / 256 need; the general one holds the whole fast table
    // of the larger size (tw 512 | twAB 512 | twBC 64) and a compacted one of the smaller (256 | 256 | 64)
    __shared__ float2 s_twL[kGeneral ? 1088 : 512];
    __shared__ float2 s_twAB[kGeneral ? 1 : 512];
    __shared__ float2 s_twBC[kGeneral ? 1 : 64];
    __shared__ float2 s_twS[kGeneral ? 576 : 64];
    __shared__ float s_slope1[1024];
    __shared__ float s_slope0[kGeneral ? 512 : 128];
    __shared__ float s_db[kHasFloor ? 256 : 1];
    __shared__ __attribute__((aligned(8))) uint8_t s_steps[kGroup ? 2 * kGroupMaxStepPairs + 8 : 8];
    __shared__ PacketGeom s_geom[8];
    __shared__ float s_work[kSynthWaves][kWaveBufFloats];   // h of the block being built
    __shared__ float s_tail[kSynthWaves][kWaveTailFloats];  // upper half of the previous block's h
    __shared__ uint4 s_desc[kSynthWaves][(kRunMax + 1) * 2];  // the run's frame descriptors

    const int lane = threadIdx.x & 63;
    // wave-uniform values are forced into SGPRs so the descriptor reads become scalar loads
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int C = a.channels;
    // which run and channel this wave owns.  Free-running waves are numbered through the grid; in group mode a
    // workgroup holds floor(8 / C) whole runs (the remaining waves idle but keep the barriers matched).
    int run_idx, ch, gw0 = 0;  // gw0: first wave of this wave's group
    bool active;
    if (kGroup) {
        // (uniform values: keep them out of the vector registers the integer division would leave them in)
        const int groups = __builtin_amdgcn_readfirstlane(kSynthWaves / C);
        // (waves that own no channel -- 8 - groups * C of them -- idle at the barriers; spreading them over the SIMDs by
        // rotating the busy set in every other workgroup was measured: no difference)
        const int slot = __builtin_amdgcn_readfirstlane(wave / C);
        ch = wave - slot * C;
        gw0 = slot * C;
        run_idx = blockIdx.x * groups + slot;
        active = slot < groups && run_idx < a.n_runs;
    } else {
        const int item = blockIdx.x * kSynthWaves + wave;
        active = item < a.n_runs * C;
        run_idx = active ? item / C : 0;
        ch = active ? item - run_idx * C : (wave & 1);
    }
    const uint32_t div_magic =
        kGroup ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(((1u << 18) + (uint32_t)C - 1u) / (uint32_t)C)) : 0u;
    if (!active) { run_idx = 0; if (kGroup) ch = 0; }
    RunDesc run = a.runs[run_idx];
    if (kSync && !active) {  // an idle wave: nothing to load, emit or save
        run.count = 0;
        run.pre_kind = kPreNone;
        run.flags = 0;
    }
    // The run record and a compact run's bytes sit in pinned HOST memory (read in place over the link: a round trip of
    // microseconds each): they are asked for here, ahead of the table staging below, so that the two round trips -- and
    // the post counts' -- pass while the tables load instead of after them.
    const int fi0 = (run.pre_kind == kPreRecompute && run.count > 0) ? -1 : 0;
    uint32_t cf_early = 0, mp_early = 0;
    int cc_early = 0;
    if (run.flags & kRunCompact) {
        const int n = run.count - fi0, f0 = run.first + fi0;
        if ((int)(threadIdx.x & 63) < n) {
            cf_early = a.cflags[f0 + (threadIdx.x & 63)];
            mp_early = a.cmap[f0 + (threadIdx.x & 63)];
            if (kHasFloor && a.ccount != nullptr) cc_early = a.ccount[run.rec_base + (int)(threadIdx.x & 63) * C + ch];
        }
    }
    if (kGeneral) {
        for (int i = threadIdx.x; i < kFastTableCount; i += kSynthThreads) s_twL[i] = a.tw_long[i];
        for (int i = threadIdx.x; i < 256; i += kSynthThreads) {
            s_twS[i] = a.tw_short[kFastTwOffset + i];
            s_twS[256 + i] = a.tw_short[kFastTwABOffset + i];
        }
        if (threadIdx.x < 64) s_twS[512 + threadIdx.x] = a.tw_short[kFastTwBCOffset + threadIdx.x];
    } else {
        const bool has_long = a.size1 == 2048 || a.size0 == 2048;
        const bool has_short = a.size0 == 256 || a.size1 == 256;
        for (int i = threadIdx.x; i < 512; i += kSynthThreads) {
            if (has_long) {
                s_twL[i] = a.tw_long[kFastTwOffset + i];
                s_twAB[i] = a.tw_long[kFastTwABOffset + i];
            }
        }
        const float2 *any = has_long ? a.tw_long : a.tw_short;
        if (threadIdx.x < 64) {
            s_twBC[threadIdx.x] = any[kFastTwBCOffset + threadIdx.x];
            if (has_short) s_twS[threadIdx.x] = a.tw_short[kFastTwOffset + threadIdx.x];
        }
    }
    for (int i = threadIdx.x; i < a.size1 / 2; i += kSynthThreads) s_slope1[i] = a.slope1[i];
    for (int i = threadIdx.x; i < a.size0 / 2 && i < (kGeneral ? 512 : 128); i += kSynthThreads) s_slope0[i] = a.slope0[i];
    if (kHasFloor && threadIdx.x < 256) s_db[threadIdx.x] = a.inv_db[threadIdx.x];
    if (threadIdx.x < 8) s_geom[threadIdx.x] = a.geom[threadIdx.x];
    if (kGroup)
        for (int i = threadIdx.x; i < 2 * a.n_step_pairs && i < 2 * kGroupMaxStepPairs; i += kSynthThreads)
            s_steps[i] = a.steps[i];
    __syncthreads();
    if (!kSync && !active) return;  // (the lock-step variants keep idle waves around for their barriers)

    // The scalars the frame loop needs, read from the kernel arguments ONCE and parked in the lanes of a vector register:
    // left as `a.x`, the compiler re-loads them from the argument segment inside the loop wherever it is short of scalar
    // registers -- a scalar load per use, each followed by an `s_waitcnt lgkmcnt(0)` that drains the wave's LDS queue as well
    // (round 3: 125 scalar loads per wave of this kernel, profiles/r3_isa_group.txt).  A lane of an opaque register comes
    // back with one v_readlane_b32 and no wait.  Pointers come back as GLOBAL pointers (a pointer rebuilt from two integers
    // has lost its address space: loads through it would be FLAT loads).
    int kv = 0;
    {
        const unsigned long long ps = reinterpret_cast<unsigned long long>(a.spec), pi = reinterpret_cast<unsigned long long>(a.inv_db),
                                 pc = reinterpret_cast<unsigned long long>(a.cposts), cs = (unsigned long long)a.channel_stride;
        const int vals[14] = {a.size0, a.size1, a.clip, a.max_steps, a.channels, a.ccount == nullptr ? 1 : 0, (int)(unsigned)ps, (int)(unsigned)(ps >> 32),
                              (int)(unsigned)pi, (int)(unsigned)(pi >> 32), (int)(unsigned)pc, (int)(unsigned)(pc >> 32), (int)(unsigned)cs, (int)(unsigned)(cs >> 32)};
#pragma unroll
        for (int i = 0; i < 14; ++i) kv = (int)(threadIdx.x & 63) == i ? vals[i] : kv;
        asm volatile("" : "+v"(kv));
    }
    auto parked64 = [&](int i) -> unsigned long long {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(kv, i + 1) << 32) | (unsigned)__builtin_amdgcn_readlane(kv, i);
    };
#define k_size0 __builtin_amdgcn_readlane(kv, 0)
#define k_size1 __builtin_amdgcn_readlane(kv, 1)
#define k_clip __builtin_amdgcn_readlane(kv, 2)
#define k_max_steps __builtin_amdgcn_readlane(kv, 3)
#define k_channels __builtin_amdgcn_readlane(kv, 4)
#define k_no_ccount __builtin_amdgcn_readlane(kv, 5)
#define k_spec reinterpret_cast<const VPZ_GLOBAL float *>(parked64(6))
#define k_inv_db reinterpret_cast<const VPZ_GLOBAL float *>(parked64(8))
#define k_cposts reinterpret_cast<const VPZ_GLOBAL int32_t *>(parked64(10))
#define k_channel_stride ((long long)parked64(12))
    const int half1 = k_size1 >> 1;

    float *hcur = s_work[wave];
    float *tail = s_tail[wave];
    int prev_n4 = 0;  // n/4 of the previous block (0: none yet)

    // which transform a frame takes depends on its block SIZE, not on its flag (size0 may be 2048 too)
    auto size_of = [&](uint32_t flags) -> int { return (flags & kFrameLong) ? k_size1 : k_size0; };
    // lanes per block of a frame; the plain variant only knows 2048 (64) and 256 (8)
    auto lpb_of = [&](uint32_t flags) -> int { return kGeneral ? (size_of(flags) >> 5) : (size_of(flags) == 2048 ? 64 : 8); };
    auto spectrum_of = [&](const FrameDesc &fd) -> const VPZ_GLOBAL float * {
        const int hh = (fd.flags & kFrameLong) ? (k_size1 >> 1) : (k_size0 >> 1);
        return k_spec + fd.spec_off + (int64_t)ch * hh;
    };
    // the raw input of a frame into registers: this wave's channel, or -- for an interleaved packet in group mode --
    // this wave's share of the packet.  cp: this lane's active floor post (lane < count).
    int cc_run = 0;  // lane i: active floor posts of this wave's channel in the run's i-th staged frame
    // the first four coupling steps of a frame's mapping (8 bytes; the host starts every mapping's steps on an 8-byte
    // boundary), read a frame ahead like the input: the step loop then runs out of scalar registers instead of three
    // dependent LDS round trips per step (step byte -> row address -> data)
    auto steps_word = [&](const FrameDesc &fd, bool valid) -> uint2 {
        const uint32_t off = valid ? 2u * ((fd.flags >> kFrameStepsOffShift) & kFrameStepsOffMask) : 0u;
        return *reinterpret_cast<const uint2 *>(s_steps + (off < 2u * kGroupMaxStepPairs ? off : 0u));
    };
    // Everything below is UNCONDITIONAL -- a frame that needs no input (none follows, a drain, a silent channel) reads
    // a few bytes of a table that always exists (the inverse dB table) instead: a load under a condition leaves the
    // compiler with a merge of "loaded" and "not loaded" registers, which it resolves with copies right behind the loads, and the copies wait
    // for the data.  That put the full memory latency in front of every frame of every variant with a floor.
    // ABI v4: does the residue's support reach the upper half of a frame's block?  If not (vpz_mapping_config.residue_end at or
    // below a quarter of the block size), the upper half is zeros by the setup header's word: in group mode a 2048 block's upper
    // half is neither loaded nor staged nor de-coupled, and nowhere is it looked up in the floor table or multiplied.  (Halves:
    // every decision is wave-uniform and the same in every wave of a group; a finer step would buy real streams nothing -- their
    // residues end in the top eighth of the block.)
    auto upper_of = [&](uint32_t flags) -> bool { return ((flags >> kFrameSkipShift) & kFrameSkipMask) < 4; };
    // ... and the 16-byte pieces of the packet there are to load then: half of them (whole wave-loads either way)
    auto support_pieces = [&](uint32_t flags, int vec_channels, int hh) -> int {
        const int total = (vec_channels * hh) >> 2;
        return (!kGeneral && (flags & kFrameLong) && hh == 1024 && !upper_of(flags)) ? total >> 1 : total;
    };
    auto prefetch = [&](const FrameDesc &fd, int slot, bool valid, float2 (&x)[8], int &cp, int &cnt, bool &ex) {
        cnt = valid ? __builtin_amdgcn_readlane(cc_run, slot) : 0;  // active posts of this wave's channel (0: silent)
        ex = valid && (k_no_ccount || (fd.flags & kFrameNoFloor) || cnt != 0);
        if (kGroup) {
            // (a batch of short blocks: their vectors lie back to back -- one packet of batch * 128 bins)
            const int hh = (size_of(fd.flags) >> 1) * (kBatchShort ? (int)((fd.flags >> kFrameBatchShift) & 7u) + 1 : 1);
            const bool shared_input = fd.flags & kFrameInterleaved;
            const VPZ_GLOBAL float *src = k_spec + fd.spec_off + (shared_input ? 0 : (int64_t)ch * hh);
            const bool regs = valid && !(shared_input && (VPZ_GROUP_DMA(a) || (VPZ_ABLATE(a) & (2048 | 4096))));  // (the packet comes by LDS-DMA)
            load_group_share(x, regs ? src : k_inv_db, shared_input ? 64 * ch : 0, shared_input ? 64 * C : 64,
                             regs ? support_pieces(fd.flags, shared_input ? C : 1, hh) : 1, lane);
        } else {
            load_spectrum(x, ex ? spectrum_of(fd) : k_inv_db, ex ? lpb_of(fd.flags) : 1, lane);
        }
        if (kHasFloor) {
            // all 64 entries of the record's row (the ones past `cnt` are dropped where the curve is rendered)
            const bool floored = ex && !(fd.flags & kFrameNoFloor);
            int l = lane;
            asm volatile("" : "+v"(l));  // keeps `cposts + lane` out of the registers that live across the frame loop
            cp = k_cposts[(size_t)(floored ? fd.rec + ch : 0) * 64 + l];
        }
    };
    bool batch_head = false, batch_member = false;
    // Stage the run's descriptors in LDS: per-frame scalar loads from global memory would put an L2 round trip on
    // every frame's critical path.  Explicit descriptors come with one coalesced read; a compact run builds them here.
    if (run.flags & kRunCompact) {
        const int n = run.count - fi0;  // staged frames (<= kRunMax + 1 <= 64), one lane each
        const uint32_t cf = cf_early, mp = mp_early;  // (lanes past the run hold zeros)
        const uint32_t pcf = __shfl_up(cf, 1);
        const PacketGeom g = s_geom[cf & 7], pg = s_geom[pcf & 7];
        const bool has_prev = lane > 0 || run.has_prev0;
        const int prev_end = lane > 0 ? pg.right_start : run.prev_end0;
        const int prev_stop = lane > 0 ? pg.right_end : run.prev_stop0;
        int left_start = has_prev ? g.left_start : g.right_start;  // StreamDecoder.cs:674 / :679
        int out_count = has_prev ? max(0, (int)g.right_start - (int)g.left_start) : 0;
        uint32_t fl = ((cf & 1) ? kFrameLong : 0u) | (g.left_use_size1 ? kFrameSlope1 : 0u) |
                      ((cf & kCfNoFloor) ? kFrameNoFloor : 0u);
        if (cf & kCfInterleaved) fl |= kFrameInterleaved;
        if (!(cf & kCfNoFloor)) {  // the mapping's coupling steps and what its setup header says about the residue's support
            const uint32_t mb = a.map_bits[mp];
            fl |= (mb & 0x00FFFF00u) | ((((cf & 1) ? mb >> kFrameSkipShift : mb >> kMapSkipShortShift) & kFrameSkipMask) << kFrameSkipShift);
        }
        if (cf & kCfSkip) { fl = kFrameDrain; out_count = 0; }
        if ((run.flags & kRunLastTrimmed) && lane == n - 1) { out_count = run.last_out_count; left_start = run.last_left_start; }
        // Batches of SHORT blocks (group mode, 256 / 2048 kernel): a short block costs a pass 70 % of what a long one
        // costs -- the transform computes eight copies of one block, the pass's fixed parts do not shrink -- for an
        // eighth of the samples, and real streams hold them in streaks of 3 to 12.  Up to eight consecutive short
        // blocks of a run go through ONE pass: their Residue2 vectors lie back to back, i.e. they de-interleave and
        // de-couple exactly like one long packet; then every lane group of eight takes one block (as imdct256_wave8
        // is laid out), and the pass emits up to 1 024 contiguous samples.  A block joins a batch if it is a plain
        // short-after-short step (all of the geometry below) of the same mapping as its predecessor; the first short
        // block after a long one goes alone (its overlap partner has the long block's shape).
        if (kBatchShort) {
            const uint32_t pmp = __shfl_up(mp, 1);
            // (the block before may be a short one -- its upper half overlaps -- or, for the head of a batch, a long one
            // with a short right window: the same 128 samples of overlap, further up in its tail)
            const bool after_short = prev_end == 128 && prev_stop == 256;
            const bool after_long = k_size1 == 2048 && prev_end == 1472 && prev_stop == 1600;
            const bool base_ok = lane < n && lane >= -fi0 && k_size0 == 256 && !(cf & 1) && (cf & kCfInterleaved) &&
                                 !(cf & (kCfNoFloor | kCfSkip)) && has_prev && out_count == 128 && left_start == 0 &&
                                 (after_short || after_long) && !a.no_batch;
            const bool base_prev = __shfl_up((int)base_ok, 1) != 0 && lane > 0 && after_short;
            const bool brk = !(base_ok && base_prev && mp == pmp);  // this frame does not continue its predecessor's streak
            const unsigned long long mask_brk = __ballot(brk);
            if (base_ok) {
                const unsigned long long below = mask_brk & ((2ull << lane) - 1ull);  // (never empty: lane 0 breaks)
                const int start = 63 - __clzll(below);
                const unsigned long long above = lane < 63 ? (mask_brk >> (lane + 1)) : 0ull;
                const int end = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
                const int pos = lane - start;
                if ((pos & 7) == 0) {
                    const int size = min(8, end - lane);
                    fl |= (uint32_t)(size - 1) << kFrameBatchShift;
                    batch_head = true;
                } else {
                    batch_member = true;
                }
            }
        }
        // residue and output offsets: exclusive prefix sums over the run's frames
        const int half = (cf & 1) ? (k_size1 >> 1) : (k_size0 >> 1);
        int spec_sz = lane < n ? C * half : 0;
        int out_sz = (lane < n && lane >= -fi0) ? out_count : 0;
        int spec_incl = spec_sz, out_incl = out_sz;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t0 = __shfl_up(spec_incl, d), t1 = __shfl_up(out_incl, d);
            if (lane >= d) { spec_incl += t0; out_incl += t1; }
        }
        if (lane < n) {
            const int64_t spec_off = run.spec_base + (spec_incl - spec_sz);
            const int64_t out_off = run.out_base + (out_incl - out_sz);
            uint4 lo, hi;
            lo.x = (uint32_t)spec_off; lo.y = (uint32_t)((uint64_t)spec_off >> 32);
            lo.z = (uint32_t)out_off; lo.w = (uint32_t)((uint64_t)out_off >> 32);
            hi.x = (uint32_t)(run.rec_base + lane * C);
            const int plen_d = (has_prev && !(cf & kCfSkip)) ? prev_stop - prev_end : 0, pend_d = (has_prev && !(cf & kCfSkip)) ? prev_end : 0;
            hi.y = (uint32_t)left_start | ((uint32_t)plen_d << 16);
            hi.z = (uint32_t)pend_d | ((uint32_t)out_count << 16);
            if (frame_is_steady(fl, a.size1, left_start, plen_d, pend_d, out_count)) fl |= kFrameSteady;
            hi.w = fl;
            s_desc[wave][2 * lane] = lo;
            s_desc[wave][2 * lane + 1] = hi;
        }
    } else {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.frames + (run.first + fi0));
        const int n16 = (run.count - fi0) * 2;
        for (int i = lane; i < n16; i += 64) s_desc[wave][i] = src[i];
    }
    // The run's post counts (ExecuteChannel of every frame for this wave's channel), one lane per frame, with ONE load
    // ahead of the loop: read per frame, the count would put a dependent global load -- and, vmcnt being in order, the
    // drain of the previous frame's PCM stores -- in front of every prefetch.
    if (kHasFloor && a.ccount != nullptr) {
        if (run.flags & kRunCompact) {
            cc_run = cc_early;
        } else {
            __builtin_amdgcn_wave_barrier();
            if (lane < run.count - fi0) cc_run = a.ccount[(int)s_desc[wave][2 * lane + 1].x + ch];
        }
    }
    // trip count: the run's own in the free-running variants, the workgroup's longest in the lock-step ones; the members
    // of a batch ride with its head
    (void)batch_head;
    int iters = run.count - fi0 - (kBatchShort ? (int)__popcll(__ballot(batch_member)) : 0);
    if (kSync) {
        if (threadIdx.x == 0) s_iters = 0;
        __syncthreads();
        if (lane == 0) atomicMax(&s_iters, iters);
        __syncthreads();
        iters = s_iters;
    }
    auto frame_at = [&](int fi) -> FrameDesc {  // broadcast LDS read, then into SGPRs
        const uint4 lo = s_desc[wave][(fi - fi0) * 2], hi = s_desc[wave][(fi - fi0) * 2 + 1];
        FrameDesc fd;
        const uint32_t w0 = __builtin_amdgcn_readfirstlane(lo.x), w1 = __builtin_amdgcn_readfirstlane(lo.y);
        const uint32_t w2 = __builtin_amdgcn_readfirstlane(lo.z), w3 = __builtin_amdgcn_readfirstlane(lo.w);
        const uint32_t w4 = __builtin_amdgcn_readfirstlane(hi.x), w5 = __builtin_amdgcn_readfirstlane(hi.y);
        const uint32_t w6 = __builtin_amdgcn_readfirstlane(hi.z), w7 = __builtin_amdgcn_readfirstlane(hi.w);
        fd.spec_off = (int64_t)(((uint64_t)w1 << 32) | w0);
        fd.out_off = (int64_t)(((uint64_t)w3 << 32) | w2);
        fd.rec = (int32_t)w4;
        fd.left_start = (uint16_t)(w5 & 0xFFFF);
        fd.packet_len = (uint16_t)(w5 >> 16);
        fd.prev_end = (uint16_t)(w6 & 0xFFFF);
        fd.out_count = (uint16_t)(w6 >> 16);
        fd.flags = w7;
        return fd;
    };

    // ---- block preceding the run: from the saved state, or recomputed as "frame -1" of the loop
    if (run.pre_kind == kPreState) {
        const float *st = a.state_h + (size_t)run.state_slot * a.state_slot_floats + ((size_t)run.stream * k_channels + ch) * half1;
        prev_n4 = run.prev_long ? (k_size1 >> 2) : (k_size0 >> 2);
        for (int i = lane; i < prev_n4; i += 64) tail[i] = st[i];
    }
    out_t *out_base = reinterpret_cast<out_t *>(a.out) + (a.stream_out_off ? a.stream_out_off[run.stream] : 0);
    {   // wave-uniform, but the offset arrives through a vector load: move the pointer to scalar registers
        const uint64_t ob = reinterpret_cast<uint64_t>(out_base);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)ob), hi = __builtin_amdgcn_readfirstlane((uint32_t)(ob >> 32));
        out_base = reinterpret_cast<out_t *>(((uint64_t)hi << 32) | lo);
    }
    float clip_peak = 0.0f;

    // ---- software pipeline: the input of frame i+1 is in flight while frame i is synthesised
    float2 xcur[8];
    int cpcur = 0, cntcur = 0;
    bool excur = false;
    FrameDesc fd_next = frame_at(fi0);
    prefetch(fd_next, 0, run.count > 0 && !(fd_next.flags & kFrameDrain), xcur, cpcur, cntcur, excur);
    uint2 stwcur = make_uint2(0u, 0u);
    if (kGroup) stwcur = steps_word(fd_next, run.count > 0 && !(fd_next.flags & kFrameDrain));
    // the first frame's input has to be there before the loop is entered: with loads pending at the loop header the
    // compiler's wait-count bookkeeping falls back to "wait for everything" at the first use inside the loop -- after
    // the next frame's loads have been issued, i.e. it would wait for those too
#pragma unroll
    for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(xcur[m].x), "v"(xcur[m].y));
    if (kHasFloor) asm volatile("" ::"v"(cpcur));
    // Floor1 curve of a frame's channel: table indices rendered into this wave's LDS row (free until the transform
    // needs it: in group mode the wave has just taken its spectrum out of it), the lane's 16 indices into fy.  Only the
    // bins below the spectrum's last non-zero one are rendered: real streams leave the top of the spectrum empty (the
    // residue ends below N/2; low-bitrate streams use a small part of it), and a zero times any table entry is zero.
    auto render_curve = [&](const FrameDesc &f, int cp, int m, const float2 (&x)[8], uint32_t (&fy)[4]) {
        const int lpb = lpb_of(f.flags);
        const int top = spectrum_top(x, lpb, lane);  // highest point k = k0 + lpb * m with a non-zero bin (wave-uniform)
        const int n = size_of(f.flags) >> 1;
        int n_render = 2 * (top + 1);
        if (VPZ_ABLATE(a) & 64) n_render = n;
        cp = lane < m ? cp : 0;  // lanes below the record's post count hold a post
        if (n_render > 0 &&
            !render_floor_indices_fast(reinterpret_cast<uint8_t *>(hcur), reinterpret_cast<int *>(hcur) + 256, n, n_render, cp, m, lane))
            render_floor_indices<32>(reinterpret_cast<uint8_t *>(hcur), reinterpret_cast<int *>(hcur) + 256, n, n_render, cp,
                                     m, lane);
        __builtin_amdgcn_wave_barrier();
        load_floor_indices(fy, reinterpret_cast<const uint8_t *>(hcur), lpb, lane);
        __builtin_amdgcn_wave_barrier();
    };
    // (kept across the loop where the instantiation has the registers -- planar output, 112 of 128; the interleaved-output ones are
    // at the limit and would spill: they compute the positions per frame, as every instantiation did before)
    constexpr bool kKeepStagePositions = kGroup && kOut == 0;
    int stage_pre[4] = {0, 0, 0, 0};
    if (kKeepStagePositions) {
        stage_positions(stage_pre, C, div_magic, ch, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(stage_pre[i]));  // (kept, not recomputed: four registers for 30 instructions a frame)
    }
    uint32_t fycur[4];
#ifdef VPZ_STAMPS
    unsigned long long t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
    int fi = fi0;
    for (int it = 0; it < iters; ++it) {
        const bool live = !kSync || fi < run.count;  // wave-uniform; idle iterations only keep the barriers matched
        const FrameDesc fd = fd_next;
        // blocks this pass covers: 1, or a batch of short ones (see the run builder)
        const int bsz = (kBatchShort && live) ? (int)((fd.flags >> kFrameBatchShift) & 7u) + 1 : 1;
        const int fin = fi + bsz;
        float2 xnext[8];
        int cpnext = 0, cntnext = 0;
        bool exnext = false;
        uint2 stwnext = make_uint2(0u, 0u);
        {
            const bool has_next = fin < run.count;  // (a wave that idles reads a stale descriptor; nothing of it is used)
            fd_next = frame_at(has_next ? fin : fi);
            prefetch(fd_next, fin - fi0, has_next && !(fd_next.flags & kFrameDrain) && !(VPZ_ABLATE(a) & 4), xnext, cpnext, cntnext,
                     exnext);
            if (kGroup) stwnext = steps_word(fd_next, has_next && !(fd_next.flags & kFrameDrain));
        }
        const bool drain = fd.flags & kFrameDrain;
        const bool batch = kBatchShort && bsz > 1;      // (wave-uniform) 2..8 short blocks in this pass
        const int nblk = size_of(fd.flags);
        const int nstage = batch ? 256 * bsz : nblk;    // what goes through the rows: the batch as one packet

        const bool is_long = nblk == 2048;  // "long" below means: the 2048-point transform
        const int n4 = kGeneral ? (nblk >> 2) : (is_long ? 512 : 64);
        const bool build = live && !drain;
        const bool exec = build && (excur || (kBatchShort && bsz > 1));  // (a batch settles silence block by block)
        VPZ_STAMP(0);  // descriptor + prefetch issue
        // ---- group mode: the packet goes through the group's LDS rows (de-interleave, inverse coupling)
        if (kGroup) {
            const bool stage = build;  // every packet goes through the rows (see load_group_share)
            // (ABI v4) a 2048 block whose residue ends in the lower half: only that half goes through the rows
            const bool stage_upper = kGeneral || batch || !is_long || upper_of(fd.flags);
            __syncthreads();  // every wave of the group is done with its row (previous block emitted)
            VPZ_STAMP(1);  // first barrier
            if (stage && !(VPZ_ABLATE(a) & 32)) {
                if ((fd.flags & kFrameInterleaved) && (VPZ_ABLATE(a) & 4096)) {
                    stage_by_lds_dma_gather((const float *)(k_spec + fd.spec_off), hcur, C, nstage >> 1, ch, lane);
                } else if ((fd.flags & kFrameInterleaved) && (VPZ_GROUP_DMA(a) || (VPZ_ABLATE(a) & 2048))) {
                    stage_by_lds_dma((const float *)(k_spec + fd.spec_off), s_work[gw0], C, nstage >> 1, ch, lane);
                } else if (fd.flags & kFrameInterleaved) {
                    if (!kKeepStagePositions) {
                        int lo = lane;
                        asm volatile("" : "+v"(lo));  // (per frame here: an opaque lane id keeps the compiler from hoisting them after all)
                        stage_positions(stage_pre, C, div_magic, ch, lo);
                    }
                    stage_interleaved(xcur, s_work[gw0], C, div_magic, nstage >> 1, ch, lane, stage_upper, stage_pre);
                } else {
                    stage_planar(xcur, hcur, nblk >> 1, lane, stage_upper);
                }
            }
            __syncthreads();
            VPZ_STAMP(2);  // staging + barrier
            // inverse coupling, steps in reverse order (Mapping.cs:166); the host has cut each mapping's steps into
            // LEVELS of steps that touch disjoint channels (bit 7 of a step's first byte: a new level starts here), so
            // that a workgroup barrier is needed between levels only -- (0,1),(2,3) of a 5.1 mapping run together
            const int n_steps = (int)((fd.flags >> kFrameStepsShift) & 0xFF);
            int sidx = n_steps - 1;
            const uint8_t *st = s_steps + 2 * ((fd.flags >> kFrameStepsOffShift) & kFrameStepsOffMask);
            const unsigned long long stw = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)stwcur.y) << 32) |
                                           (uint32_t)__builtin_amdgcn_readfirstlane((int)stwcur.x);
            // byte k of the mapping's steps: out of the prefetched word while the mapping has at most four
            auto step_byte = [&](int k) -> uint32_t { return n_steps <= 4 ? (uint32_t)(stw >> (8 * k)) & 0xFFu : (uint32_t)st[k]; };
            // (group_dma) the packet lies in the rows as it came: every wave takes its channel out of it, its own coupling step
            // applied on the way; the barrier(s) below then stand between the last pick-up and the first transform
            const bool in_place = stage && VPZ_GROUP_DMA(a) && (fd.flags & kFrameInterleaved);
            if (in_place && (exec || batch)) {  // [census: cold]
                int partner = -1;
                bool is_mag = false;
                for (int k = 0; k < n_steps; ++k) {
                    const int sm = (int)(step_byte(2 * k) & 0x7F), sa = (int)step_byte(2 * k + 1);
                    if (sm == ch) { partner = sa; is_mag = true; }
                    else if (sa == ch) { partner = sm; }
                }
                if (VPZ_ABLATE(a) & 16) partner = -1;
                int lb = lane;
                asm volatile("" : "+v"(lb));  // (frame-invariant LDS addresses: keep them inside the iteration)
                const int lpb = batch ? 8 : lpb_of(fd.flags);
                pickup_interleaved(xcur, s_work[gw0], C, ch, partner, is_mag, batch ? (lb >> 3) * 64 + (lb & 7) : (lb & (lpb - 1)), lpb);
            }
            const int n_levels = VPZ_GROUP_DMA(a) ? max(k_max_steps, 1) : k_max_steps;
            for (int lvl = 0; lvl < n_levels; ++lvl) {
                if (stage && !in_place && !(VPZ_ABLATE(a) & 16)) {
                    bool first = true;
                    while (sidx >= 0 && (first || !(step_byte(2 * sidx) & 0x80))) {
                        float4 *pm = reinterpret_cast<float4 *>(s_work[gw0 + (step_byte(2 * sidx) & 0x7F)]);
                        float4 *pa = reinterpret_cast<float4 *>(s_work[gw0 + step_byte(2 * sidx + 1)]);
                        for (int g = lane + 64 * ch; g < (stage_upper ? nstage >> 3 : nstage >> 4); g += 64 * C) {
                            float4 m4 = pm[g], a4 = pa[g];
                            couple(m4.x, a4.x);
                            couple(m4.y, a4.y);
                            couple(m4.z, a4.z);
                            couple(m4.w, a4.w);
                            pm[g] = m4;
                            pa[g] = a4;
                        }
                        --sidx;
                        first = false;
                    }
                }
                __syncthreads();
            }
            VPZ_STAMP(3);  // coupling levels + barriers
            if (in_place) {
                // (already in xcur)
            } else if (batch) {  // lane group g takes block g of the batch: points l + 8 m of its 64  [census: cold]
                const float2 *row2 = reinterpret_cast<const float2 *>(hcur);
                int lb = lane;
                asm volatile("" : "+v"(lb));
#pragma unroll
                for (int m = 0; m < 8; ++m) xcur[m] = row2[(lb >> 3) * 64 + (lb & 7) + 8 * m];
            } else if (stage && exec) {
                load_spectrum(xcur, hcur, lpb_of(fd.flags), lane, stage_upper);
            }
        }
        // ---- the curve, right before the row is needed for the transform
        bool batch_silent = false;  // this lane's block of a batch is a silent channel's
        if (kBatchShort && batch) {  // [census: cold]
            // the curves of the batch's blocks, one after the other, 128 bytes each (the row is free: every lane holds its
            // spectrum); their posts are asked for here, together -- the one exposed memory round trip of the pass (asked for at
            // the top of the pass they would be eight more live registers across the staging: the kernel spills)
            const int slot0 = fi - fi0;
            int cps[8], cns[8];
            int lb = lane;
            asm volatile("" : "+v"(lb));
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const int ff = f < bsz ? f : 0;
                cns[f] = f < bsz ? __builtin_amdgcn_readlane(cc_run, slot0 + ff) : 0;
                cps[f] = k_cposts[(size_t)(fd.rec + ff * C + ch) * 64 + lb];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                if (f < bsz) {
                    if (cns[f] == 0) {  // Mapping.cs:190-194: this block's channel is silent
                        if ((lb >> 3) == f) {
                            batch_silent = true;
#pragma unroll
                            for (int m = 0; m < 8; ++m) xcur[m] = make_float2(0.0f, 0.0f);
                        }
                    } else if (!(VPZ_ABLATE(a) & 8)) {
                        const int cp = lb < cns[f] ? cps[f] : 0;
                        uint8_t *dstc = reinterpret_cast<uint8_t *>(hcur) + 128 * f;
                        if (!render_floor_indices_fast(dstc, reinterpret_cast<int *>(hcur) + 256, 128, 128, cp, cns[f], lane))
                            render_floor_indices<32>(dstc, reinterpret_cast<int *>(hcur) + 256, 128, 128, cp, cns[f], lane);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            load_floor_indices(fycur, reinterpret_cast<const uint8_t *>(hcur) + 128 * (lb >> 3), 8, lane);
            __builtin_amdgcn_wave_barrier();
        } else if (kHasFloor && exec && !(fd.flags & kFrameNoFloor) && !(VPZ_ABLATE(a) & 8)) {
            render_curve(fd, cpcur, cntcur, xcur, fycur);
        }
        VPZ_STAMP(4);  // channel pick-up + curve
        if (build) {
            // The transforms address LDS by lane-derived indices that do not depend on the frame: computed ahead of the
            // frame loop they would all stay live across it (and spill in the variants that are short of registers);
            // an opaque lane id keeps them inside the iteration that uses them.
            int ln = lane;
            asm volatile("" : "+v"(ln));
            if (!exec) {  // [census: cold]
                // Mapping.cs:190-194: the channel is silent, its whole block is zero
                for (int i = lane; i < 2 * n4; i += 64) hcur[i] = 0.0f;
            } else if (VPZ_ABLATE(a) & 2) {  // [census: cold]
                float2 *h2 = reinterpret_cast<float2 *>(hcur);
#pragma unroll
                for (int m = 0; m < 8; ++m) h2[lane + 64 * m] = xcur[m];
            } else if (kGeneral) {  // [census: cold]
                // tables of this frame's size: the long set keeps the global layout, the short set is compacted
                const bool use_long = (fd.flags & kFrameLong) || k_size0 == k_size1;
                const float2 *tw = use_long ? s_twL : s_twS;
                const float2 *ab = use_long ? s_twL + kFastTwABOffset : s_twS + 256;
                const float2 *bc = use_long ? s_twL + kFastTwBCOffset : s_twS + 512;
                if (kHasFloor && !(fd.flags & kFrameNoFloor)) apply_floor(xcur, fycur, s_db, upper_of(fd.flags));
                float2 *h2 = reinterpret_cast<float2 *>(hcur);
                if (nblk == 2048) imdct2048_wave(xcur, h2, tw, ab, bc, ln);
                else if (nblk == 1024) imdct_mid_wave<4>(xcur, h2, tw, ab, bc, ln);
                else if (nblk == 512) imdct_mid_wave<2>(xcur, h2, tw, ab, bc, ln);
                else imdct256_wave8(xcur, h2, tw, bc, ln);
            } else if (is_long) {
                build_block<kHasFloor, true>(fd.flags, ln, xcur, fycur, hcur, s_twL, s_twAB, s_twBC, s_twS, s_db);
            } else {
                build_block<kHasFloor, false>(fd.flags, ln, xcur, fycur, hcur, s_twL, s_twAB, s_twBC, s_twS, s_db);
                if (kBatchShort && batch && batch_silent) {
                    // the reference does not transform a silent channel, it clears the block (Mapping.cs:190-194): all
                    // +0.0, where the transform of zeros leaves zeros of both signs
                    float2 *h2 = reinterpret_cast<float2 *>(hcur);
#pragma unroll
                    for (int m = 0; m < 8; ++m) h2[(ln >> 3) * 64 + (ln & 7) + 8 * m] = make_float2(0.0f, 0.0f);
                }
            }
        }

        VPZ_STAMP(5);  // floor multiply + transform
        // gfx950's vmcnt counts stores as well as loads, in issue order, and the number of stores below
        // is data dependent -- so the wait for the prefetched input is forced HERE, before this
        // frame's stores are issued; otherwise it would also wait for them (HBM write latency).
#pragma unroll
        for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(xnext[m].x), "v"(xnext[m].y));
        if (kHasFloor) asm volatile("" ::"v"(cpnext));
        VPZ_STAMP(6);  // wait for the next frame's input

        if (kPair) __syncthreads();  // both channels' blocks (and tails) are in LDS
        constexpr bool kCoop = kGroup && kOut != 0;  // interleaved output written by the group's waves together
        float o[4][4];      // the 16 samples of a long-after-long frame, or of a batch of short ones
        bool coop = false;  // ... wait in `o` for the cooperative store (uniform over the group's waves)
        if (kBatchShort && batch && !(VPZ_ABLATE(a) & 1)) {  // [census: cold]
            // ---- a batch of short blocks: 128 * bsz contiguous samples.  Sample i of block f is y_f[i] over the previous
            // block's y[128 + i] (StreamDecoder.cs:782-789 with both windows short): y_f[i] = -h_f[63 - i] (i < 64),
            // h_f[i - 64] otherwise; the partner is hp[i] (i < 64), hp[127 - i] otherwise, hp = the upper half of the
            // previous block's h -- the block before in the row, or the tail for block 0.
            const float4 *h4 = reinterpret_cast<const float4 *>(hcur);
            const float4 *t4 = reinterpret_cast<const float4 *>(tail);
            const float4 *s4 = reinterpret_cast<const float4 *>(s_slope0);
            out_t *dst = kInterleaved ? out_base + fd.out_off * k_channels + ch
                                       : out_base + (int64_t)ch * k_channel_stride + fd.out_off;
            const int64_t ostep = kInterleaved ? k_channels : 1;
            const bool aligned = kInterleaved ? (reinterpret_cast<uintptr_t>(out_base + fd.out_off * C) & 15) == 0
                                              : (reinterpret_cast<uintptr_t>(dst) & (kS16 ? 7 : 15)) == 0;
            coop = kCoop && aligned;
            int lf = lane;
            asm volatile("" : "+v"(lf));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gq = lf + 64 * r;            // float4 group of the batch's output
                const int f = gq >> 5, i4 = gq & 31;   // block, group inside the block
                const bool lower = i4 < 16;
                const bool valid = f < bsz;
                const int fs = valid ? f : 0;
                const float4 wl = s4[i4], wr = s4[31 - i4];
                const float4 hv = h4[fs * 32 + (lower ? 15 - i4 : i4 - 16)];
                const int pidx = lower ? i4 : 31 - i4;
                // (both in LDS; a selected float4 would spill.  After a long block the overlap sits at y[1472..1600) of its
                // output: floats 448..511 of the tail, straight and mirrored like a short block's)
                const float4 *pp = fs > 0 ? h4 + (fs - 1) * 32 + 16 : t4 + (prev_n4 == 512 ? 112 : 0);
                const float4 pv = pp[pidx];
                const float4 v = apply_y4(hv, lower, lower);
                const float4 t = apply_y4(pv, !lower, false);
                o[r][0] = ola(v.x, wl.x, t.x, wr.w);
                o[r][1] = ola(v.y, wl.y, t.y, wr.z);
                o[r][2] = ola(v.z, wl.z, t.z, wr.y);
                o[r][3] = ola(v.w, wl.w, t.w, wr.x);
                if (valid) {
                    if (k_clip) clip_group(o[r][0], o[r][1], o[r][2], o[r][3], clip_peak);
                    if (!coop) {
                        if (aligned || kInterleaved) {
                            // (planar and aligned: one 16-byte store; interleaved and not: four at the channel stride)
                            if (kInterleaved) {
                                out_t *d = dst + (int64_t)(4 * gq) * ostep;
                                if (kS16) {
                                    store_pcm(d, (out_t)to_s16(o[r][0]));
                                    store_pcm(d + ostep, (out_t)to_s16(o[r][1]));
                                    store_pcm(d + 2 * ostep, (out_t)to_s16(o[r][2]));
                                    store_pcm(d + 3 * ostep, (out_t)to_s16(o[r][3]));
                                } else {
                                    store_pcm(d, (out_t)o[r][0]);
                                    store_pcm(d + ostep, (out_t)o[r][1]);
                                    store_pcm(d + 2 * ostep, (out_t)o[r][2]);
                                    store_pcm(d + 3 * ostep, (out_t)o[r][3]);
                                }
                            } else if (kS16) {
                                store_nt(reinterpret_cast<uint2 *>(dst) + gq, pack_s16(o[r][0], o[r][1]), pack_s16(o[r][2], o[r][3]));
                            } else {
                                store_pcm4(reinterpret_cast<float4 *>(dst) + gq, make_float4(o[r][0], o[r][1], o[r][2], o[r][3]));
                            }
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                store_pcm(dst + 4 * gq + c, kS16 ? (out_t)to_s16(o[r][c]) : (out_t)o[r][c]);
                        }
                    }
                }
            }
        } else if (live && fi >= 0 && fd.out_count > 0 && !(VPZ_ABLATE(a) & 1)) {
            // ---- window + overlap-add + clip + store (StreamDecoder.cs:782-789, 573-591)
            // equal block sizes share one slope table (s_slope0 only holds a 128-entry short slope)
            const float *slope = ((fd.flags & kFrameSlope1) || k_size0 == k_size1) ? s_slope1 : s_slope0;
            const int plen = fd.packet_len;
            out_t *dst = kInterleaved ? out_base + fd.out_off * k_channels + ch
                                       : out_base + (int64_t)ch * k_channel_stride + fd.out_off;
            // every window boundary of the 256/2048 geometries is a multiple of 64 samples, so unless
            // an EOS trim cut the packet a float4 never straddles a mirror / overlap boundary
            // (interleaved output keeps the float4 arithmetic and scatters the four samples with the
            // channel stride; the other channels' waves fill the gaps of the same cache lines)
            const bool vec = !drain && ((fd.out_count | fd.left_start | plen | fd.prev_end) & 3) == 0 &&
                             (kInterleaved || (reinterpret_cast<uintptr_t>(dst) & (kS16 ? 7 : 15)) == 0);
            const int64_t ostep = kInterleaved ? k_channels : 1;
            auto store4 = [&](int g, float o0, float o1, float o2, float o3) {
                if (kInterleaved) {
                    out_t *d = dst + (int64_t)(4 * g) * ostep;
                    if (kS16) {
                        store_pcm(d, (out_t)to_s16(o0));
                        store_pcm(d + ostep, (out_t)to_s16(o1));
                        store_pcm(d + 2 * ostep, (out_t)to_s16(o2));
                        store_pcm(d + 3 * ostep, (out_t)to_s16(o3));
                    } else {
                        store_pcm(d, (out_t)o0);
                        store_pcm(d + ostep, (out_t)o1);
                        store_pcm(d + 2 * ostep, (out_t)o2);
                        store_pcm(d + 3 * ostep, (out_t)o3);
                    }
                } else if (kS16) {
                    store_nt(reinterpret_cast<uint2 *>(dst) + g, pack_s16(o0, o1), pack_s16(o2, o3));
                } else {
                    store_pcm4(reinterpret_cast<float4 *>(dst) + g, make_float4(o0, o1, o2, o3));
                }
            };
            // stereo pair: L / R blocks and tails of this stream sit in the two adjacent wave buffers
            const float4 *hL4 = reinterpret_cast<const float4 *>(s_work[wave & ~1]);
            const float4 *hR4 = reinterpret_cast<const float4 *>(s_work[wave | 1]);
            const float4 *tL4 = reinterpret_cast<const float4 *>(s_tail[wave & ~1]);
            const float4 *tR4 = reinterpret_cast<const float4 *>(s_tail[wave | 1]);
            out_t *pair_row = out_base + fd.out_off * 2;  // sample s of the packet at elements [2s, 2s+1]
            const bool vec_pair = kPair && !drain && ((fd.out_count | fd.left_start | plen | fd.prev_end) & 3) == 0 &&
                                  (reinterpret_cast<uintptr_t>(pair_row) & 15) == 0;
            auto store_pair = [&](int g, float l0, float l1, float l2, float l3, float r0, float r1, float r2, float r3) {
                if (k_clip) {
                    clip_group(l0, l1, l2, l3, clip_peak);
                    clip_group(r0, r1, r2, r3, clip_peak);
                }
                if (kS16) {
                    store_nt(reinterpret_cast<uint4 *>(pair_row) + g, pack_s16(l0, r0), pack_s16(l1, r1), pack_s16(l2, r2),
                             pack_s16(l3, r3));
                } else {
                    store_pcm4_pair(reinterpret_cast<float4 *>(pair_row) + 2 * g, make_float4(l0, r0, l1, r1), make_float4(l2, r2, l3, r3));
                }
            };
            if (vec_pair && VPZ_STEADY(fd) && prev_n4 == 512) {
                // long after long, long windows (see the single-channel version below): channel-0's wave writes
                // samples [0, 512) -- the negated mirror half --, channel-1's wave samples [512, 1024)
                const float4 *s4 = reinterpret_cast<const float4 *>(s_slope1);
#pragma unroll 1
                for (int rr = 0; rr < 2; ++rr) {
                    const int g = lane + 64 * (2 * ch + rr);
                    const float4 wl = s4[g], wr = s4[255 - g];
                    if (ch == 0) {
                        const float4 hl = hL4[127 - g], pl = tL4[g], hr = hR4[127 - g], pr = tR4[g];
                        store_pair(g, ola(-hl.w, wl.x, pl.x, wr.w), ola(-hl.z, wl.y, pl.y, wr.z),
                                   ola(-hl.y, wl.z, pl.z, wr.y), ola(-hl.x, wl.w, pl.w, wr.x),
                                   ola(-hr.w, wl.x, pr.x, wr.w), ola(-hr.z, wl.y, pr.y, wr.z),
                                   ola(-hr.y, wl.z, pr.z, wr.y), ola(-hr.x, wl.w, pr.w, wr.x));
                    } else {
                        const float4 hl = hL4[g - 128], pl = tL4[255 - g], hr = hR4[g - 128], pr = tR4[255 - g];
                        store_pair(g, ola(hl.x, wl.x, pl.w, wr.w), ola(hl.y, wl.y, pl.z, wr.z),
                                   ola(hl.z, wl.z, pl.y, wr.y), ola(hl.w, wl.w, pl.x, wr.x),
                                   ola(hr.x, wl.x, pr.w, wr.w), ola(hr.y, wl.y, pr.z, wr.z),
                                   ola(hr.z, wl.z, pr.y, wr.y), ola(hr.w, wl.w, pr.x, wr.x));
                    }
                }
            } else if (vec_pair) {  // [census: cold]
                // any other aligned geometry: the branch-free form below, for both channels, over this wave's
                // half of the float4 groups
                const float4 *s4 = reinterpret_cast<const float4 *>(slope);
                const int cnt4 = fd.out_count >> 2, mid = (cnt4 + 1) >> 1;
                const int g0 = ch ? mid : 0, g1 = ch ? cnt4 : mid;
                const int pn4 = prev_n4;
                for (int gb = g0; gb < g1; gb += 64) {
                    const int g = gb + lane;
                    const bool lv = g < g1;
                    const int i = (lv ? g : g1 - 1) << 2;
                    const Y4Map mc = map_y4(fd.left_start + i, n4);
                    const bool in = i < plen;
                    const int ii = in ? i : 0;
                    const int q = fd.prev_end + ii;
                    const bool pc = q >= 3 * pn4;
                    int pidx = (pc ? (4 * pn4 - 4 - q) : (q - 2 * pn4)) >> 2;
                    pidx = in ? pidx : 0;
                    const int ridx = in ? ((plen - 4 - ii) >> 2) : 0;
                    const float4 wl = s4[ii >> 2], wr = s4[ridx];
                    const float4 vl = apply_y4(hL4[mc.idx4], mc.rev, mc.neg), ql = apply_y4(tL4[pidx], pc, false);
                    const float4 vr = apply_y4(hR4[mc.idx4], mc.rev, mc.neg), qr = apply_y4(tR4[pidx], pc, false);
                    const float l0 = in ? ola(vl.x, wl.x, ql.x, wr.w) : vl.x, l1 = in ? ola(vl.y, wl.y, ql.y, wr.z) : vl.y;
                    const float l2 = in ? ola(vl.z, wl.z, ql.z, wr.y) : vl.z, l3 = in ? ola(vl.w, wl.w, ql.w, wr.x) : vl.w;
                    const float r0 = in ? ola(vr.x, wl.x, qr.x, wr.w) : vr.x, r1 = in ? ola(vr.y, wl.y, qr.y, wr.z) : vr.y;
                    const float r2 = in ? ola(vr.z, wl.z, qr.z, wr.y) : vr.z, r3 = in ? ola(vr.w, wl.w, qr.w, wr.x) : vr.w;
                    if (lv) store_pair(g, l0, l1, l2, l3, r0, r1, r2, r3);
                }
            } else if (!kPair && vec && VPZ_STEADY(fd) && prev_n4 == 512) {
                // long block after a long block with long windows on both sides (the steady state of
                // every stream): the geometry is a compile-time constant -- first half of the output
                // is the negated mirror of h[0:512) over the straight previous tail, second half is
                // h[0:512) straight over the mirrored tail; both window halves come from one table.
                const float4 *h4 = reinterpret_cast<const float4 *>(hcur);
                const float4 *t4 = reinterpret_cast<const float4 *>(tail);
                const float4 *s4 = reinterpret_cast<const float4 *>(s_slope1);
                coop = kCoop && C > 2 && (reinterpret_cast<uintptr_t>(out_base + fd.out_off * C) & 15) == 0;
                int lf = lane;
                asm volatile("" : "+v"(lf));  // (no store address of this path may be computed ahead of the frame loop)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int g = lf + 64 * r;
                    const float4 wl = s4[g], wr = s4[255 - g];
                    if (r < 2) {
                        const float4 hv = h4[127 - g], pv = t4[g];
                        o[r][0] = ola(-hv.w, wl.x, pv.x, wr.w);
                        o[r][1] = ola(-hv.z, wl.y, pv.y, wr.z);
                        o[r][2] = ola(-hv.y, wl.z, pv.z, wr.y);
                        o[r][3] = ola(-hv.x, wl.w, pv.w, wr.x);
                    } else {
                        const float4 hv = h4[g - 128], pv = t4[255 - g];
                        o[r][0] = ola(hv.x, wl.x, pv.w, wr.w);
                        o[r][1] = ola(hv.y, wl.y, pv.z, wr.z);
                        o[r][2] = ola(hv.z, wl.z, pv.y, wr.y);
                        o[r][3] = ola(hv.w, wl.w, pv.x, wr.x);
                    }
                    if (kInterleaved && !coop) {  // scattered stores: finish each group of four at once (short live ranges)
                        if (k_clip) clip_group(o[r][0], o[r][1], o[r][2], o[r][3], clip_peak);
                        store4(g, o[r][0], o[r][1], o[r][2], o[r][3]);
                    }
                }
                if ((!kInterleaved || coop) && k_clip) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) clip_group(o[r][0], o[r][1], o[r][2], o[r][3], clip_peak);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (!kInterleaved) store4(lf + 64 * r, o[r][0], o[r][1], o[r][2], o[r][3]);
            } else if (!kPair && vec) {  // [census: cold]
                // branch-free: lanes past the end clamp their reads and skip only the store; samples
                // past the overlap take weights (1, 0)
                const float4 *h4 = reinterpret_cast<const float4 *>(hcur);
                const float4 *t4 = reinterpret_cast<const float4 *>(tail);
                const float4 *s4 = reinterpret_cast<const float4 *>(slope);
                const int cnt4 = fd.out_count >> 2;
                const int nr = (cnt4 + 63) >> 6;
                const int pn4 = prev_n4;
                int lv = lane;
                asm volatile("" : "+v"(lv));  // (see stage_interleaved: no address of this path may outlive a frame)
                for (int r = 0; r < nr; ++r) {
                    const int g = lv + 64 * r;
                    const bool live = g < cnt4;
                    const int i = (live ? g : cnt4 - 1) << 2;
                    const Y4Map mc = map_y4(fd.left_start + i, n4);
                    const bool in = i < plen;
                    const int ii = in ? i : 0;
                    const int q = fd.prev_end + ii;          // in [2*pn4, 4*pn4) whenever `in`
                    const bool pc = q >= 3 * pn4;
                    int pidx = (pc ? (4 * pn4 - 4 - q) : (q - 2 * pn4)) >> 2;
                    pidx = in ? pidx : 0;
                    const int ridx = in ? ((plen - 4 - ii) >> 2) : 0;
                    const float4 hv = h4[mc.idx4];
                    const float4 pv = t4[pidx];
                    const float4 wl = s4[ii >> 2];
                    const float4 wr = s4[ridx];
                    const float4 v = apply_y4(hv, mc.rev, mc.neg);
                    const float4 t = apply_y4(pv, pc, false);
                    // (v * v_lhs) + (v_prev * v_rhs), StreamDecoder.cs:788; wr is read reversed.
                    // Scalars only from here on: a float4 that is selected / passed by reference ends
                    // up in scratch memory.
                    float o0 = ola(v.x, wl.x, t.x, wr.w);
                    float o1 = ola(v.y, wl.y, t.y, wr.z);
                    float o2 = ola(v.z, wl.z, t.z, wr.y);
                    float o3 = ola(v.w, wl.w, t.w, wr.x);
                    o0 = in ? o0 : v.x;
                    o1 = in ? o1 : v.y;
                    o2 = in ? o2 : v.z;
                    o3 = in ? o3 : v.w;
                    if (k_clip) {
                        // (a lane past the end has re-computed the last group: real samples, counted twice at worst)
                        clip_group(o0, o1, o2, o3, clip_peak);
                    }
                    if (live) store4(g, o0, o1, o2, o3);
                }
            } else {
                int l0 = lane;
                asm volatile("" : "+v"(l0));  // (rare path: its addresses must not be computed ahead of the frame loop)
                for (int i = l0; i < fd.out_count; i += 64) {  // [census: cold]
                    float v;
                    if (drain) {
                        v = tail_at(tail, fd.prev_end + i, prev_n4);
                    } else {
                        v = y_from_h(hcur, fd.left_start + i, n4);
                        if (i < plen) {
                            const float t = tail_at(tail, fd.prev_end + i, prev_n4);
                            v = ola(v, slope[i], t, slope[plen - 1 - i]);
                        }
                    }
                    if (k_clip) {
                        v = clip_track(v, clip_peak);
                    }
                    store_pcm(dst + i * ostep, kS16 ? (out_t)to_s16(v) : (out_t)v);
                }
            }
        }
        VPZ_STAMP(7);  // window + overlap-add + stores
        if (kPair) __syncthreads();  // the partner is done reading this wave's block and tail
        if (live && !drain) {
            // keep what a later block can overlap with: y[N/2 .. N) lives in the upper half of h
            if (kGeneral && !is_long) {  // [census: cold]
                for (int i = lane; i < n4; i += 64) tail[i] = hcur[n4 + i];
            } else if (is_long) {
                const float4 *src = reinterpret_cast<const float4 *>(hcur + 512);
                float4 *dt = reinterpret_cast<float4 *>(tail);
                const float4 t0 = src[lane], t1 = src[lane + 64];
                dt[lane] = t0;
                dt[lane + 64] = t1;
            } else {
                tail[lane] = hcur[(batch ? 128 * (bsz - 1) : 0) + 64 + lane];  // (a batch: its last block)
            }
            prev_n4 = n4;
        }
        if (kCoop) {
            // the block is spent (its lower half went into `o`, its upper half into the tail): the row takes the frame's
            // output, and once every channel's is there the group's waves write the packet out together
            if (coop) {
                __builtin_amdgcn_wave_barrier();
                int lf = lane;
                asm volatile("" : "+v"(lf));
                float4 *row4 = reinterpret_cast<float4 *>(hcur);
#pragma unroll
                for (int r = 0; r < 4; ++r) row4[lf + 64 * r] = make_float4(o[r][0], o[r][1], o[r][2], o[r][3]);
            }
            __syncthreads();
            if (coop) emit_interleaved_rows<kS16>(s_work[gw0], out_base + fd.out_off * C, C, div_magic, ch, lane, batch ? 128 * bsz : 1024);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) xcur[m] = xnext[m];
        cpcur = cpnext;
        cntcur = cntnext;
        stwcur = stwnext;
        excur = exnext;
        fi = fin;
        VPZ_STAMP(8);  // tail
    }
#ifdef VPZ_STAMPS
    if (a.stamps && active && lane == 0) {
        for (int k = 0; k < 9; ++k) atomicAdd(&a.stamps[k], t_acc[k]);
        atomicAdd(&a.stamps[15], 1ull);
    }
#endif

    // ---- keep the last block's tail for the next batch (the reference keeps _prevPacketBuf)
    if ((run.flags & kRunSaveState) && prev_n4 > 0) {
        float *st = a.state_h + (size_t)(run.state_slot ^ 1) * a.state_slot_floats + ((size_t)run.stream * k_channels + ch) * half1;
        for (int i = lane; i < prev_n4; i += 64) st[i] = tail[i];
    }
    // HasClipped is sticky until ResetDecoder: the flag holds the stream's reset epoch, so a reset costs no device work
    if (k_clip && __any(clip_peak > 0.99999994f) && lane == 0) atomicMax(&a.clipped[run.stream], run.clip_epoch);
}

#undef k_size0
#undef k_size1
#undef k_clip
#undef k_max_steps
#undef k_channels
#undef k_no_ccount
#undef k_spec
#undef k_inv_db
#undef k_cposts
#undef k_channel_stride

// ---------------------------------------------------------------------------------------------
// Any-block-size path (64 .. 8192): three plain passes over HBM instead of the fused kernel --
//   generic_floor_kernel : Floor1 curve x spectrum in place (or zeros for a silent channel)
//   imdct_exact_kernel   : the reference's own IMDCT schedule (imdct_exact.hip), gathered by offset
//   generic_ola_kernel   : window + overlap-add + clip + store from the full IMDCT outputs
// It exists so that every Vorbis block-size pair decodes; sizes out of {256, 512, 1024, 2048} take the fused kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void generic_floor_kernel(const GenericFrame *__restrict__ frames, int channels,
                                                           int half1, float *__restrict__ spec,
                                                           const uint8_t *__restrict__ ccount,
                                                           const uint8_t *__restrict__ curve_y,
                                                           const float *__restrict__ inv_db)
{
    const GenericFrame fr = frames[blockIdx.x / channels];
    const int ch = blockIdx.x % channels;
    if (fr.flags & (kFrameDrain | kFrameNoFloor)) return;
    const int half = fr.n >> 1;
    float *x = spec + fr.spec_off + (int64_t)ch * half;
    if (ccount[fr.rec + ch] == 0) {  // Mapping.cs:190-194
        for (int i = threadIdx.x; i < half; i += 256) x[i] = 0.0f;
        return;
    }
    const uint8_t *row = curve_y + (size_t)(fr.rec + ch) * half1;
    for (int i = threadIdx.x; i < half; i += 256) x[i] *= inv_db[row[i]];
}

// y[pos] of a block whose full IMDCT output sits in memory
__global__ __launch_bounds__(256) void generic_ola_kernel(const GenericFrame *__restrict__ frames, int channels,
                                                         int size0, int size1, const float *__restrict__ ybuf,
                                                         float *__restrict__ state_y,
                                                         const float *__restrict__ slope0,
                                                         const float *__restrict__ slope1, float *__restrict__ out,
                                                         const int64_t *__restrict__ stream_out_off,
                                                         int64_t channel_stride, int interleaved, int clip,
                                                         int32_t *__restrict__ clipped, int s16)
{
    const GenericFrame fr = frames[blockIdx.x / channels];
    const int ch = blockIdx.x % channels;
    const int half1 = size1 >> 1;
    const bool drain = fr.flags & kFrameDrain;
    const float *ycur = drain ? nullptr : ybuf + fr.y_off + (int64_t)ch * fr.n;
    // previous block: its full output in ybuf, or the saved upper half (positions prev_n/2 .. prev_n)
    const float *yprev = nullptr;
    int prev_base = 0;
    if (fr.prev_y_off >= 0) {
        yprev = ybuf + fr.prev_y_off + (int64_t)ch * fr.prev_n;
    } else if (fr.prev_y_off == -1) {
        yprev = state_y + ((int64_t)fr.stream * channels + ch) * half1;
        prev_base = fr.prev_n >> 1;
    }
    const float *slope = (fr.flags & kFrameSlope1) ? slope1 : slope0;
    // element index of the first sample (float32 or, for the s16 layouts, int16 elements)
    const int64_t first = (stream_out_off ? stream_out_off[fr.stream] : 0) +
                          (interleaved ? fr.out_off * channels + ch : (int64_t)ch * channel_stride + fr.out_off);
    float *dst = out + first;
    int16_t *dst16 = reinterpret_cast<int16_t *>(out) + first;
    const int64_t step = interleaved ? channels : 1;
    bool clipped_any = false;
    for (int i = threadIdx.x; i < fr.out_count; i += 256) {
        float v;
        if (drain) {
            v = yprev[fr.prev_end + i - prev_base];
        } else {
            v = ycur[fr.left_start + i];
            if (i < fr.packet_len) {
                const float t = yprev[fr.prev_end + i - prev_base];
                v = ola(v, slope[i], t, slope[fr.packet_len - 1 - i]);
            }
        }
        if (clip) {
            clipped_any |= was_clipped(v);
            v = clip_value(v);
        }
        if (s16) dst16[i * step] = (int16_t)to_s16(v);
        else dst[i * step] = v;
    }
    if (clip && clipped_any) atomicOr(&clipped[fr.stream], 1);
}

// Runs after generic_ola_kernel (same stream): the last block of each stream keeps y[n/2 .. n) for
// the next call.  A separate launch because the block that reads a stream's old state and the block
// that would overwrite it are different workgroups of the OLA launch.
__global__ __launch_bounds__(256) void generic_save_state_kernel(const GenericFrame *__restrict__ frames,
                                                                const int32_t *__restrict__ save_list, int channels,
                                                                int size1, const float *__restrict__ ybuf,
                                                                float *__restrict__ state_y)
{
    const GenericFrame fr = frames[save_list[blockIdx.x / channels]];
    const int ch = blockIdx.x % channels;
    const int h = fr.n >> 1;
    const float *ycur = ybuf + fr.y_off + (int64_t)ch * fr.n;
    float *st = state_y + ((int64_t)fr.stream * channels + ch) * (size1 >> 1);
    for (int i = threadIdx.x; i < h; i += 256) st[i] = ycur[h + i];
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_floor1_unwrap(int n_rec, const int16_t *posts, const uint8_t *post_counts, const uint8_t *rec_info,
                                const FloorDev *floors, int n_floors, int32_t *cposts, uint8_t *ccount, int16_t *dbg_y,
                                uint8_t *dbg_f, hipStream_t stream, int f0_fused)
{
    if (n_rec <= 0) return hipSuccess;
    unsigned long long *stamps = nullptr;
#ifdef VPZ_STAMPS
    static unsigned long long *d_stamps = nullptr;
    if (!d_stamps) (void)hipMalloc(&d_stamps, 16 * sizeof(unsigned long long));
    (void)hipMemsetAsync(d_stamps, 0, 16 * sizeof(unsigned long long), stream);
    stamps = d_stamps;
#endif
    if (n_floors <= kPrepFloorsInLds)
        hipLaunchKernelGGL(floor1_unwrap_kernel<true>, dim3((n_rec + kUnwrapRecs * kUnwrapWaves - 1) / (kUnwrapRecs * kUnwrapWaves)), dim3(64 * kUnwrapWaves), 0, stream, n_rec, posts,
                           post_counts, rec_info, floors, n_floors, cposts, ccount, dbg_y, dbg_f, stamps, f0_fused);
    else
        hipLaunchKernelGGL(floor1_unwrap_kernel<false>, dim3((n_rec + kUnwrapRecs * kUnwrapWaves - 1) / (kUnwrapRecs * kUnwrapWaves)), dim3(64 * kUnwrapWaves), 0, stream, n_rec, posts,
                           post_counts, rec_info, floors, n_floors, cposts, ccount, dbg_y, dbg_f, stamps, f0_fused);
#ifdef VPZ_STAMPS
    {
        unsigned long long h[16];
        (void)hipMemcpyAsync(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost, stream);
        (void)hipStreamSynchronize(stream);
        fprintf(stderr, "[unwrap stamps] %llu waves, cycles per wave: staging %.0f, walk %.0f, selection %.0f, copy-out %.0f\n", h[15],
                h[15] ? (double)h[0] / h[15] : 0.0, h[15] ? (double)h[1] / h[15] : 0.0, h[15] ? (double)h[2] / h[15] : 0.0,
                h[15] ? (double)h[3] / h[15] : 0.0);
    }
#endif
    return hipGetLastError();
}

hipError_t launch_floor1_render(int n_rec, const int32_t *cposts, const uint8_t *ccount, const uint8_t *rec_info,
                                int half0, int half1, uint8_t *curve_y, hipStream_t stream)
{
    if (n_rec <= 0) return hipSuccess;
    hipLaunchKernelGGL(floor1_render_kernel, dim3((n_rec + kRenderWaves - 1) / kRenderWaves), dim3(64 * kRenderWaves), 0,
                       stream, n_rec, cposts, ccount, rec_info, half0, half1, curve_y);
    return hipGetLastError();
}

hipError_t launch_coupling(const void *pkts, int n_pkts, const uint8_t *steps, int channels,
                           const float *residue, float *temp, int max_half, hipStream_t stream)
{
    if (n_pkts <= 0) return hipSuccess;
    // grid.y is limited to 65535: chunk the packet list
    const CouplingPacket *p = static_cast<const CouplingPacket *>(pkts);
    for (int done = 0; done < n_pkts; done += 65535) {
        int cnt = n_pkts - done < 65535 ? n_pkts - done : 65535;
        if (channels <= kCouplingTileChannels)
            hipLaunchKernelGGL(coupling_tile_kernel,
                               dim3((max_half + coupling_tile_width(channels) - 1) / coupling_tile_width(channels), cnt),
                               dim3(256), sizeof(float) * (size_t)(coupling_tile_width(channels) + 1) * (size_t)channels,
                               stream, p + done, steps, channels, residue, temp);
        else
            hipLaunchKernelGGL(coupling_kernel, dim3((max_half + 255) / 256, cnt), dim3(256), 0, stream,
                               p + done, steps, channels, residue, temp, max_half);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// block-size pairs the fused kernel takes at all, and those among them that need its general variant
bool synth_supports_sizes(int size0, int size1)
{
    auto ok = [](int n) { return n == 256 || n == 512 || n == 1024 || n == 2048; };
    return ok(size0) && ok(size1);
}
bool synth_needs_general(int size0, int size1)
{
    auto plain = [](int n) { return n == 256 || n == 2048; };
    return !(plain(size0) && plain(size1));
}

// group mode needs every channel of a run inside one workgroup
bool synth_group_supported(int channels) { return channels >= 2 && channels <= kGroupMaxChannels; }

hipError_t launch_synth(const SynthArgs &args, bool has_floor, hipStream_t stream)
{
    const long items = (long)args.n_runs * args.channels;
    if (items <= 0) return hipSuccess;
    const bool group = args.group != 0;
    const int grid = group ? (int)((args.n_runs + kSynthWaves / args.channels - 1) / (kSynthWaves / args.channels))
                           : (int)((items + kSynthWaves - 1) / kSynthWaves);
    // the interleaved store patterns cost registers the planar steady state needs: one instantiation each
    const int out_kind = !args.interleaved ? 0 : (args.channels == 2 ? 2 : 1);
    static const int extra_lds = getenv("VPZ_SYNTH_EXTRA_LDS") ? atoi(getenv("VPZ_SYNTH_EXTRA_LDS")) : 0;  // occupancy experiments
#define VPZ_LAUNCH_SYNTH(F, O, G, R)                                                                                  \
    do {                                                                                                              \
        if (args.s16)                                                                                                 \
            hipLaunchKernelGGL((synth_kernel<F, O, G, R, true>), dim3(grid), dim3(kSynthThreads), extra_lds, stream, args);  \
        else                                                                                                          \
            hipLaunchKernelGGL((synth_kernel<F, O, G, R, false>), dim3(grid), dim3(kSynthThreads), extra_lds, stream, args); \
    } while (0)
#define VPZ_LAUNCH_SYNTH_OUT(F, G, R)                      \
    do {                                                   \
        if (out_kind == 0) VPZ_LAUNCH_SYNTH(F, 0, G, R);   \
        else if (out_kind == 1) VPZ_LAUNCH_SYNTH(F, 1, G, R); \
        else VPZ_LAUNCH_SYNTH(F, 2, G, R);                 \
    } while (0)
#define VPZ_LAUNCH_SYNTH_GEN(F, R)                                       \
    do {                                                                 \
        if (synth_needs_general(args.size0, args.size1)) VPZ_LAUNCH_SYNTH_OUT(F, true, R); \
        else VPZ_LAUNCH_SYNTH_OUT(F, false, R);                          \
    } while (0)
    if (has_floor) {
        if (group) VPZ_LAUNCH_SYNTH_GEN(true, true);
        else VPZ_LAUNCH_SYNTH_GEN(true, false);
    } else {
        if (group) VPZ_LAUNCH_SYNTH_GEN(false, true);
        else VPZ_LAUNCH_SYNTH_GEN(false, false);
    }
#undef VPZ_LAUNCH_SYNTH_GEN
#undef VPZ_LAUNCH_SYNTH_OUT
#undef VPZ_LAUNCH_SYNTH
    return hipGetLastError();
}

// wavefronts of synth_kernel the chip keeps resident and busy (for sizing runs so that the grid fills an
// integral number of rounds); in group mode only floor(8 / channels) * channels waves of a workgroup own work
int synth_resident_waves(bool has_floor, int num_cu, int channels, bool group)
{
    int per_cu = 0;
    hipError_t e = has_floor
                       ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, synth_kernel<true, 0, false, false, false>, kSynthThreads, 0)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, synth_kernel<false, 0, false, false, false>, kSynthThreads, 0);
    if (e != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 2;
    }
    const int busy = group ? (kSynthWaves / channels) * channels : kSynthWaves;
    return num_cu * per_cu * busy;
}

hipError_t launch_generic_floor(const GenericFrame *frames, int n_frames, int channels, int half1, float *spec,
                                const uint8_t *ccount, const uint8_t *curve_y, const float *inv_db,
                                hipStream_t stream)
{
    if (n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(generic_floor_kernel, dim3(n_frames * channels), dim3(256), 0, stream, frames, channels, half1,
                       spec, ccount, curve_y, inv_db);
    return hipGetLastError();
}

hipError_t launch_generic_ola(const GenericFrame *frames, int n_frames, int channels, int size0, int size1,
                              const float *ybuf, float *state_y, const float *slope0, const float *slope1, float *out,
                              const int64_t *stream_out_off, int64_t channel_stride, int interleaved, int clip,
                              int32_t *clipped, int s16, hipStream_t stream)
{
    if (n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(generic_ola_kernel, dim3(n_frames * channels), dim3(256), 0, stream, frames, channels, size0,
                       size1, ybuf, state_y, slope0, slope1, out, stream_out_off, channel_stride, interleaved, clip,
                       clipped, s16);
    return hipGetLastError();
}

hipError_t launch_generic_save_state(const GenericFrame *frames, const int32_t *save_list, int n_save, int channels,
                                     int size1, const float *ybuf, float *state_y, hipStream_t stream)
{
    if (n_save <= 0) return hipSuccess;
    hipLaunchKernelGGL(generic_save_state_kernel, dim3(n_save * channels), dim3(256), 0, stream, frames, save_list,
                       channels, size1, ybuf, state_y);
    return hipGetLastError();
}

size_t coupling_packet_size() { return sizeof(CouplingPacket); }
void fill_coupling_packet(void *dst, int64_t src_off, int64_t dst_off, int32_t half, int32_t steps_off,
                          int32_t steps, int32_t interleaved)
{
    CouplingPacket *p = static_cast<CouplingPacket *>(dst);
    p->src_off = src_off;
    p->dst_off = dst_off;
    p->half = half;
    p->steps_off = steps_off;
    p->steps = steps;
    p->interleaved = interleaved;
}

}  // namespace vpz
