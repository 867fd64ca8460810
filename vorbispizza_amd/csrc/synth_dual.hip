// synth_dual_kernel -- the stereo fast path of vpz_decoder_synth: ONE wavefront synthesises BOTH channels of a run of
// consecutive blocks of one stream.
//
//   Residue2.cs:42-51   the interleaved vector [bin][2] never needs a de-interleave pass: the float4 at index k IS
//                       (L[2k], R[2k], L[2k+1], R[2k+1]), i.e. point k of both channels' transforms for the lane that
//                       loads it -- 16-byte loads, every wave-load one contiguous 1 KiB span;
//   Mapping.cs:166-195  inverse coupling in registers (couple(), element-wise), steps in reverse order; then per channel
//                       ExecuteChannel ? floor curve x spectrum -> inverse MDCT : a block of +0.0;
//   Floor1.cs:222-397   both channels' curves rendered side by side into the two LDS rows the transforms then use;
//   Mdct.cs             two transforms per wave, step by step side by side (imdct2048_wave_x2): twiddles read once for
//                       both, each one's LDS round trips covered by the other's arithmetic;
//   StreamDecoder.cs:764-791, 515-638   window + overlap-add + clip, and the store: interleaved output leaves as dense
//                       (L R L R) 16-byte pieces written by the one wave that holds both channels.
//
// No workgroup barrier after the tables are staged, no lock step between waves, no LDS staging of the packet: what group
// mode of synth_kernel spends on them (16 % on the de-interleave's LDS stores, 4 % on barriers, the
// coupling's LDS round trip) is gone.  A workgroup is 4 waves x 2 channels: the same LDS per channel as synth_kernel's,
// 2 workgroups = 8 waves per CU, up to 256 VGPRs per lane.
// Up to eight consecutive SHORT blocks go through one pass (lane group g takes block g, both channels), for floored and
// for already-floored (VPZ_PKT_NO_FLOOR) packets, interleaved or planar input.
// Arithmetic per channel is operation for operation that of synth_kernel: the two give the same bits.
#include <cstdlib>
#include <type_traits>

#include "imdct_core.hpp"
#include "synth_common.hpp"
#include "synth_desc.hpp"
#include "vpz_internal.hpp"

// The same kernel for CHANNEL PAIRS of streams with more than two channels (synth_pairs.hip compiles this file with
// VPZ_DUAL_PAIRS=1): a stream whose coupling steps, over all its mappings, join its channels two by two (5.1 as libvorbis
// writes it: L-R, Ls-Rs; C and LFE on their own) is, to the arithmetic, a set of stereo streams -- no value of one pair ever
// meets a value of another (Mapping.cs:166-195).  A workgroup then walks four runs of ONE pair (SynthArgs.pair_ch: its two
// channels), the pairs of a chunk of runs are neighbours in the grid and on one XCD (an interleaved packet's lines are fetched
// by all of them: from HBM once), and what changes against the stereo kernel is addressing only -- rows of a [C][bin] packet or
// columns of a [bin][C] one, the records and the saved state of channels chA / chB, PCM rows or columns of those two.
#ifndef VPZ_DUAL_PAIRS
#define VPZ_DUAL_PAIRS 0
#endif
#if VPZ_DUAL_PAIRS
#define synth_dual_kernel synth_pairs_kernel
#endif

namespace vpz {

constexpr bool kPairs = VPZ_DUAL_PAIRS != 0;
constexpr int kDualWaves = VPZ_DUAL_WAVES;
#ifndef VPZ_DUAL_WPS
#define VPZ_DUAL_WPS (VPZ_DUAL_WAVES >= 10 ? 3 : 2)
#endif
constexpr int kDualWavesPerSimd = VPZ_DUAL_WPS;  // 4-wave workgroups: two per CU (LDS); 10-wave ones: one
constexpr int kDualThreads = 64 * kDualWaves;

// Both channels' Floor1 curves at once (render_floor_indices_fast, phase by phase for the two records): the phases are
// chains of dependent LDS round trips, side by side they overlap.  c = 0 / 1: out[c] (one byte per bin), aux[c],
// n_render[c] (0: nothing to render for this channel), cp[c] / m[c] (the record's active posts, this lane's and their
// count).  Returns false -- nothing usable written -- if a record breaks the closed form's bound (corrupt streams only).
__device__ __forceinline__ bool render_floor_indices_fast_x2(uint8_t *out0, uint8_t *out1, int *aux0, int *aux1, int n,
                                                             int n_render0, int n_render1, int cp0, int cp1, int m0, int m1,
                                                             int lane)
{
    uint8_t *out[2] = {out0, out1};
    int *aux[2] = {aux0, aux1};
    const int n_render[2] = {n_render0, n_render1}, cp[2] = {cp0, cp1}, mm[2] = {m0, m1};
    int x0[2], x1[2];
    float x0f[2], y0f[2], dyf[2], rinv[2];
    bool mine[2];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int p1 = __shfl_down(cp[c], 1);
        x0[c] = cp[c] & 0xFFFF;
        const int y0 = cp[c] >> 16;
        int x1raw = n, y1 = y0;  // flat tail after the last active post (Floor1.cs:259-262)
        if (lane + 1 < mm[c]) { x1raw = p1 & 0xFFFF; y1 = p1 >> 16; }
        x1[c] = x1raw < n ? x1raw : n;  // Math.Min(hx, n) enters the slope: quirk q2 (Floor1.cs:248)
        const int adx = x1[c] - x0[c];
        const int dy = y1 - y0;
        mine[c] = n_render[c] > 0 && lane < mm[c] && adx > 0;
        bad |= mine[c] && iabs(dy) * adx > (1 << 21);
        x0f[c] = (float)x0[c];
        y0f[c] = (float)y0;
        dyf[c] = (float)dy;
        rinv[c] = __builtin_amdgcn_rcpf((float)(adx > 0 ? adx : 1));
    }
    if (__any(bad)) return false;
#pragma unroll
    for (int c = 0; c < 2; ++c)
        if (lane < 32) aux[c][lane] = 0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 2; ++c)
        if (mine[c]) {
            atomicOr(reinterpret_cast<unsigned int *>(&aux[c][x0[c] >> 5]), 1u << (x0[c] & 31));
            reinterpret_cast<float4 *>(aux[c] + 64)[lane] = make_float4(x0f[c], y0f[c], dyf[c], rinv[c]);
        }
    __builtin_amdgcn_wave_barrier();
    {
        const int c0 = (lane < 32) ? __popc((unsigned)aux[0][lane]) : 0;
        const int c1 = (lane < 32) ? __popc((unsigned)aux[1][lane]) : 0;
        const int i0 = wave_scan32(c0), i1 = wave_scan32(c1);
        if (lane < 32) {
            aux[0][32 + lane] = i0 - c0;
            aux[1][32 + lane] = i1 - c1;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int rounds0 = (n_render[0] + 255) >> 8, rounds1 = (n_render[1] + 255) >> 8;
    const int rounds = rounds0 > rounds1 ? rounds0 : rounds1;
    for (int r = 0; r < rounds; ++r) {
        const int w = lane + 64 * r, x = 4 * w;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (x < n_render[c]) {
                const int *bitmap = aux[c], *prefix = aux[c] + 32;
                const float4 *seg = reinterpret_cast<const float4 *>(aux[c] + 64);
                const unsigned bw = (unsigned)bitmap[x >> 5];
                const int j = prefix[x >> 5] + __popc(bw & ((2u << (x & 31)) - 1u)) - 1;
                const float4 sg = seg[j];
                float t = fmaf((float)x - sg.x, sg.z, copysignf(0.5f, sg.z));
                uint32_t pk = 0;
                pk = __builtin_amdgcn_cvt_pk_u8_f32(sg.y + truncf(t * sg.w), 0, pk);
                t += sg.z;
                pk = __builtin_amdgcn_cvt_pk_u8_f32(sg.y + truncf(t * sg.w), 1, pk);
                t += sg.z;
                pk = __builtin_amdgcn_cvt_pk_u8_f32(sg.y + truncf(t * sg.w), 2, pk);
                t += sg.z;
                pk = __builtin_amdgcn_cvt_pk_u8_f32(sg.y + truncf(t * sg.w), 3, pk);
                reinterpret_cast<uint32_t *>(out[c])[w] = pk;
            }
        }
    }
    // a post inside a word: its bins up to the end of that word (or of its segment) belong to its own line
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int xe = x1[c] < n_render[c] ? x1[c] : n_render[c];
        const int al = x0[c] & 3;
        if (mine[c] && al != 0 && x0[c] < xe) {
            const int cnt = min(4 - al, xe - x0[c]);
            float t = copysignf(0.5f, dyf[c]);
            out[c][x0[c]] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(y0f[c] + truncf(t * rinv[c]), 0, 0u);
            t += dyf[c];
            if (cnt > 1) out[c][x0[c] + 1] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(y0f[c] + truncf(t * rinv[c]), 0, 0u);
            t += dyf[c];
            if (cnt > 2) out[c][x0[c] + 2] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(y0f[c] + truncf(t * rinv[c]), 0, 0u);
        }
    }
    return true;
}

// Floor0.Apply (Floor0.cs:164-225) inside the pass: the record's curve over the bark indices (floor0_curve_kernel) is staged in
// the wave's LDS row, and every bin is multiplied by the value of ITS bark index -- the reference multiplies a run of bins with
// the same barkMap value by one q (:221-222), i.e. q is a function of the index.  bark: this (floor, block size)'s indices in the
// order a lane holds its bins (SynthArgs.f0_bark), 16 per lane.
__device__ __forceinline__ void floor0_multiply(float2 (&x)[8], float *row, const float *curve, int k_count, const uint16_t *bark,
                                                int lpb, int lane)
{
    const uint4 *b4 = reinterpret_cast<const uint4 *>(bark + (lane & (lpb - 1)) * 16);
    const uint4 b0 = b4[0], b1 = b4[1];
    for (int i = lane; i < k_count; i += 64) row[i] = curve[i];
    __builtin_amdgcn_wave_barrier();
    const uint32_t w[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float t[16];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        t[2 * m] = row[w[m] & 0xFFFFu];
        t[2 * m + 1] = row[w[m] >> 16];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        x[m].x *= t[2 * m];
        x[m].y *= t[2 * m + 1];
    }
    __builtin_amdgcn_wave_barrier();
}

// Both channels at once: every global load of the two curves (and the bark indices) is issued before the first value is used -- copied
// in a loop, 64 values per trip, a curve of 256 bark bands was four trips to the L2 one behind the other, per channel and pass.
// do0 / do1: the channel has a type-0 floor (wave-uniform).  k_count <= kFloor0MaxBark = 1024: at most 16 values per lane.
__device__ __forceinline__ void floor0_multiply_x2(float2 (&x0)[8], float2 (&x1)[8], float *row0, float *row1, const float *curve0,
                                                   const float *curve1, int k_count, const uint16_t *bark0, const uint16_t *bark1,
                                                   bool do0, bool do1, int lpb, int lane)
{
    const VPZ_GLOBAL float *c0 = (const VPZ_GLOBAL float *)(do0 ? curve0 : curve1), *c1 = (const VPZ_GLOBAL float *)(do1 ? curve1 : curve0);
    const VPZ_GLOBAL uint4 *p0 = (const VPZ_GLOBAL uint4 *)((do0 ? bark0 : bark1) + (lane & (lpb - 1)) * 16);
    const VPZ_GLOBAL uint4 *p1 = (const VPZ_GLOBAL uint4 *)((do1 ? bark1 : bark0) + (lane & (lpb - 1)) * 16);
    float v0[16], v1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = lane + 64 * j;
        const int ii = i < k_count ? i : 0;  // (unconditional loads: a lane past the end reads the head again and stores nothing)
        v0[j] = c0[ii];
        v1[j] = c1[ii];
    }
    const uint4 a0 = p0[0], a1 = p0[1], b0 = p1[0], b1 = p1[1];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = lane + 64 * j;
        if (i < k_count) {
            if (do0) row0[i] = v0[j];
            if (do1) row1[i] = v1[j];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t w0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const uint32_t w1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float t0[16], t1[16];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        t0[2 * m] = do0 ? row0[w0[m] & 0xFFFFu] : 1.0f;
        t0[2 * m + 1] = do0 ? row0[w0[m] >> 16] : 1.0f;
        t1[2 * m] = do1 ? row1[w1[m] & 0xFFFFu] : 1.0f;
        t1[2 * m + 1] = do1 ? row1[w1[m] >> 16] : 1.0f;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (do0) {
#pragma unroll
        for (int m = 0; m < 8; ++m) { x0[m].x *= t0[2 * m]; x0[m].y *= t0[2 * m + 1]; }
    }
    if (do1) {
#pragma unroll
        for (int m = 0; m < 8; ++m) { x1[m].x *= t1[2 * m]; x1[m].y *= t1[2 * m + 1]; }
    }
    __builtin_amdgcn_wave_barrier();
}

// kIlvIn : every packet of the batch is the Residue2-interleaved vector [bin][2] (else: every packet planar [2][bin])
// kOut   : 0 planar output, 1 interleaved
// kI16   : the residue is 16-bit integers (VPZ_RESIDUE_I16, ABI v5): half the bytes per value from HBM, widened in registers
//          (v_cvt_f32_i32 of the sign-extended halves: exact) -- floored batches only, others get the widened float32 copy
// kExp: tuning experiments, A/B on one box through VPZ_DUAL_EXP (none at the moment; DESIGN.md 4.7 lists what was tried)
template <bool kHasFloor, bool kIlvIn, int kOut, bool kS16, bool kI16 = false, int kExp = 0>
__global__ __launch_bounds__(kDualThreads, kDualWavesPerSimd) void synth_dual_kernel(SynthArgs a)
{
    using out_t = typename std::conditional<kS16, int16_t, float>::type;
    constexpr bool kInterleavedOut = kOut != 0;
    __shared__ float2 s_twL[512];
    __shared__ float2 s_twAB[512];
    __shared__ float2 s_twBC[64];
    __shared__ float2 s_twS[64];
    __shared__ float s_slope1[1024];
    __shared__ float s_slope0[128];
    __shared__ float s_db[kHasFloor ? 256 : 1];
    __shared__ __attribute__((aligned(8))) uint8_t s_steps[2 * kGroupMaxStepPairs + 8];
    __shared__ PacketGeom s_geom[8];
    __shared__ float s_work[kDualWaves][2][kWaveBufFloats];   // h of the two blocks being built
    __shared__ float s_tail[kDualWaves][2][kWaveTailFloats];  // upper halves of the previous blocks' h
    __shared__ uint4 s_desc[kDualWaves][(kMaxRunLengthDual + 1) * 2];
    __shared__ int s_done[kDualWaves];  // kPreNeighbour: wave w has left the last tail of its run in s_tail[w]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // pairs: the grid is laid out in groups of 8 chunks x n_pairs workgroups -- workgroup id % 8 (its XCD) is its chunk's, the
    // pairs of a chunk are 8 ids apart
    const int n_pairs = kPairs ? a.n_pairs : 1;
    const int pair = kPairs ? ((int)blockIdx.x % (8 * n_pairs)) / 8 : 0;
    const int chunk = kPairs ? ((int)blockIdx.x / (8 * n_pairs)) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
    // (channels of a packet and this workgroup's two, "L" and "R" below: parked with the pointers further down -- nC(), cA(), cB())
    const int C0 = kPairs ? a.channels : 2;
    const int chA0 = kPairs ? (int)a.pair_ch[2 * pair] : 0;
    const int chB0 = kPairs ? (int)a.pair_ch[2 * pair + 1] : 1;
    const uint32_t *map_bits = kPairs ? a.map_bits + (size_t)pair * a.n_mappings : a.map_bits;
    const int run_idx = chunk * kDualWaves + wave;
    const bool active = run_idx < a.n_runs;
#ifdef VPZ_WAVE_TIMES
    const unsigned long long t_wave_begin = __builtin_amdgcn_s_memtime();
#endif
    // A compact run's flag bytes, INLINE: 32 bytes per run (flag byte and mapping index of its first 16 staged frames) at an
    // address that depends on the run's index only -- asked for together with the run record, so that a short run's descriptors
    // cost ONE trip to the pinned host memory they sit in instead of two in a row (record, then the bytes at its `first`)
    uint32_t icf = 0, imp = 0;
    if (a.run_inline != nullptr && lane < 16) {
        const uint8_t *inl = a.run_inline + (size_t)(active ? run_idx : 0) * 32;
        icf = inl[lane];
        imp = inl[16 + lane];
    }
    RunDesc run = a.runs[active ? run_idx : 0];
    if (!active) {
        run.count = 0;
        run.pre_kind = kPreNone;
        run.flags = 0;
    }
    // The input of a frame: 32 registers, va[m] | vb[m] = the 16 bytes lane-point m of an interleaved packet comes in
    // ((L[2k], R[2k]) | (L[2k+1], R[2k+1])), or the point's two 8-byte pairs of a planar one ((L[2k], L[2k+1]) |
    // (R[2k], R[2k+1])).  Every load is UNCONDITIONAL (a frame that needs no input reads the head of the inverse dB
    // table): a load under a condition would make the wave wait for it right behind the load (see synth_kernel).
    // Points of a lane: k = lane + 64 m for a 2048 block; for 256 blocks lane group g = lane >> 3 takes block g of the
    // pass (block 0 again where the pass has fewer), k = (lane & 7) + 8 m.
    // The pointers a pass needs once -- the spectra, the inverse dB table (what a frame without input reads), the active posts -- would
    // be re-loaded from the argument segment once per pass (scalar registers are short; an s_load's wait drains the LDS queue with
    // it): parked in the lanes of a vector register, fetched with v_readlane_b32 (see synth_kernel).
    int kv = 0;
    {
        const uint64_t sp = reinterpret_cast<uint64_t>(a.spec), db = reinterpret_cast<uint64_t>(a.inv_db);
        const uint64_t cpp = a.cposts != nullptr ? reinterpret_cast<uint64_t>(a.cposts) : db;
        const uint64_t cs = (uint64_t)a.channel_stride;
        const int vals[17] = {(int)(uint32_t)sp, (int)(uint32_t)(sp >> 32), (int)(uint32_t)db, (int)(uint32_t)(db >> 32),
                              (int)(uint32_t)cpp, (int)(uint32_t)(cpp >> 32), a.cposts != nullptr ? 1 : 0, a.f0_stride,
                              C0, chA0, chB0, a.size0, a.size1, a.ccount != nullptr ? 1 : 0,
                              (int)(uint32_t)cs, (int)(uint32_t)(cs >> 32), a.clip};
#pragma unroll
        for (int i = 0; i < 17; ++i)
            if (kPairs || i < 8 || i >= 14) kv = lane == i ? vals[i] : kv;
        asm volatile("" : "+v"(kv));
    }
    // (pairs: the block sizes come from the parked values as well -- the frame loop has no scalar register to keep them in)
    auto size_of = [&](uint32_t flags) -> int {
        return kPairs ? __builtin_amdgcn_readlane(kv, (flags & kFrameLong) ? 12 : 11) : ((flags & kFrameLong) ? a.size1 : a.size0);
    };
    auto nC = [&]() -> int { return kPairs ? __builtin_amdgcn_readlane(kv, 8) : 2; };
    auto cA = [&]() -> int { return kPairs ? __builtin_amdgcn_readlane(kv, 9) : 0; };
    auto cB = [&]() -> int { return kPairs ? __builtin_amdgcn_readlane(kv, 10) : 1; };
    // (pairs: adjacent channels are one 8-byte column of an interleaved packet / PCM row)
    auto adjacent = [&]() -> bool { return !kPairs || (cB() == cA() + 1 && !(cA() & 1)); };
    auto parked64 = [&](int i) -> uint64_t {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(kv, i + 1) << 32) | (uint32_t)__builtin_amdgcn_readlane(kv, i);
    };
    auto prefetch = [&](const FrameDesc &fd, bool valid, float2 (&va)[8], float2 (&vb)[8], int &cpa, int &cpb) {
        const bool is_long = size_of(fd.flags) == 2048;
        const int bsz = (int)((fd.flags >> kFrameBatchShift) & 7u) + 1;
        int l = lane;
        asm volatile("" : "+v"(l));  // (frame-invariant lane arithmetic stays inside the iteration that uses it)
        const int g = l >> 3, gg = g < bsz ? g : 0;
        const VPZ_GLOBAL float *src = valid ? (const VPZ_GLOBAL float *)parked64(0) + fd.spec_off : (const VPZ_GLOBAL float *)parked64(2);
        if (kI16) {
            // 16-bit values: a point's four values are 8 bytes (interleaved: L[2k] R[2k] L[2k+1] R[2k+1]) or two words (planar: the L
            // pair and, half a packet on, the R pair); kept as they came -- va[m] holds the two words -- and widened where the frame
            // is taken up, so that nothing here waits for the loads
            const VPZ_GLOBAL int16_t *s16 = valid ? (const VPZ_GLOBAL int16_t *)parked64(0) + fd.spec_off : (const VPZ_GLOBAL int16_t *)parked64(2);
            const int step = !valid ? 0 : (is_long ? 64 : 8);
            if (kIlvIn) {
                const VPZ_GLOBAL uint2 *s8 = (const VPZ_GLOBAL uint2 *)s16;
                const int base = !valid ? 0 : (is_long ? l : 64 * gg + (l & 7));
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const uint2 v = s8[base + step * m];
                    va[m] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
                    vb[m] = make_float2(0.0f, 0.0f);
                }
            } else {
                const VPZ_GLOBAL uint32_t *s4 = (const VPZ_GLOBAL uint32_t *)s16;
                const int base = !valid ? 0 : (is_long ? l : 128 * gg + (l & 7));
                const int rofs = !valid ? 0 : (is_long ? 512 : 64);
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    va[m] = make_float2(__uint_as_float(s4[base + step * m]), __uint_as_float(s4[base + step * m + rofs]));
                    vb[m] = make_float2(0.0f, 0.0f);
                }
            }
        } else if (kIlvIn && kPairs) {
            // columns chA, chB of the [bin][C] vector: point k = bins 2k, 2k + 1 -- two 8-byte pieces (adjacent channels) or four
            // values; block gg of a pass of short blocks is the gg-th packet, C * 128 floats further on
            // Point groups beyond the residue's declared support (ABI v4; the frame's skip field) are +0.0 by the setup header's word:
            // their loads go to a line of zeros behind the inverse dB table instead -- still unconditional, but every lane asks for
            // the same 16 bytes, and a third of every line of the vector's upper part is not pulled through this CU's cache for nothing
            const int live = !valid ? 8 : 8 - (int)((fd.flags >> kFrameSkipShift) & kFrameSkipMask);
            const VPZ_GLOBAL float *zeros = (const VPZ_GLOBAL float *)parked64(2) + 256;
            const int base = !valid ? 0 : (is_long ? 2 * l * nC() : gg * nC() * 128 + 2 * (l & 7) * nC()) + cA();
            const int step = !valid ? 0 : (is_long ? 128 * nC() : 16 * nC());
            const int row = !valid ? 0 : nC();
            if (adjacent()) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const bool z = m >= live;
                    const VPZ_GLOBAL float *pm = z ? zeros : src;
                    const int o = z ? 0 : base + step * m;
                    va[m] = *(const VPZ_GLOBAL float2 *)(pm + o);
                    vb[m] = *(const VPZ_GLOBAL float2 *)(pm + o + (z ? 0 : row));
                }
            } else {  // [census: cold]
                const int db = !valid ? 0 : cB() - cA();
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const bool z = m >= live;
                    const VPZ_GLOBAL float *pm = z ? zeros : src;
                    const int o = z ? 0 : base + step * m, r1 = z ? 0 : row, d1 = z ? 0 : db;
                    va[m] = make_float2(pm[o], pm[o + d1]);
                    vb[m] = make_float2(pm[o + r1], pm[o + r1 + d1]);
                }
            }
        } else if (kIlvIn) {
            const VPZ_GLOBAL float4 *s4 = (const VPZ_GLOBAL float4 *)src;
            const int base = !valid ? 0 : (is_long ? l : 64 * gg + (l & 7));
            const int step = !valid ? 0 : (is_long ? 64 : 8);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float4 v = s4[base + step * m];
                va[m] = make_float2(v.x, v.y);
                vb[m] = make_float2(v.z, v.w);
            }
        } else if (kPairs) {
            // rows chA, chB of the [C][bin] packet (8-byte pieces: two bins of one channel)
            const VPZ_GLOBAL float2 *s2 = (const VPZ_GLOBAL float2 *)src;
            const int hb = is_long ? 512 : 64;  // a row in 8-byte pieces
            const int live = !valid ? 8 : 8 - (int)((fd.flags >> kFrameSkipShift) & kFrameSkipMask);  // (see the interleaved case)
            const VPZ_GLOBAL float2 *zeros = (const VPZ_GLOBAL float2 *)((const VPZ_GLOBAL float *)parked64(2) + 256);
            const int base = !valid ? 0 : (is_long ? l : gg * nC() * 64 + (l & 7)) + cA() * hb;
            const int step = !valid ? 0 : (is_long ? 64 : 8);
            const int rofs = !valid ? 0 : (cB() - cA()) * hb;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const bool z = m >= live;
                const VPZ_GLOBAL float2 *pm = z ? zeros : s2;
                const int o = z ? 0 : base + step * m;
                va[m] = pm[o];
                vb[m] = pm[o + (z ? 0 : rofs)];
            }
        } else {
            const VPZ_GLOBAL float2 *s2 = (const VPZ_GLOBAL float2 *)src;
            // block gg of the pass: its L row at + gg * 2 * 128 floats, its R row half a packet further on
            const int base = !valid ? 0 : (is_long ? l : 128 * gg + (l & 7));
            const int step = !valid ? 0 : (is_long ? 64 : 8);
            const int rofs = !valid ? 0 : (is_long ? 512 : 64);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                va[m] = s2[base + step * m];
                vb[m] = s2[base + step * m + rofs];
            }
        }
        if (kHasFloor) {
            const bool floored = valid && !(fd.flags & kFrameNoFloor) && __builtin_amdgcn_readlane(kv, 6) != 0;  // (a.cposts != nullptr)
            const VPZ_GLOBAL int32_t *cp = (const VPZ_GLOBAL int32_t *)parked64(4);  // (the posts, or the table's head)
            const size_t rec = floored ? (size_t)fd.rec : 0;
            cpa = cp[(floored ? rec + cA() : 0) * 64 + l];
            cpb = cp[(floored ? rec + cB() : 0) * 64 + l];
        }
    };
    {
        const bool has_long = a.size1 == 2048 || a.size0 == 2048;
        const bool has_short = a.size0 == 256 || a.size1 == 256;
        for (int i = threadIdx.x; i < 512; i += kDualThreads) {
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
    for (int i = threadIdx.x; i < a.size1 / 2; i += kDualThreads) s_slope1[i] = a.slope1[i];
    for (int i = threadIdx.x; i < a.size0 / 2 && i < 128; i += kDualThreads) s_slope0[i] = a.slope0[i];
    if (kHasFloor && threadIdx.x < 256) s_db[threadIdx.x] = a.inv_db[threadIdx.x];
    if (threadIdx.x < 8) s_geom[threadIdx.x] = a.geom[threadIdx.x];
    if (threadIdx.x < kDualWaves) s_done[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < 2 * a.n_step_pairs && i < 2 * kGroupMaxStepPairs; i += kDualThreads) s_steps[i] = a.steps[i];
    // ---- what depends on the run record (pinned host memory: one trip, under way since the kernel's first instructions)
    const int fi0 = (run.pre_kind == kPreRecompute && run.count > 0) ? -1 : 0;
    uint32_t cf_early = 0, mp_early = 0;
    int cc_early = 0;
    float2 va[8], vb[8];
    int cpa = 0, cpb = 0;
    bool early_input = false;  // the first staged frame's input is under way already
    if (run.flags & kRunCompact) {
        const int n = run.count - fi0, f0 = run.first + fi0;
        if (a.run_inline != nullptr && n <= 16) {
            cf_early = icf;  // (zero beyond the run's frames, like the lanes the other branch leaves alone)
            mp_early = imp;
        } else if (lane < n) {
            cf_early = a.cflags[f0 + lane];
            mp_early = a.cmap[f0 + lane];
        }
        if (lane < n && kHasFloor && a.ccount != nullptr) {  // both channels' post counts: records 2p, 2p + 1
            if (kPairs)
                cc_early = (int)a.ccount[run.rec_base + lane * nC() + cA()] | ((int)a.ccount[run.rec_base + lane * nC() + cB()] << 8);
            else
                cc_early = *reinterpret_cast<const uint16_t *>(a.ccount + run.rec_base + lane * 2);
        }
        // The first staged frame of a run that starts with a 2048 block: its input is asked for HERE, ahead of the barrier and of
        // the descriptors' derivation (its place is the run's spec_base, its records the run's rec_base whatever the other
        // frames are; a short first frame may head a batch of blocks, which only the derivation knows)
        const uint32_t cf0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cf_early);
        if (n > 0 && (cf0 & 1u) && !(cf0 & kCfSkip) && a.size1 == 2048) {
            FrameDesc fd0{};
            fd0.spec_off = run.spec_base;
            fd0.rec = run.rec_base;
            fd0.flags = kFrameLong | ((cf0 & kCfNoFloor) ? kFrameNoFloor : 0u);
            prefetch(fd0, true, va, vb, cpa, cpb);
            early_input = true;
        }
    }
    __syncthreads();
    if (!active) return;  // the only workgroup barrier is behind us: waves run free from here

    const int half1 = a.size1 >> 1;
#ifndef VPZ_DUAL_NO_SLOPE_REGS
    // The steady state's window values -- the same eight 16-byte pieces of the long slope for every such frame of this lane -- are read
    // ONCE here and kept in 32 registers: 8 KB less through the CU's LDS pipe per pass (it is ~70 % busy under this kernel and what bounds it)
    float4 swl[4], swr[4];
    {
        const float4 *s4 = reinterpret_cast<const float4 *>(s_slope1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            swl[r] = s4[lane + 64 * r];
            swr[r] = s4[255 - (lane + 64 * r)];
        }
    }
#endif
#ifndef VPZ_DUAL_NO_TW_REGS
    // ... and the 2048-point transform's twiddles of this lane (22 entries of three tables, the same for every long block): 44 registers,
    // 11 KB less through the LDS pipe per pass
    // (the variants that keep the tail in registers as well -- see kTailRegs below -- leave the third set in LDS: 256 registers is all there is)
#if defined(VPZ_DUAL_TAIL_REGS) && !defined(VPZ_DUAL_NO_SLOPE_REGS)
    constexpr bool kBCRegs = kHasFloor || kPairs;
#else
    constexpr bool kBCRegs = true;
#endif
    float2 rtw[8], rtwAB[8], rtwBC[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        rtw[m] = s_twL[lane + 64 * m];
        rtwAB[m] = s_twAB[64 * m + lane];
        rtwBC[m] = kBCRegs ? s_twBC[8 * (lane & 7) + m] : make_float2(0.0f, 0.0f);
    }
#endif
    float *hL = s_work[wave][0], *hR = s_work[wave][1];
    float *tailL = s_tail[wave][0], *tailR = s_tail[wave][1];
    int prev_n4 = 0;  // n/4 of the previous block (0: none yet)
    // kPreNeighbour: the run's first frame is emitted in one pass more at the end of the loop (see synth_desc.hpp)
    const bool defer = run.pre_kind == kPreNeighbour && run.count > 0 && wave > 0;
    float4 stashL0, stashL1, stashR0, stashR1;  // h[0:512) of the deferred frame, both channels
    stashL0 = stashL1 = stashR0 = stashR1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // The tail of a 2048 block in REGISTERS (no-floor stereo variants: they have the room): lane l keeps 16-byte pieces l and l + 64 of
    // both channels' tails -- what the steady state's first two output rounds read as they are and its last two mirrored (lane 63 - l's
    // pieces) --, so a steady frame neither writes its tail to LDS nor reads the previous one from there: 12 KB less through the LDS pipe
    // per pass.  Everything else (other geometries, short blocks, the neighbour's hand-over, the saved state) wants the rows in LDS:
    // flush_tail() puts them there first.
    // MEASURED AND NOT ADOPTED (kernel time under rocprofv3, builds alternating, profiles/r5_ab_lds_diet.txt): 222.6-224.6 us with the
    // tail in registers (and the third twiddle set back in LDS: 256 registers is all there is) against 219.2-220.5 without.
    // -DVPZ_DUAL_TAIL_REGS builds it; bit-equal (tests/test_dual_gpu.py, test_full_size_gpu.py under that build).
#if defined(VPZ_DUAL_TAIL_REGS) && !defined(VPZ_DUAL_NO_SLOPE_REGS)
    constexpr bool kTailRegs = !kHasFloor && !kPairs;
#else
    constexpr bool kTailRegs = false;
#endif
    float4 tl0, tl1, tr0, tr1;
    tl0 = tl1 = tr0 = tr1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    bool tail_regs = false;  // (wave-uniform) the current tail is in tl0 .. tr1 and NOT in s_tail
    auto flush_tail = [&]() {
        if (!kTailRegs || !tail_regs) return;
        int lq = lane;
        asm volatile("" : "+v"(lq));
        float4 *dl = reinterpret_cast<float4 *>(tailL), *dr = reinterpret_cast<float4 *>(tailR);
        dl[lq] = tl0;
        dl[lq + 64] = tl1;
        dr[lq] = tr0;
        dr[lq + 64] = tr1;
        __builtin_amdgcn_wave_barrier();
        tail_regs = false;
    };
    bool published = false;
    auto publish = [&]() {  // this wave's s_tail rows hold the upper half of its run's last block from here on
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&s_done[wave], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        published = true;
    };

    // ---- the run's frame descriptors into LDS (explicit ones copied, a compact run's derived here)
    bool batch_member = false;
    if (run.flags & kRunCompact) {
        const int n = run.count - fi0;  // staged frames (<= kMaxRunLengthDual + 1 = 64), one lane each
        const uint32_t cf = cf_early, mp = mp_early;
        const uint32_t pcf = __shfl_up(cf, 1);
        const PacketGeom g = s_geom[cf & 7], pg = s_geom[pcf & 7];
        const bool has_prev = lane > 0 || run.has_prev0;
        const int prev_end = lane > 0 ? pg.right_start : run.prev_end0;
        const int prev_stop = lane > 0 ? pg.right_end : run.prev_stop0;
        int left_start = has_prev ? g.left_start : g.right_start;  // StreamDecoder.cs:674 / :679
        int out_count = has_prev ? max(0, (int)g.right_start - (int)g.left_start) : 0;
        uint32_t fl = ((cf & 1) ? kFrameLong : 0u) | (g.left_use_size1 ? kFrameSlope1 : 0u) |
                      ((cf & kCfNoFloor) ? kFrameNoFloor : 0u);
        if (!(cf & kCfNoFloor)) {  // the mapping's coupling steps and what its setup header says about the residue's support
            const uint32_t mb = map_bits[mp];
            fl |= (mb & 0x00FFFF00u) | ((((cf & 1) ? mb >> kFrameSkipShift : mb >> kMapSkipShortShift) & kFrameSkipMask) << kFrameSkipShift);
        }
        if (cf & kCfSkip) { fl = kFrameDrain; out_count = 0; }
        if ((run.flags & kRunLastTrimmed) && lane == n - 1) { out_count = run.last_out_count; left_start = run.last_left_start; }
        // Batches of SHORT blocks: a short block costs a pass most of what a long one costs (the 256-point transform
        // computes eight copies of one block, the pass's fixed parts do not shrink) for an eighth of the samples, and
        // real streams hold them in streaks.  Up to eight consecutive short blocks of a run go through ONE pass: lane
        // group g takes block g.  A block joins a batch if it is a plain short-after-short step of its predecessor's
        // mapping and kind (floored or not); the first short block after a long one heads a batch.
        {
            const uint32_t pmp = __shfl_up(mp, 1);
            const uint32_t pnf = __shfl_up(cf & kCfNoFloor, 1);
            const bool after_short = prev_end == 128 && prev_stop == 256;
            const bool after_long = a.size1 == 2048 && prev_end == 1472 && prev_stop == 1600;
            const bool base_ok = lane < n && lane >= -fi0 && a.size0 == 256 && a.size1 != 256 && !(cf & 1) && !(cf & kCfSkip) &&
                                 has_prev && out_count == 128 && left_start == 0 && (after_short || after_long) &&
                                 !a.no_batch;
            const bool base_prev = __shfl_up((int)base_ok, 1) != 0 && lane > 0 && after_short;
            const bool brk = !(base_ok && base_prev && mp == pmp && (cf & kCfNoFloor) == pnf);
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
                } else {
                    batch_member = true;
                }
            }
        }
        const int half = (cf & 1) ? (a.size1 >> 1) : (a.size0 >> 1);
        int spec_sz = lane < n ? nC() * half : 0;
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
            hi.x = (uint32_t)(run.rec_base + lane * nC());
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
        for (int i = lane; i < n16; i += 64) {
            uint4 w = src[i];
            if (kPairs && (i & 1) && !(w.w & (kFrameDrain | kFrameNoFloor)))  // (the frame names its mapping: this pair's steps of it)
                w.w = (w.w & ~0x00FFFF00u) | (map_bits[(w.w >> kFrameStepsOffShift) & 0xFFu] & 0x00FFFF00u);
            s_desc[wave][i] = w;
        }
    }
    // post counts of the run's frames (ExecuteChannel), both channels in one register: lane i = the i-th staged frame
    int cc_run = 0;
    if (kHasFloor && a.ccount != nullptr) {
        if (run.flags & kRunCompact) {
            cc_run = cc_early;
        } else {
            __builtin_amdgcn_wave_barrier();
            if (lane < run.count - fi0) {
                const uint8_t *cc = a.ccount + (int)s_desc[wave][2 * lane + 1].x;
                cc_run = (int)cc[cA()] | ((int)cc[cB()] << 8);
            }
        }
    }
    int iters = run.count - fi0 - (int)__popcll(__ballot(batch_member));
    if (VPZ_ABLATE(a) & 64) iters = 0;  // (tuning only: the kernel's prologue and nothing else)
    const int iters_real = iters;       // passes that transform a block
    if (defer && iters > 0) ++iters;    // + the pass that emits the deferred first frame
    __builtin_amdgcn_wave_barrier();
    auto frame_from = [&](const uint4 lo, const uint4 hi) -> FrameDesc {  // a descriptor's two LDS words into SGPRs
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
    auto frame_at = [&](int fi) -> FrameDesc {  // broadcast LDS read, then into SGPRs
        return frame_from(s_desc[wave][(fi - fi0) * 2], s_desc[wave][(fi - fi0) * 2 + 1]);
    };

    // ---- block preceding the run: from the saved state, or recomputed as "frame -1" of the loop
    if (run.pre_kind == kPreState) {
        const float *st = a.state_h + (size_t)run.state_slot * a.state_slot_floats + (size_t)run.stream * nC() * half1;
        prev_n4 = run.prev_long ? (a.size1 >> 2) : (a.size0 >> 2);
        for (int i = lane; i < prev_n4; i += 64) {
            tailL[i] = st[cA() * half1 + i];
            tailR[i] = st[cB() * half1 + i];
        }
    }
    out_t *out_base = reinterpret_cast<out_t *>(a.out) + (a.stream_out_off ? a.stream_out_off[run.stream] : 0);
    {   // wave-uniform, but the offset arrives through a vector load: move the pointer to scalar registers
        const uint64_t ob = reinterpret_cast<uint64_t>(out_base);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)ob), hi = __builtin_amdgcn_readfirstlane((uint32_t)(ob >> 32));
        out_base = reinterpret_cast<out_t *>(((uint64_t)hi << 32) | lo);
    }
    float clip_peak = 0.0f;

    // the first four coupling steps of a frame's mapping, read a frame ahead like the input
    auto steps_word = [&](const FrameDesc &fd, bool valid) -> uint2 {
        const uint32_t off = valid ? 2u * ((fd.flags >> kFrameStepsOffShift) & kFrameStepsOffMask) : 0u;
        return *reinterpret_cast<const uint2 *>(s_steps + (off < 2u * kGroupMaxStepPairs ? off : 0u));
    };

    FrameDesc fd_next = frame_at(fi0);
    bool valid_cur = run.count > 0 && !(fd_next.flags & kFrameDrain);
    if (!early_input) prefetch(fd_next, valid_cur, va, vb, cpa, cpb);
    uint2 stwcur = steps_word(fd_next, valid_cur);
    // the first frame's input has to be there before the loop is entered (see synth_kernel: wait-count bookkeeping)
#pragma unroll
    for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(va[m].x), "v"(va[m].y), "v"(vb[m].x), "v"(vb[m].y));
    if (kHasFloor) asm volatile("" ::"v"(cpa), "v"(cpb));

#ifdef VPZ_STAMPS
    unsigned long long t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
    unsigned long long n_long_frames = 0;
#endif
#ifdef VPZ_WAVE_TIMES
    unsigned wt_long = 0, wt_short = 0, wt_batch = 0, wt_members = 0;
#endif
    int fi = fi0;
    for (int it = 0; it < iters; ++it) {
        const FrameDesc fd = fd_next;
        const int bsz = (int)((fd.flags >> kFrameBatchShift) & 7u) + 1;  // blocks this pass covers
        const int fin = fi + bsz;
        float2 na[8], nb[8];
        int cpna = 0, cpnb = 0;
        uint2 stwnext;
        bool valid_next;
        const bool first_deferred = defer && it == 0;            // transform and park: no PCM yet
        const bool deferred_pass = defer && it == iters_real;     // no transform: the parked frame's PCM
        {
            const bool has_next = fin < run.count;
            fd_next = frame_at(has_next ? fin : (defer ? 0 : fi));  // (behind the run's last frame: the deferred one again)
            valid_next = has_next && !(fd_next.flags & kFrameDrain) && !(VPZ_ABLATE(a) & 4);  // (4: tuning only, no input loads)
            prefetch(fd_next, valid_next, na, nb, cpna, cpnb);
            stwnext = steps_word(fd_next, valid_next);
        }
        VPZ_STAMP(0);  // descriptor + prefetch issue
#ifdef VPZ_STAMPS
        if (size_of(fd.flags) == 2048) ++n_long_frames;
#endif
#ifdef VPZ_WAVE_TIMES
        if (size_of(fd.flags) == 2048) ++wt_long; else if (bsz > 1) { ++wt_batch; wt_members += bsz; } else ++wt_short;
#endif
        const bool drain = fd.flags & kFrameDrain;
        const bool batch = bsz > 1;
        const int nblk = size_of(fd.flags);
        const bool is_long = nblk == 2048;
        const int n4 = is_long ? 512 : 64;
        const bool no_floor = !kHasFloor || (fd.flags & kFrameNoFloor) || (kPairs ? __builtin_amdgcn_readlane(kv, 13) == 0 : a.ccount == nullptr);
        const int slot = fi - fi0;
        // ABI v4: point groups m >= 8 - skip lie beyond the residue's support -- zeros by the setup header's word (their loads
        // stay: the vector is in memory with its zeros; what is saved is the arithmetic).  Halves are all this path looks at.
        const bool upper = ((fd.flags >> kFrameSkipShift) & kFrameSkipMask) < 4;
        const float *ptL = tailL, *ptR = tailR;  // the previous block's tail: this wave's own rows, or -- deferred pass -- the neighbour's

        if (deferred_pass) {
            // the neighbour's last tail save is behind it once its flag is up (it raised it right after; it waits for nobody)
            // (the bound is a safety net against a hung device should the flag never come up: about a second, then wrong PCM)
            int spins = 0;
            while (__hip_atomic_load(&s_done[wave - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 && ++spins < (1 << 23))
                __builtin_amdgcn_s_sleep(4);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            ptL = s_tail[wave - 1][0];
            ptR = s_tail[wave - 1][1];
            prev_n4 = 512;
            int ls = lane;
            asm volatile("" : "+v"(ls));
            float4 *dl = reinterpret_cast<float4 *>(hL), *dr = reinterpret_cast<float4 *>(hR);
            dl[ls] = stashL0;
            dl[ls + 64] = stashL1;
            dr[ls] = stashR0;
            dr[ls + 64] = stashR1;
            __builtin_amdgcn_wave_barrier();
        } else if (!drain) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            // ---- the two spectra of this lane's points
            float2 xL[8], xR[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (kI16) {
                    const int w0 = (int)__float_as_uint(va[m].x), w1 = (int)__float_as_uint(va[m].y);
                    const float lo0 = (float)(int)(int16_t)(w0 & 0xFFFF), hi0 = (float)(w0 >> 16);
                    const float lo1 = (float)(int)(int16_t)(w1 & 0xFFFF), hi1 = (float)(w1 >> 16);
                    if (kIlvIn) {  // w0 = L[2k] | R[2k] << 16, w1 = L[2k+1] | R[2k+1] << 16
                        xL[m] = make_float2(lo0, lo1);
                        xR[m] = make_float2(hi0, hi1);
                    } else {       // w0 = L[2k] | L[2k+1] << 16, w1 = R[2k] | R[2k+1] << 16
                        xL[m] = make_float2(lo0, hi0);
                        xR[m] = make_float2(lo1, hi1);
                    }
                } else if (kIlvIn) {
                    xL[m] = make_float2(va[m].x, vb[m].x);
                    xR[m] = make_float2(va[m].y, vb[m].y);
                } else {
                    xL[m] = va[m];
                    xR[m] = vb[m];
                }
            }
            // ---- inverse coupling, steps in reverse order (Mapping.cs:166-172); a stereo step is (0, 1) or (1, 0)
            if (kHasFloor && !(fd.flags & kFrameNoFloor) && !(VPZ_ABLATE(a) & 16)) {  // (a batch without floors has no coupling either)
                const int n_steps = (int)((fd.flags >> kFrameStepsShift) & 0xFF);
                const uint8_t *st = s_steps + 2 * ((fd.flags >> kFrameStepsOffShift) & kFrameStepsOffMask);
                const unsigned long long stw = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)stwcur.y) << 32) |
                                               (uint32_t)__builtin_amdgcn_readfirstlane((int)stwcur.x);
                for (int i = n_steps - 1; i >= 0; --i) {
                    const uint32_t mag = (n_steps <= 4 ? (uint32_t)(stw >> (16 * i)) : (uint32_t)st[2 * i]) & 0x7Fu;
                    // (the upper half of the points only if the residue's support reaches it: zeros de-couple to zeros)
                    if (mag == 0) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) { couple(xL[m].x, xR[m].x); couple(xL[m].y, xR[m].y); }
                        if (upper) {
#pragma unroll
                            for (int m = 4; m < 8; ++m) { couple(xL[m].x, xR[m].x); couple(xL[m].y, xR[m].y); }
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < 4; ++m) { couple(xR[m].x, xL[m].x); couple(xR[m].y, xL[m].y); }
                        if (upper) {
#pragma unroll
                            for (int m = 4; m < 8; ++m) { couple(xR[m].x, xL[m].x); couple(xR[m].y, xL[m].y); }
                        }
                    }
                }
            }
            VPZ_STAMP(1);  // unpack + coupling
            // ---- Floor1 curves x spectra (Floor1.cs:222-268), silence (Mapping.cs:190-194)
            bool silentL = false, silentR = false;  // per lane: this lane's block of the pass is a silent channel's
            bool f0L = false, f0R = false;          // (wave-uniform) the channel's floor is type 0
            if (!no_floor) {
                uint8_t *rowL = reinterpret_cast<uint8_t *>(hL), *rowR = reinterpret_cast<uint8_t *>(hR);
                int *auxL = reinterpret_cast<int *>(hL) + 256, *auxR = reinterpret_cast<int *>(hR) + 256;
                uint32_t fyL[4] = {0, 0, 0, 0}, fyR[4] = {0, 0, 0, 0};
                if (!batch) {
                    const int cc = __builtin_amdgcn_readlane(cc_run, slot);
                    const int cntL = cc & 0xFF, cntR = (cc >> 8) & 0xFF;
                    silentL = cntL == 0;
                    silentR = cntR == 0;
                    const int lpb = is_long ? 64 : 8;
                    const int n = nblk >> 1;
                    f0L = !kPairs && cntL == kFloor0Marker;  // (type-0 floors: no curve to render, see floor0_multiply below;
                    f0R = !kPairs && cntR == kFloor0Marker;  // the pair route is not taken by setups that have them)
                    const int nrL = (silentL || f0L || (VPZ_ABLATE(a) & 512)) ? 0 : 2 * (spectrum_top(xL, lpb, ln) + 1);
                    const int nrR = (silentR || f0R || (VPZ_ABLATE(a) & 512)) ? 0 : 2 * (spectrum_top(xR, lpb, ln) + 1);
                    const int pa = ln < cntL ? cpa : 0, pb = ln < cntR ? cpb : 0;
                    if ((nrL > 0 || nrR > 0) && !(VPZ_ABLATE(a) & 8)) {
                        if (!render_floor_indices_fast_x2(rowL, rowR, auxL, auxR, n, nrL, nrR, pa, pb, cntL, cntR, ln)) {  // [census: cold]
                            if (nrL > 0) render_floor_indices<32>(rowL, auxL, n, nrL, pa, cntL, ln);
                            if (nrR > 0) render_floor_indices<32>(rowR, auxR, n, nrR, pb, cntR, ln);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (!(VPZ_ABLATE(a) & 256)) {
                    load_floor_indices(fyL, rowL, lpb, ln);
                    load_floor_indices(fyR, rowR, lpb, ln);
                    }
                    __builtin_amdgcn_wave_barrier();
                } else {  // [census: cold]
                    // the curves of the pass's blocks, 128 bytes each, block by block (both channels side by side); their
                    // posts are asked for here together
                    int cps[8][2];
#pragma unroll
                    for (int f = 0; f < 8; ++f) {
                        const int ff = f < bsz ? f : 0;
                        const int32_t *posts = kPairs ? (const int32_t *)parked64(4) : a.cposts;
                        cps[f][0] = posts[(size_t)(fd.rec + nC() * ff + cA()) * 64 + ln];
                        cps[f][1] = posts[(size_t)(fd.rec + nC() * ff + cB()) * 64 + ln];
                    }
#pragma unroll
                    for (int f = 0; f < 8; ++f) {
                        if (f < bsz) {
                            const int cc = __builtin_amdgcn_readlane(cc_run, slot + f);
                            const int cntL = cc & 0xFF, cntR = (cc >> 8) & 0xFF;
                            if ((ln >> 3) == f) {
                                silentL = cntL == 0;
                                silentR = cntR == 0;
                            }
                            const int pa = ln < cntL ? cps[f][0] : 0, pb = ln < cntR ? cps[f][1] : 0;
                            const int nrL = cntL ? 128 : 0, nrR = cntR ? 128 : 0;
                            if (nrL | nrR) {
                                if (!render_floor_indices_fast_x2(rowL + 128 * f, rowR + 128 * f, auxL, auxR, 128, nrL, nrR, pa, pb,
                                                                  cntL, cntR, ln)) {
                                    if (nrL) render_floor_indices<32>(rowL + 128 * f, auxL, 128, 128, pa, cntL, ln);
                                    if (nrR) render_floor_indices<32>(rowR + 128 * f, auxR, 128, 128, pb, cntR, ln);
                                }
                                __builtin_amdgcn_wave_barrier();
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    load_floor_indices(fyL, rowL + 128 * (ln >> 3), 8, ln);
                    load_floor_indices(fyR, rowR + 128 * (ln >> 3), 8, ln);
                    __builtin_amdgcn_wave_barrier();
                }
                VPZ_STAMP(2);  // curves
                if (!(VPZ_ABLATE(a) & 256)) {
                if (!f0L) apply_floor(xL, fyL, s_db, upper);
                if (!f0R) apply_floor(xR, fyR, s_db, upper);
                if (f0L || f0R) {  // [census: cold]
                    const int lpb0 = is_long ? 64 : 8;
                    const int f0s = __builtin_amdgcn_readlane(kv, 7);  // (a.f0_stride)
                    const int kc = min(f0s, kFloor0MaxBark);
                    // (a channel without a type-0 floor: its pointers are the other channel's, its values are not touched)
                    const size_t fl_l = f0L ? (size_t)__builtin_amdgcn_readlane(cpa, 0) : 0, fl_r = f0R ? (size_t)__builtin_amdgcn_readlane(cpb, 0) : 0;
                    floor0_multiply_x2(xL, xR, hL, hR, a.f0_curve + (size_t)(fd.rec + cA()) * f0s, a.f0_curve + (size_t)(fd.rec + cB()) * f0s, kc,
                                       a.f0_bark + (fl_l * 2 + (is_long ? 1 : 0)) * 1024, a.f0_bark + (fl_r * 2 + (is_long ? 1 : 0)) * 1024,
                                       f0L, f0R, lpb0, ln);
                }
                }
            }
            // ---- the two transforms, side by side
            if (VPZ_ABLATE(a) & 2) {  // [census: cold]
                // (tuning only: the spectra as they are)
                float2 *h2L = reinterpret_cast<float2 *>(hL), *h2R = reinterpret_cast<float2 *>(hR);
#pragma unroll
                for (int m = 0; m < 8; ++m) { h2L[ln + 64 * m] = xL[m]; h2R[ln + 64 * m] = xR[m]; }
            } else if (is_long) {
#ifdef VPZ_DUAL_NO_TW_REGS
                imdct2048_wave_x2(xL, xR, reinterpret_cast<float2 *>(hL), reinterpret_cast<float2 *>(hR), s_twL, s_twAB, s_twBC, ln);
#else
                imdct2048_wave_x2_regs<kBCRegs>(xL, xR, reinterpret_cast<float2 *>(hL), reinterpret_cast<float2 *>(hR), rtw, rtwAB, rtwBC, s_twBC, ln);
#endif
            } else {
                imdct256_wave8_x2(xL, xR, reinterpret_cast<float2 *>(hL), reinterpret_cast<float2 *>(hR), s_twS, s_twBC, ln);
            }
            // the reference does not transform a silent channel, it clears the block (Mapping.cs:190-194): all +0.0,
            // where the transform of zeros (times a curve left over in the row) leaves zeros of both signs or worse
            if (__any(silentL || silentR)) {  // [census: cold]
                __builtin_amdgcn_wave_barrier();
                float2 *h2L = reinterpret_cast<float2 *>(hL), *h2R = reinterpret_cast<float2 *>(hR);
                if (is_long) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (silentL) h2L[ln + 64 * q] = make_float2(0.0f, 0.0f);
                        if (silentR) h2R[ln + 64 * q] = make_float2(0.0f, 0.0f);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (silentL) h2L[(ln >> 3) * 64 + (ln & 7) + 8 * q] = make_float2(0.0f, 0.0f);
                        if (silentR) h2R[(ln >> 3) * 64 + (ln & 7) + 8 * q] = make_float2(0.0f, 0.0f);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }

        VPZ_STAMP(3);  // floor multiply + transforms
        // gfx950's vmcnt counts stores as well as loads, in issue order: wait for the prefetched input HERE, ahead of
        // this frame's stores
#pragma unroll
        for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(na[m].x), "v"(na[m].y), "v"(nb[m].x), "v"(nb[m].y));
        if (kHasFloor) asm volatile("" ::"v"(cpna), "v"(cpnb));

        VPZ_STAMP(4);  // wait for the next frame's input
        // ---- window + overlap-add + clip + store (StreamDecoder.cs:782-789, 515-638)
        // 4 consecutive samples of both channels, sample 4 g .. 4 g + 3 of the pass's output
        out_t *row_i = out_base + fd.out_off * nC() + cA();                    // interleaved: sample s at [C s + chA], [C s + chB]
        const int64_t ch_stride = (int64_t)parked64(14);                    // (a.channel_stride, a.clip: parked like the pointers)
        const bool clip_on = __builtin_amdgcn_readlane(kv, 16) != 0;
        out_t *row_l = out_base + cA() * ch_stride + fd.out_off;            // planar
        out_t *row_r = out_base + cB() * ch_stride + fd.out_off;
        // (pairs, interleaved: a sample's two values are one 8-byte -- 4-byte for 16-bit PCM -- piece when the channels are adjacent)
        const bool aligned = kInterleavedOut ? (kPairs ? adjacent() && (reinterpret_cast<uintptr_t>(row_i) & (kS16 ? 3 : 7)) == 0
                                                       : (reinterpret_cast<uintptr_t>(row_i) & 15) == 0)
                                             : ((reinterpret_cast<uintptr_t>(row_l) | reinterpret_cast<uintptr_t>(row_r)) & (kS16 ? 7 : 15)) == 0;
        auto emit4 = [&](int g, float l0, float l1, float l2, float l3, float r0, float r1, float r2, float r3) {
            if (VPZ_ABLATE(a) & 32) {  // (tuning only: the arithmetic without the stores)
                clip_peak = fmaxf(clip_peak, fmaxf(fmaxf(l0 + r0, l1 + r1), fmaxf(l2 + r2, l3 + r3)));
                return;
            }
            if (clip_on) {
                clip_group(l0, l1, l2, l3, clip_peak);
                clip_group(r0, r1, r2, r3, clip_peak);
            }
            if (kInterleavedOut && kPairs) {
                // four samples' columns chA, chB of the [sample][C] rows: plain stores -- the other pairs' workgroups write the rest
                // of these lines, and the L2 they share puts them together
                out_t *q = row_i + (size_t)(4 * g) * nC();
                if (kS16) {
                    store_pcm(reinterpret_cast<uint32_t *>(q), pack_s16(l0, r0));
                    store_pcm(reinterpret_cast<uint32_t *>(q + nC()), pack_s16(l1, r1));
                    store_pcm(reinterpret_cast<uint32_t *>(q + 2 * nC()), pack_s16(l2, r2));
                    store_pcm(reinterpret_cast<uint32_t *>(q + 3 * nC()), pack_s16(l3, r3));
                } else {
                    float *qf = reinterpret_cast<float *>(q);
                    store_pcm2(qf, l0, r0);
                    store_pcm2(qf + nC(), l1, r1);
                    store_pcm2(qf + 2 * nC(), l2, r2);
                    store_pcm2(qf + 3 * nC(), l3, r3);
                }
            } else if (kInterleavedOut) {
                if (kS16) {
                    store_nt(reinterpret_cast<uint4 *>(row_i) + g, pack_s16(l0, r0), pack_s16(l1, r1), pack_s16(l2, r2), pack_s16(l3, r3));
                } else {
                    store_pcm4_pair(reinterpret_cast<float4 *>(row_i) + 2 * g, make_float4(l0, r0, l1, r1), make_float4(l2, r2, l3, r3));
                }
            } else if (kS16) {
                store_nt(reinterpret_cast<uint2 *>(row_l) + g, pack_s16(l0, l1), pack_s16(l2, l3));
                store_nt(reinterpret_cast<uint2 *>(row_r) + g, pack_s16(r0, r1), pack_s16(r2, r3));
            } else {
                store_pcm4(reinterpret_cast<float4 *>(row_l) + g, make_float4(l0, l1, l2, l3));
                store_pcm4(reinterpret_cast<float4 *>(row_r) + g, make_float4(r0, r1, r2, r3));
            }
        };
        auto emit1 = [&](int i, float l, float r) {  // sample i of the pass, no alignment assumed
            if (clip_on) {
                l = clip_track(l, clip_peak);
                r = clip_track(r, clip_peak);
            }
            if (kInterleavedOut) {
                store_pcm(row_i + (size_t)i * nC(), kS16 ? (out_t)to_s16(l) : (out_t)l);
                store_pcm(row_i + (size_t)i * nC() + (cB() - cA()), kS16 ? (out_t)to_s16(r) : (out_t)r);
            } else {
                store_pcm(row_l + i, kS16 ? (out_t)to_s16(l) : (out_t)l);
                store_pcm(row_r + i, kS16 ? (out_t)to_s16(r) : (out_t)r);
            }
        };
        const float4 *hL4 = reinterpret_cast<const float4 *>(hL), *hR4 = reinterpret_cast<const float4 *>(hR);
        const float4 *tL4 = reinterpret_cast<const float4 *>(ptL), *tR4 = reinterpret_cast<const float4 *>(ptR);
        int lf = lane;
        asm volatile("" : "+v"(lf));  // (no address of the epilogue may be computed ahead of the frame loop)
        const bool reg_steady = kTailRegs && tail_regs && !deferred_pass && !batch && fi >= 0 && !first_deferred && !drain && aligned &&
                                is_long && (fd.flags & kFrameSlope1) && fd.left_start == 0 && fd.packet_len == 1024 && fd.prev_end == 1024 &&
                                fd.out_count == 1024 && prev_n4 == 512 && a.size1 == 2048;
        if (kTailRegs && tail_regs && !reg_steady && !deferred_pass && !first_deferred) flush_tail();  // (whoever reads the tail below reads LDS)
        if (VPZ_ABLATE(a) & 1) {
            // (tuning only: no window / overlap-add / stores)
#ifndef VPZ_DUAL_NO_SLOPE_REGS
        } else if (reg_steady) {
            // the steady state with the previous tail in registers: rounds 0, 1 over this lane's own pieces, rounds 2, 3 over lane
            // 63 - l's (tL4[255 - g] = piece 127 - l, 63 - l); same operations and operands as the LDS form below
            float4 ml1, mr1, ml0, mr0;
            lane_mirror64_x2(tl1.x, tr1.x, ml1.x, mr1.x);
            lane_mirror64_x2(tl1.y, tr1.y, ml1.y, mr1.y);
            lane_mirror64_x2(tl1.z, tr1.z, ml1.z, mr1.z);
            lane_mirror64_x2(tl1.w, tr1.w, ml1.w, mr1.w);
            lane_mirror64_x2(tl0.x, tr0.x, ml0.x, mr0.x);
            lane_mirror64_x2(tl0.y, tr0.y, ml0.y, mr0.y);
            lane_mirror64_x2(tl0.z, tr0.z, ml0.z, mr0.z);
            lane_mirror64_x2(tl0.w, tr0.w, ml0.w, mr0.w);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = lf + 64 * r;
                const float4 wl = swl[r], wr = swr[r];
                if (r < 2) {
                    const float4 hl = hL4[127 - g], hr = hR4[127 - g];
                    const float4 pl = r == 0 ? tl0 : tl1, pr = r == 0 ? tr0 : tr1;
                    emit4(g, ola(-hl.w, wl.x, pl.x, wr.w), ola(-hl.z, wl.y, pl.y, wr.z), ola(-hl.y, wl.z, pl.z, wr.y),
                          ola(-hl.x, wl.w, pl.w, wr.x), ola(-hr.w, wl.x, pr.x, wr.w), ola(-hr.z, wl.y, pr.y, wr.z),
                          ola(-hr.y, wl.z, pr.z, wr.y), ola(-hr.x, wl.w, pr.w, wr.x));
                } else {
                    const float4 hl = hL4[g - 128], hr = hR4[g - 128];
                    const float4 pl = r == 2 ? ml1 : ml0, pr = r == 2 ? mr1 : mr0;
                    emit4(g, ola(hl.x, wl.x, pl.w, wr.w), ola(hl.y, wl.y, pl.z, wr.z), ola(hl.z, wl.z, pl.y, wr.y),
                          ola(hl.w, wl.w, pl.x, wr.x), ola(hr.x, wl.x, pr.w, wr.w), ola(hr.y, wl.y, pr.z, wr.z),
                          ola(hr.z, wl.z, pr.y, wr.y), ola(hr.w, wl.w, pr.x, wr.x));
                }
            }
#endif
        } else if (batch && fi >= 0) {  // [census: cold]
            // ---- a batch of short blocks: 128 * bsz contiguous samples.  Sample i of block f is y_f[i] over the previous
            // block's y[128 + i] (both windows short): y_f[i] = -h_f[63 - i] (i < 64), h_f[i - 64] otherwise; the partner
            // is hp[i] (i < 64), hp[127 - i] otherwise, hp = the upper half of the previous block's h -- the block before
            // in the row, or the tail for block 0 (after a long block: floats 448..511 of its tail).
            const float4 *s4 = reinterpret_cast<const float4 *>(s_slope0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gq = lf + 64 * r;
                const int f = gq >> 5, i4 = gq & 31;
                const bool lower = i4 < 16;
                const bool valid = f < bsz;
                const int fs = valid ? f : 0;
                const float4 wl = s4[i4], wr = s4[31 - i4];
                const int hidx = fs * 32 + (lower ? 15 - i4 : i4 - 16);
                const int pidx = lower ? i4 : 31 - i4;
                // (the partner rows are LDS either way: select the address, not the loaded value)
                const float4 *ppl = fs > 0 ? hL4 + (fs - 1) * 32 + 16 : tL4 + (prev_n4 == 512 ? 112 : 0);
                const float4 *ppr = fs > 0 ? hR4 + (fs - 1) * 32 + 16 : tR4 + (prev_n4 == 512 ? 112 : 0);
                const float4 hl = hL4[hidx], hr = hR4[hidx];
                const float4 pl = ppl[pidx], pr = ppr[pidx];
                const float4 vl = apply_y4(hl, lower, lower), vr = apply_y4(hr, lower, lower);
                const float4 ql = apply_y4(pl, !lower, false), qr = apply_y4(pr, !lower, false);
                const float l0 = ola(vl.x, wl.x, ql.x, wr.w), l1 = ola(vl.y, wl.y, ql.y, wr.z);
                const float l2 = ola(vl.z, wl.z, ql.z, wr.y), l3 = ola(vl.w, wl.w, ql.w, wr.x);
                const float r0 = ola(vr.x, wl.x, qr.x, wr.w), r1 = ola(vr.y, wl.y, qr.y, wr.z);
                const float r2 = ola(vr.z, wl.z, qr.z, wr.y), r3 = ola(vr.w, wl.w, qr.w, wr.x);
                if (valid) {
                    if (aligned) {
                        emit4(gq, l0, l1, l2, l3, r0, r1, r2, r3);
                    } else {
                        emit1(4 * gq, l0, r0);
                        emit1(4 * gq + 1, l1, r1);
                        emit1(4 * gq + 2, l2, r2);
                        emit1(4 * gq + 3, l3, r3);
                    }
                }
            }
        } else if (fi >= 0 && fd.out_count > 0 && !first_deferred) {
            // equal block sizes share one slope table (s_slope0 only holds a 128-entry short slope)
            const float *slope = ((fd.flags & kFrameSlope1) || a.size0 == a.size1) ? s_slope1 : s_slope0;
            const int plen = fd.packet_len;
            // every window boundary of the 256 / 2048 geometries is a multiple of 64 samples, so unless an EOS trim cut the
            // packet a group of four samples never straddles a mirror / overlap boundary
            const bool vec = !drain && aligned && ((fd.out_count | fd.left_start | plen | fd.prev_end) & 3) == 0;
            // (the descriptor carries the steady state as ONE bit too -- kFrameSteady --, and testing that bit instead of the fields
            // is SLOWER: north_star's line 0.236-0.241 ms with the field tests, 0.245-0.261 with the bit, alternating on one box;
            // -DVPZ_DUAL_STEADY_BIT builds the bit test)
#ifdef VPZ_DUAL_STEADY_BIT
            if ((fd.flags & kFrameSteady) && aligned && prev_n4 == 512) {
#else
            if (vec && is_long && (fd.flags & kFrameSlope1) && fd.left_start == 0 && plen == 1024 && fd.prev_end == 1024 &&
                fd.out_count == 1024 && prev_n4 == 512 && a.size1 == 2048) {
#endif
                // long after long with long windows on both sides (the steady state of every stream): first half of the
                // output = negated mirror of h[0:512) over the straight previous tail, second half = h[0:512) straight over
                // the mirrored tail; the window values are read once for both channels
#ifdef VPZ_DUAL_NO_SLOPE_REGS
                const float4 *s4 = reinterpret_cast<const float4 *>(s_slope1);
#endif
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int g = lf + 64 * r;
#ifdef VPZ_DUAL_NO_SLOPE_REGS
                    const float4 wl = s4[g], wr = s4[255 - g];
#else
                    const float4 wl = swl[r], wr = swr[r];
#endif
                    if (r < 2) {
                        const float4 hl = hL4[127 - g], pl = tL4[g], hr = hR4[127 - g], pr = tR4[g];
                        emit4(g, ola(-hl.w, wl.x, pl.x, wr.w), ola(-hl.z, wl.y, pl.y, wr.z), ola(-hl.y, wl.z, pl.z, wr.y),
                              ola(-hl.x, wl.w, pl.w, wr.x), ola(-hr.w, wl.x, pr.x, wr.w), ola(-hr.z, wl.y, pr.y, wr.z),
                              ola(-hr.y, wl.z, pr.z, wr.y), ola(-hr.x, wl.w, pr.w, wr.x));
                    } else {
                        const float4 hl = hL4[g - 128], pl = tL4[255 - g], hr = hR4[g - 128], pr = tR4[255 - g];
                        emit4(g, ola(hl.x, wl.x, pl.w, wr.w), ola(hl.y, wl.y, pl.z, wr.z), ola(hl.z, wl.z, pl.y, wr.y),
                              ola(hl.w, wl.w, pl.x, wr.x), ola(hr.x, wl.x, pr.w, wr.w), ola(hr.y, wl.y, pr.z, wr.z),
                              ola(hr.z, wl.z, pr.y, wr.y), ola(hr.w, wl.w, pr.x, wr.x));
                    }
                }
            } else if (vec) {  // [census: cold]
                // any other aligned geometry, branch-free: lanes past the end clamp their reads and skip only the store;
                // samples past the overlap take weights (1, 0)
                const float4 *s4 = reinterpret_cast<const float4 *>(slope);
                const int cnt4 = fd.out_count >> 2;
                const int nr = (cnt4 + 63) >> 6;
                const int pn4 = prev_n4;
                for (int r = 0; r < nr; ++r) {
                    const int g = lf + 64 * r;
                    const bool lv = g < cnt4;
                    const int i = (lv ? g : cnt4 - 1) << 2;
                    const Y4Map mc = map_y4(fd.left_start + i, n4);
                    const bool in = i < plen;
                    const int ii = in ? i : 0;
                    const int q = fd.prev_end + ii;  // in [2 pn4, 4 pn4) whenever `in`
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
                    if (lv) emit4(g, l0, l1, l2, l3, r0, r1, r2, r3);
                }
            } else {
                for (int i = lf; i < fd.out_count; i += 64) {  // [census: cold]
                    float l, r;
                    if (drain) {
                        l = tail_at(ptL, fd.prev_end + i, prev_n4);
                        r = tail_at(ptR, fd.prev_end + i, prev_n4);
                    } else {
                        l = y_from_h(hL, fd.left_start + i, n4);
                        r = y_from_h(hR, fd.left_start + i, n4);
                        if (i < plen) {
                            const float wl = slope[i], wr = slope[plen - 1 - i];
                            l = ola(l, wl, tail_at(ptL, fd.prev_end + i, prev_n4), wr);
                            r = ola(r, wl, tail_at(ptR, fd.prev_end + i, prev_n4), wr);
                        }
                    }
                    emit1(i, l, r);
                }
            }
        }
        VPZ_STAMP(5);  // window + overlap-add + stores
        // ---- keep what a later block can overlap with: y[N/2 .. N) lives in the upper half of h
        if (!drain && !deferred_pass && !(VPZ_ABLATE(a) & 1024)) {
            __builtin_amdgcn_wave_barrier();
            if (first_deferred) {  // (a steady frame: a 2048 block) what its PCM needs of h, parked until the run's end
                stashL0 = hL4[lf];
                stashL1 = hL4[lf + 64];
                stashR0 = hR4[lf];
                stashR1 = hR4[lf + 64];
            }
            if (is_long) {
                const float4 *sl = reinterpret_cast<const float4 *>(hL + 512), *sr = reinterpret_cast<const float4 *>(hR + 512);
                float4 *dl = reinterpret_cast<float4 *>(tailL), *dr = reinterpret_cast<float4 *>(tailR);
                const float4 a0 = sl[lf], a1 = sl[lf + 64], b0 = sr[lf], b1 = sr[lf + 64];
                if (kTailRegs && a.size1 == 2048) {  // (kept in registers; flush_tail() for whoever wants the rows)
                    tl0 = a0;
                    tl1 = a1;
                    tr0 = b0;
                    tr1 = b1;
                    tail_regs = true;
                } else {
                    dl[lf] = a0;
                    dl[lf + 64] = a1;
                    dr[lf] = b0;
                    dr[lf + 64] = b1;
                }
            } else {
                const int o = (batch ? 128 * (bsz - 1) : 0) + 64 + lf;  // (a batch: its last block)
                const float tl = hL[o], tr = hR[o];
                tailL[lf] = tl;
                tailR[lf] = tr;
            }
            prev_n4 = n4;
            __builtin_amdgcn_wave_barrier();
        }
        if (it + 1 == iters_real) {  // the run's last tail is in place (in LDS): the next wave of the workgroup may overlap with it
            flush_tail();
            publish();
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            va[m] = na[m];
            vb[m] = nb[m];
        }
        cpa = cpna;
        cpb = cpnb;
        stwcur = stwnext;
        fi = fin;
        VPZ_STAMP(6);  // tail
    }
#ifdef VPZ_STAMPS
    if (a.stamps && lane == 0) {
        unsigned long long tot = 0;
        for (int k = 0; k < 9; ++k) { atomicAdd(&a.stamps[k], t_acc[k]); tot += t_acc[k]; }
        atomicAdd(&a.stamps[15], 1ull);
        atomicMax(&a.stamps[14], tot);                      // slowest wave
        atomicAdd(&a.stamps[13], tot * tot / 1000000ull);   // (for the spread)
        atomicAdd(&a.stamps[12], (unsigned long long)iters);
        if (run_idx < (1 << 16)) {  // this wave's own record (VPZ_STAMPS_DUMP)
            unsigned long long *rec = a.stamps + 16 + 16 * (size_t)run_idx;
            for (int k = 0; k < 9; ++k) rec[k] = t_acc[k];
#ifdef VPZ_WAVE_TIMES
            rec[2] = wt_long; rec[3] = wt_short; rec[4] = wt_batch; rec[5] = wt_members;  // passes by kind, blocks in batches
#endif
            rec[6] = (unsigned long long)run.pre_kind;
            rec[9] = (unsigned long long)iters;
            rec[10] = n_long_frames;
        }
    }
#endif

    if (!published) publish();  // (a run without passes: nobody may wait for it for ever)
    // ---- keep the last block's tail for the next batch (the reference keeps _prevPacketBuf)
    if ((run.flags & kRunSaveState) && prev_n4 > 0) {
        float *st = a.state_h + (size_t)(run.state_slot ^ 1) * a.state_slot_floats + (size_t)run.stream * nC() * half1;
        for (int i = lane; i < prev_n4; i += 64) {
            st[cA() * half1 + i] = tailL[i];
            st[cB() * half1 + i] = tailR[i];
        }
    }
    // HasClipped is sticky until ResetDecoder: the flag holds the stream's reset epoch
    if (__builtin_amdgcn_readlane(kv, 16) != 0 && __any(clip_peak > 0.99999994f) && lane == 0) atomicMax(&a.clipped[run.stream], run.clip_epoch);
#ifdef VPZ_WAVE_TIMES
    // diagnostic builds (-DVPZ_WAVE_TIMES): when each wave ran and where (HW_ID), two clock reads per wave
    if (a.stamps && lane == 0 && run_idx < (1 << 16)) {
        unsigned long long *rec = a.stamps + 16 + 16 * (size_t)run_idx;
        rec[0] = t_wave_begin;
        rec[1] = __builtin_amdgcn_s_memtime();
        rec[2] = wt_long; rec[3] = wt_short; rec[4] = wt_batch; rec[5] = wt_members;  // passes by kind, blocks in batches
        rec[6] = (unsigned long long)run.pre_kind;
        rec[9] = (unsigned long long)iters;
        rec[10] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID, all 32 bits
        atomicAdd(&a.stamps[15], 1ull);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------
#if VPZ_DUAL_PAIRS
// channel pairs of streams with 4, 6, 8, ... channels (the decoder has checked that the mappings' coupling steps join the channels
// two by two and built SynthArgs.pair_ch and the per-pair step tables)
bool synth_pairs_supported(int channels, int size0, int size1)
{
    auto plain = [](int n) { return n == 256 || n == 2048; };
    return channels >= 4 && channels <= 254 && (channels & 1) == 0 && plain(size0) && plain(size1);
}

hipError_t launch_synth_pairs(const SynthArgs &args, bool has_floor, bool interleaved_in, hipStream_t stream)
{
    if (args.n_runs <= 0) return hipSuccess;
    if (args.spec_i16 || args.n_pairs < 1 || 2 * args.n_pairs != args.channels || !args.pair_ch || args.n_mappings < 1)
        return hipErrorInvalidValue;  // (float32 residue, every channel in one pair)
    const int chunks = (args.n_runs + kDualWaves - 1) / kDualWaves;
    const int grid = (chunks + 7) / 8 * 8 * args.n_pairs;  // (groups of 8 chunks x n_pairs, see the kernel's first lines)
#else
bool synth_dual_supported(int channels, int size0, int size1)
{
    auto plain = [](int n) { return n == 256 || n == 2048; };
    return channels == 2 && plain(size0) && plain(size1);
}

hipError_t launch_synth_dual(const SynthArgs &args, bool has_floor, bool interleaved_in, hipStream_t stream)
{
    if (args.n_runs <= 0) return hipSuccess;
    if (args.spec_i16 && !has_floor) return hipErrorInvalidValue;  // (16-bit values are read in place by the floored variants only)
    const int grid = (args.n_runs + kDualWaves - 1) / kDualWaves;
#endif
    static const int extra_lds = getenv("VPZ_SYNTH_EXTRA_LDS") ? atoi(getenv("VPZ_SYNTH_EXTRA_LDS")) : 0;  // occupancy experiments
#define VPZ_LAUNCH_DUAL(F, I, O)                                                                                          \
    do {                                                                                                                  \
        if (!kPairs && F && args.spec_i16) {                                                                              \
            if (args.s16)                                                                                                 \
                hipLaunchKernelGGL((synth_dual_kernel<F, I, O, true, F && !kPairs>), dim3(grid), dim3(kDualThreads), extra_lds, stream, args); \
            else                                                                                                          \
                hipLaunchKernelGGL((synth_dual_kernel<F, I, O, false, F && !kPairs>), dim3(grid), dim3(kDualThreads), extra_lds, stream, args); \
        } else if (args.s16)                                                                                              \
            hipLaunchKernelGGL((synth_dual_kernel<F, I, O, true>), dim3(grid), dim3(kDualThreads), extra_lds, stream, args); \
        else                                                                                                              \
            hipLaunchKernelGGL((synth_dual_kernel<F, I, O, false>), dim3(grid), dim3(kDualThreads), extra_lds, stream, args); \
    } while (0)
#define VPZ_LAUNCH_DUAL_OUT(F, I)                \
    do {                                         \
        if (args.interleaved) VPZ_LAUNCH_DUAL(F, I, 1); \
        else VPZ_LAUNCH_DUAL(F, I, 0);           \
    } while (0)
    if (has_floor) {
        if (interleaved_in) VPZ_LAUNCH_DUAL_OUT(true, true);
        else VPZ_LAUNCH_DUAL_OUT(true, false);
    } else {
        if (interleaved_in) VPZ_LAUNCH_DUAL_OUT(false, true);
        else VPZ_LAUNCH_DUAL_OUT(false, false);
    }
#undef VPZ_LAUNCH_DUAL_OUT
#undef VPZ_LAUNCH_DUAL
    return hipGetLastError();
}

#if !VPZ_DUAL_PAIRS
int synth_dual_waves() { return kDualWaves; }

// channel-blocks the chip keeps resident and busy under synth_dual_kernel (a wavefront holds two)
int synth_dual_resident_slots(bool has_floor, int num_cu)
{
    int per_cu = 0;
    hipError_t e = has_floor ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, synth_dual_kernel<true, true, 1, false>, kDualThreads, 0)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, synth_dual_kernel<false, false, 0, false>, kDualThreads, 0);
    if (e != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 2;
    }
    return num_cu * per_cu * kDualWaves * 2;
}
#endif

}  // namespace vpz
