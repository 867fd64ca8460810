// Device helpers shared by the fused synthesis kernels (synth_kernels.hip: one wavefront per channel; synth_dual.hip: one
// wavefront per stereo stream): inverse coupling, the Floor1 curve render in LDS, clip / overlap-add / store epilogue
// helpers, the mirror addressing of the IMDCT output.  Everything is __forceinline__ device code.
#pragma once

#include <cstdint>
#include <type_traits>

#include "imdct_core.hpp"
#include "synth_desc.hpp"

// pointers that are GLOBAL by construction (the host pass of hipcc only parses the kernels: no address spaces there)
#if defined(__HIP_DEVICE_COMPILE__)
#define VPZ_GLOBAL __attribute__((address_space(1)))
#else
#define VPZ_GLOBAL
#endif

namespace vpz {

constexpr int kWaveBufFloats = 1160;  // h (<=1024 floats) | transposes (1152) | floor curve + 129 ints
constexpr int kWaveTailFloats = 512;  // upper half of the previous block's h: all a later block can overlap with

__device__ __forceinline__ int iabs(int x) { int s = x >> 31; return (x ^ s) - s; }

__device__ __forceinline__ void couple(float &m, float &a)
{
    // Mapping.cs:209-225, the Vector<T> form, lane-wise with the same bit operations
    const float oldM = m, oldA = a;
    const uint32_t posM = oldM > 0.0f ? 0xFFFFFFFFu : 0u;
    const uint32_t posA = oldA > 0.0f ? 0xFFFFFFFFu : 0u;
    const uint32_t signedA = __float_as_uint(oldA) ^ (0x80000000u & posM);
    m = oldM - __uint_as_float(signedA & ~posA);
    a = oldM + __uint_as_float(signedA & posA);
}

// ---------------------------------------------------------------------------------------------
// Floor1 curve render into LDS (curve[0..n) = inverse_dB_table[y(x)]), one wavefront.
// Closed form of the reference DDA (Floor1.cs:372-397): inside a segment (x0,y0)-(x1,y1) with
// adx = x1-x0, k = x-x0:  y = y0 + trunc(dy*k/adx);  the DDA error term after k steps is
// (|dy|*k mod adx) - adx.  Each lane starts four runs of 4 bins with one exact integer division
// and then steps the reference DDA.  Segment end uses min(hx, n) in the slope (quirk q2).
// aux layout (ints): [0..32) bitmap of active post x, [32..64) exclusive prefix popcounts,
// [64..129) compacted active posts, x in the low 16 bits, y (signed) in the high 16.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int div_floor_small(int a, int b)
{
    // exact floor(a / b) for 0 <= a < 2^31, 0 < b <= 32768 as long as the quotient stays below ~10^5 (it is at most
    // the y range here): the float quotient is off by less than 1 (relative error ~2^-22), the remainder -- in
    // integers -- says which way
    int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
    int r = a - q * b;
    if (r < 0) { --q; }
    else if (r >= b) { ++q; }
    return q;
}

// Floor1.RenderPoint (Floor1.cs:355-370)
__device__ __forceinline__ int render_point(int x0, int y0, int x1, int y1, int X)
{
    int dy = y1 - y0;
    int adx = x1 - x0;
    int ady = iabs(dy);
    int err = ady * (X - x0);
    int off = div_floor_small(err, adx);  // err >= 0, adx > 0: C#'s truncating division
    return dy < 0 ? y0 - off : y0 + off;
}

// One rendered segment between two active posts, packed for LDS:
//   A = x0 | x1 << 16          (x1 already clipped to n: quirk q2, Floor1.cs:248)
//   B = (y0 & 0xFFFF) | base << 16   (base = dy / adx, C# truncating division, signed)
//   C = adx | rseg << 13 | (dy < 0) << 31,  rseg = |dy| - |base| * adx
struct Seg {
    int x0, y0, x1, adx, sy, base, rseg;
};

__device__ __forceinline__ Seg unpack_segment(int A, int B, int C)
{
    Seg s;
    s.x0 = A & 0xFFFF;
    s.x1 = (A >> 16) & 0xFFFF;
    s.y0 = (int)(int16_t)(B & 0xFFFF);
    s.base = B >> 16;
    s.adx = C & 0x1FFF;
    s.rseg = (C >> 13) & 0x1FFF;
    s.sy = (C < 0) ? -1 : 1;
    return s;
}

// kWords = bitmap words (32 bins each) the block size needs: 32 for n <= 1024, 128 for n <= 4096.
// aux layout (ints, wave-private LDS): [0..kWords) bitmap of active post x, [kWords..2*kWords) exclusive prefix
// popcounts, then the packed segments A, B, C (65 entries each).
// cp: this lane's active post (lane < m), x | (finalY * multiplier) << 16, in X order -- what floor1_unwrap_kernel
// leaves per record.  Output: one byte per bin = index into the inverse dB table.  Every lane renders a contiguous
// run of bins: one exact division for its first bin (closed form of the DDA: y = y0 + trunc(dy*k/adx), error
// term (|dy|*k mod adx) - adx), then the reference's DDA (Floor1.cs:386-396) step by step.
constexpr int render_aux_ints(int words) { return 2 * words + 3 * 65; }

// n_render <= n: bins [n_render, n) are not needed (the spectrum is zero there: whatever index they get multiplies a
// zero) -- the lanes share the first n_render bins among themselves and the rest of the row is left as it is.
template <int kWords>
__device__ __forceinline__ void render_floor_indices(uint8_t *out, int *aux, int n, int n_render, int cp, int m, int lane)
{
    int *bitmap = aux, *prefix = aux + kWords;
    int *segA = aux + 2 * kWords, *segB = segA + 65, *segC = segB + 65;
    for (int w = lane; w < kWords; w += 64) bitmap[w] = 0;
    __builtin_amdgcn_wave_barrier();
    // one lane per segment between two active posts (Floor1.cs:238-252): slope parameters, with the only division a
    // segment needs; the next post comes from the neighbouring lane
    const int p1 = __shfl_down(cp, 1);
    if (lane < m) {
        const int x0 = cp & 0xFFFF, y0 = cp >> 16;
        if (x0 < n) atomicOr(reinterpret_cast<unsigned int *>(&bitmap[x0 >> 5]), 1u << (x0 & 31));
        int x1raw = n, y1 = y0;  // flat tail after the last active post (Floor1.cs:259-262)
        if (lane + 1 < m) { x1raw = p1 & 0xFFFF; y1 = p1 >> 16; }
        const int x1 = x1raw < n ? x1raw : n;  // Math.Min(hx, n) enters the slope: quirk q2 (Floor1.cs:248)
        const int adx = x1 - x0;
        const int dy = y1 - y0;
        const int ady = iabs(dy);
        const int ab = adx > 0 ? div_floor_small(ady, adx) : 0;
        const int base = dy < 0 ? -ab : ab;
        const int rseg = adx > 0 ? ady - ab * adx : 0;
        segA[lane] = x0 | (x1 << 16);
        segB[lane] = (y0 & 0xFFFF) | (base << 16);
        segC[lane] = (adx > 0 ? adx : 0) | (rseg << 13) | (dy < 0 ? (int)0x80000000 : 0);
    }
    __builtin_amdgcn_wave_barrier();
    // exclusive prefix popcount over the bitmap words
    if (kWords <= 32) {
        int c = (lane < kWords) ? __popc((unsigned)bitmap[lane]) : 0;
        int incl = c;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            int t = __shfl_up(incl, d);
            if ((lane & 31) >= d) incl += t;
        }
        if (lane < kWords) prefix[lane] = incl - c;
    } else {
        constexpr int per = kWords / 64;  // words per lane
        int c[per > 0 ? per : 1];
        int sum = 0;
#pragma unroll
        for (int i = 0; i < per; ++i) { c[i] = __popc((unsigned)bitmap[lane * per + i]); sum += c[i]; }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        int run = incl - sum;
#pragma unroll
        for (int i = 0; i < per; ++i) { prefix[lane * per + i] = run; run += c[i]; }
    }
    __builtin_amdgcn_wave_barrier();
    const int per_lane = n_render >= 256 ? (((n_render + 63) >> 6) + 3) & ~3 : 4;  // bins per lane, a multiple of 4
    int xx = lane * per_lane;
    if (xx >= n_render) return;
    const unsigned w = (unsigned)bitmap[xx >> 5];
    int j = prefix[xx >> 5] + __popc(w & ((2u << (xx & 31)) - 1u)) - 1;
    Seg s = unpack_segment(segA[j], segB[j], segC[j]);
    // the segment after this one is fetched ahead: a switch inside the bin loop then costs no LDS round trip
    // (segA/B/C hold 65 entries; entries past the last active post are never switched to)
    int nA = segA[j + 1], nB = segB[j + 1], nC = segC[j + 1];
    int yy, err;
    {
        const int k = xx - s.x0;
        const int ady = iabs(s.base) * s.adx + s.rseg;
        const int aa = ady * k;
        const int q = div_floor_small(aa, s.adx);
        yy = s.y0 + s.sy * q;
        err = (aa - q * s.adx) - s.adx;
    }
    uint32_t *out4 = reinterpret_cast<uint32_t *>(out + xx);
    for (int t4 = 0; t4 < per_lane; t4 += 4) {
        uint32_t packed = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int yi = yy < 0 ? 0 : (yy > 255 ? 255 : yy);  // the reference would index outside its table
            packed |= (uint32_t)yi << (8 * t);
            ++xx;
            if (xx == s.x1 && xx < n) {  // the next segment starts exactly on its post
                ++j;
                s = unpack_segment(nA, nB, nC);
                nA = segA[j + 1];
                nB = segB[j + 1];
                nC = segC[j + 1];
                yy = s.y0;
                err = -s.adx;
            } else {
                yy += s.base;
                err += s.rseg;
                if (err >= 0) { err -= s.adx; yy += s.sy; }
            }
        }
        out4[t4 >> 2] = packed;
    }
}

// The same curve without a serial walk, for blocks of up to 2048 samples (n <= 1024 bins): the DDA of Floor1.cs:386-396
// draws y(x) = y0 + sign(dy) * floor(|dy| * (x - x0) / adx), and that quotient is computed per bin in float32 --
//     trunc((dy * k +- 0.5) * rcp(adx)),  k = x - x0
// is exact as long as |dy| * adx <= 2^21: dy * k +- 0.5 is a float, the true value of (|dy| k + 0.5) / adx is at least
// 0.5 / adx away from an integer, and the two roundings (v_rcp_f32: 1 ulp, the product: 1/2 ulp) move it by less than
// |dy| k * 1.5 * 2^-23 / adx.  Valid streams have |dy| <= 255; a record that breaks the bound (only a corrupt one can)
// makes the function return false, nothing written, and the caller takes the integer walk above.
// Every lane owns the 4 bins of one output word per round (64 consecutive words per round: no bank conflicts), looks
// the segment of its FIRST bin up (bitmap of the post positions + prefix popcounts) and evaluates all four bins on that
// segment's line; the bins of a word that lie behind a post inside it are then rewritten by the lane that owns that
// post's segment (at most 3 bins each) -- no loop over segments, no divergence, 5 instructions per bin.
// aux (ints, wave-private LDS, 16-byte aligned): [0..32) bitmap, [32..64) prefix popcounts, then one float4 per
// segment: x0, y0, dy, 1 / adx.
constexpr int kRenderFastAuxInts = 64 + 4 * 65;

// inclusive prefix sum over lanes 0..31 (and, separately, 32..63) in the cross-lane data path of the vector ALU: no LDS
// round trips (a ds_bpermute chain is five dependent ones), no lane masks for the compiler to hoist and spill
__device__ __forceinline__ int wave_scan32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    return v;
}

__device__ __forceinline__ bool render_floor_indices_fast(uint8_t *out, int *aux, int n, int n_render, int cp, int m, int lane)
{
    int *bitmap = aux, *prefix = aux + 32;
    float4 *seg = reinterpret_cast<float4 *>(aux + 64);
    const int p1 = __shfl_down(cp, 1);
    const int x0 = cp & 0xFFFF, y0 = cp >> 16;
    int x1raw = n, y1 = y0;  // flat tail after the last active post (Floor1.cs:259-262)
    if (lane + 1 < m) { x1raw = p1 & 0xFFFF; y1 = p1 >> 16; }
    const int x1 = x1raw < n ? x1raw : n;  // Math.Min(hx, n) enters the slope: quirk q2 (Floor1.cs:248)
    const int adx = x1 - x0;
    const int dy = y1 - y0;
    const bool mine = lane < m && adx > 0;  // this lane's segment has bins
    if (__any(mine && iabs(dy) * adx > (1 << 21))) return false;
    const float x0f = (float)x0, y0f = (float)y0, dyf = (float)dy;
    const float rinv = __builtin_amdgcn_rcpf((float)(adx > 0 ? adx : 1));  // v_rcp_f32: 1 ulp, which the bound allows for
    if (lane < 32) bitmap[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (mine) {
        atomicOr(reinterpret_cast<unsigned int *>(&bitmap[x0 >> 5]), 1u << (x0 & 31));
        seg[lane] = make_float4(x0f, y0f, dyf, rinv);
    }
    __builtin_amdgcn_wave_barrier();
    {   // exclusive prefix popcount over the bitmap words (lanes 0..31; the upper half scans zeros)
        const int c = (lane < 32) ? __popc((unsigned)bitmap[lane]) : 0;
        const int incl = wave_scan32(c);
        if (lane < 32) prefix[lane] = incl - c;
    }
    __builtin_amdgcn_wave_barrier();
    // posts are counted, segments are stored by lane: a post at or beyond n has no bit and no bins, and only the
    // posts at the END of the X order can be there -- so "posts at or below x, minus one" is the segment's lane
    uint32_t *out4 = reinterpret_cast<uint32_t *>(out);
    const int rounds = (n_render + 255) >> 8;
    for (int r = 0; r < rounds; ++r) {
        const int w = lane + 64 * r, x = 4 * w;
        if (x < n_render) {
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
            out4[w] = pk;
        }
    }
    // a post inside a word: its bins up to the end of that word (or of its segment) belong to its own line
    const int xe = x1 < n_render ? x1 : n_render;
    const int a = x0 & 3;
    if (mine && a != 0 && x0 < xe) {
        const int cnt = min(4 - a, xe - x0);
        float t = copysignf(0.5f, dyf);
        out[x0] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(y0f + truncf(t * rinv), 0, 0u);
        t += dyf;
        if (cnt > 1) out[x0 + 1] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(y0f + truncf(t * rinv), 0, 0u);
        t += dyf;
        if (cnt > 2) out[x0 + 2] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(y0f + truncf(t * rinv), 0, 0u);
    }
    return true;
}

// Diagnostic builds (VPZ_EXTRA_HIPCC_FLAGS=-DVPZ_STAMPS): shader-clock stamps between the phases of a frame, summed per
// phase over all waves into a.stamps.  Compiled out otherwise.
#ifdef VPZ_STAMPS
#define VPZ_STAMP(k)                                                              \
    do {                                                                          \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();            \
        t_acc[k] += (unsigned)(t_now - t_last);                                   \
        t_last = t_now;                                                           \
    } while (0)
#else
#define VPZ_STAMP(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// synth_kernel
// ---------------------------------------------------------------------------------------------
// Utils.ClipValue (Utils.cs:44-58): strict comparisons, NaN passes through
__device__ __forceinline__ float clip_value(float v)
{
    return v > 0.99999994f ? 0.99999994f : (v < -0.99999994f ? -0.99999994f : v);
}
__device__ __forceinline__ bool was_clipped(float v) { return v > 0.99999994f || v < -0.99999994f; }
// HasClipped in the fused kernel: the lane keeps the largest magnitude it has emitted (one instruction per sample; a NaN
// leaves it alone, as it leaves Utils.ClipValue's comparisons false) and the test against the limit is made once per run
__device__ __forceinline__ float clip_track(float v, float &peak)
{
    peak = fmaxf(peak, fabsf(v));
    return clip_value(v);
}
// ... and a group of samples at once: Utils.ClipValue changes a sample only if its magnitude is above the limit, which in
// audio is the exception -- so the group's peak is taken (a maximum per sample; NaN leaves it alone, and NaN is what
// ClipValue passes through), and the compare-and-select per sample runs only where some lane of the wave saw a peak above
// the limit.  Same results, a fifth of the instructions.
__device__ __forceinline__ void clip_group(float &a, float &b, float &c, float &d, float &peak)
{
    const float m = fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d)));
    peak = fmaxf(peak, m);
    if (__any(m > 0.99999994f)) {
        a = clip_value(a);
        b = clip_value(b);
        c = clip_value(c);
        d = clip_value(d);
    }
}
// OverlapBuffers' `(v * v_lhs) + (v_prev * v_rhs)` (StreamDecoder.cs:788) with the reference's roundings --
// two products, one sum, never contracted into an FMA -- so that every emission path of the kernel
// (float4 / pair / scalar) gives the same bits for the same sample.
__device__ __forceinline__ float ola(float v, float wl, float t, float wr)
{
#pragma clang fp contract(off)
    const float a = v * wl;
    const float b = t * wr;
    return a + b;
}

// The reference's own tests turn PCM into 16-bit samples with `(int)(x * 32768f)` clamped to the short range
// (NVorbis.Tests/AssetTest.cs:131-132); the s16 output layouts do exactly that in the store epilogue, which halves the
// PCM write traffic.  v_cvt_i32_f32 truncates toward zero like the C# cast and saturates (NaN gives 0).
__device__ __forceinline__ int to_s16(float v)
{
    const int i = (int)(v * 32768.0f);
    return i < -32768 ? -32768 : (i > 32767 ? 32767 : i);
}
__device__ __forceinline__ uint32_t pack_s16(float lo, float hi)
{
    return ((uint32_t)to_s16(lo) & 0xFFFFu) | ((uint32_t)to_s16(hi) << 16);
}
// PCM leaves through pointers that are GLOBAL by construction.  The output base travels through scalar registers as two
// integers (see out_base), which costs the compiler its knowledge of the address space: it would emit FLAT stores --
// slower, and counted by the LDS wait counter as well.  Every PCM store goes through one of these.
typedef uint32_t vpz_u4v __attribute__((ext_vector_type(4)));
typedef uint32_t vpz_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_nt(uint2 *p, uint32_t a, uint32_t b)
{
    vpz_u2v t = {a, b};
    __builtin_nontemporal_store(t, (VPZ_GLOBAL vpz_u2v *)reinterpret_cast<vpz_u2v *>(p));
}
__device__ __forceinline__ void store_nt(uint4 *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    vpz_u4v t = {a, b, c, d};
    __builtin_nontemporal_store(t, (VPZ_GLOBAL vpz_u4v *)reinterpret_cast<vpz_u4v *>(p));
}
__device__ __forceinline__ void store_pcm4(float4 *p, float4 v)
{
    vpz_f4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (VPZ_GLOBAL vpz_f4v *)reinterpret_cast<vpz_f4v *>(p));
}
// Two adjacent 16-byte pieces per lane (interleaved stereo float32: L R L R | L R L R): two store instructions, each covering HALF of
// every 32-byte sector the wave writes.  As non-temporal stores the halves reach memory separately (rocprofv3 WRITE_SIZE: 510 MB
// for 428 MB of PCM on the configs[4] share, profiles/r4_pmc_write_layouts.txt); as plain stores they meet in the L2 first.
__device__ __forceinline__ void store_pcm4_pair(float4 *p, float4 v0, float4 v1)
{
    vpz_f4v t0 = {v0.x, v0.y, v0.z, v0.w}, t1 = {v1.x, v1.y, v1.z, v1.w};
#ifdef VPZ_PAIR_STORES_NT
    __builtin_nontemporal_store(t0, (VPZ_GLOBAL vpz_f4v *)reinterpret_cast<vpz_f4v *>(p));
    __builtin_nontemporal_store(t1, (VPZ_GLOBAL vpz_f4v *)reinterpret_cast<vpz_f4v *>(p) + 1);
#else
    *((VPZ_GLOBAL vpz_f4v *)reinterpret_cast<vpz_f4v *>(p)) = t0;
    *((VPZ_GLOBAL vpz_f4v *)reinterpret_cast<vpz_f4v *>(p) + 1) = t1;
#endif
}
template <class T>
__device__ __forceinline__ void store_pcm(T *p, T v) { *(VPZ_GLOBAL T *)p = v; }
// two adjacent float32 samples (a column pair of an interleaved PCM row: the pair route), one 8-byte global store, plain -- the
// rest of the row's lines comes from other workgroups, the L2 puts them together
typedef float vpz_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_pcm2(float *p, float a, float b)
{
    vpz_f2v t = {a, b};
    *((VPZ_GLOBAL vpz_f2v *)reinterpret_cast<vpz_f2v *>(p)) = t;
}

// Branch-free addressing of the IMDCT output through its mirror symmetries (Mdct.cs:378-381).
// y[pos..pos+3] (pos, n4 multiples of 4) = h4[idx] possibly reversed / negated.
struct Y4Map {
    int idx4;   // float4 index into h
    bool rev, neg;
};
__device__ __forceinline__ Y4Map map_y4(int pos, int n4)
{
    const bool a = pos < n4, c = pos >= 3 * n4;
    Y4Map m;
    m.idx4 = (a ? (n4 - 4 - pos) : (c ? (5 * n4 - 4 - pos) : (pos - n4))) >> 2;
    m.rev = a || c;
    m.neg = a;
    return m;
}
__device__ __forceinline__ float4 apply_y4(float4 t, bool rev, bool neg)
{
    float4 v = rev ? make_float4(t.w, t.z, t.y, t.x) : t;
    const uint32_t sgn = neg ? 0x80000000u : 0u;
    v.x = __uint_as_float(__float_as_uint(v.x) ^ sgn);
    v.y = __uint_as_float(__float_as_uint(v.y) ^ sgn);
    v.z = __uint_as_float(__float_as_uint(v.z) ^ sgn);
    v.w = __uint_as_float(__float_as_uint(v.w) ^ sgn);
    return v;
}
// previous block's output at position q (q in [N/2, N)) from the saved upper half of its h:
// tail[j] = h[n4 + j]
__device__ __forceinline__ float tail_at(const float *tail, int q, int pn4)
{
    return q < 3 * pn4 ? tail[q - 2 * pn4] : tail[4 * pn4 - 1 - q];
}

// the two floor-table indices (bytes) of bins 2k, 2k+1 for each of the lane's 8 points
// (`lpb`: lanes per block = block size / 32: 64 for 2048, 32 / 16 for 1024 / 512, 8 for 256; a block smaller than
// 2048 is transformed by every lane group of the wave at once, the first group's copy is the one used)
// (packed two points to a register: byte 0 / 1 = point 2j, byte 2 / 3 = point 2j + 1)
// All eight reads are issued before the first one is used: left to itself the compiler read them in pairs with an
// `s_waitcnt lgkmcnt(0)` behind each pair -- four LDS round trips in a row per channel (profiles/r4_isa_*.txt).
__device__ __forceinline__ void load_floor_indices(uint32_t (&fy)[4], const uint8_t *row, int lpb, int lane)
{
    const uint16_t *s = reinterpret_cast<const uint16_t *>(row);
    const int k0 = lane & (lpb - 1);
    const int st = lpb;
    uint32_t t[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) t[m] = s[k0 + st * m];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) fy[j] = t[2 * j] | (t[2 * j + 1] << 16);
}
// Floor1.Apply's multiply (Floor1.cs:383,395) on the lane's 8 points: the sixteen table values first (independent LDS
// reads, all in flight together), then the sixteen products
// upper (wave-uniform): false when the upper half of the block lies beyond the residue's support (ABI v4) -- its bins are
// +0.0 and stay +0.0 under any table entry: neither looked up nor multiplied
__device__ __forceinline__ void apply_floor(float2 (&x)[8], const uint32_t (&fy)[4], const float *s_db, bool upper = true)
{
    float t[16];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const uint32_t v = fy[m >> 1] >> (16 * (m & 1));
        t[2 * m] = s_db[v & 0xFFu];
        t[2 * m + 1] = s_db[(v >> 8) & 0xFFu];
    }
    if (upper) {
#pragma unroll
        for (int m = 4; m < 8; ++m) {
            const uint32_t v = fy[m >> 1] >> (16 * (m & 1));
            t[2 * m] = s_db[v & 0xFFu];
            t[2 * m + 1] = s_db[(v >> 8) & 0xFFu];
        }
    } else {
#pragma unroll
        for (int m = 8; m < 16; ++m) t[m] = 1.0f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        x[m].x *= t[2 * m];
        x[m].y *= t[2 * m + 1];
    }
}

// Highest point k = k0 + lpb * m (bins 2k, 2k + 1) of the wave's spectrum that holds a non-zero bin, -1 if there is none:
// bins [2 * (top + 1), n) are zero (the residue ends below N/2 in real streams) and need no curve.  lpb: lanes that hold
// distinct points (64 for a 2048 block, 8 for a 256 one: every lane group holds the same block).  Per lane the highest of its
// eight points with a bit set outside the sign (three vector instructions per point; -0.0 counts as zero, as `!= 0.0f` has
// it; a NaN counts as a value), then one maximum over the wave in the cross-lane data path -- where sixteen ballots, each
// examined by the scalar unit, took 250 instructions per stereo pass (profiles/r4_isa_*.txt).
__device__ __forceinline__ int spectrum_top(const float2 (&x)[8], int lpb, int lane)
{
    int t = -1;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const float u = __uint_as_float(__float_as_uint(x[m].x) | __float_as_uint(x[m].y));
        t = __builtin_amdgcn_classf(u, 0x39F) ? m : t;  // every class but +-0
    }
    int key = t < 0 ? -1 : t * lpb + (lane & (lpb - 1));
    const int lowest = (int)0x80000000;
    key = max(key, __builtin_amdgcn_update_dpp(lowest, key, 0x111, 0xF, 0xF, false));  // row_shr:1
    key = max(key, __builtin_amdgcn_update_dpp(lowest, key, 0x112, 0xF, 0xF, false));  // row_shr:2
    key = max(key, __builtin_amdgcn_update_dpp(lowest, key, 0x114, 0xF, 0xF, false));  // row_shr:4
    key = max(key, __builtin_amdgcn_update_dpp(lowest, key, 0x118, 0xF, 0xF, false));  // row_shr:8
    key = max(key, __builtin_amdgcn_update_dpp(lowest, key, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1 and 3
    key = max(key, __builtin_amdgcn_update_dpp(lowest, key, 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(key, 63);
}

// Builds h of one channel-block into the wave-private LDS buffer `hbuf`:
// (optional) Floor1 curve x spectrum, then the inverse MDCT.  kLong selects N = 2048 / 256.
template <bool kHasFloor, bool kLong>
__device__ __forceinline__ void build_block(uint32_t fd_flags, int lane, float2 (&x)[8], const uint32_t (&fy)[4],
                                            float *hbuf, const float2 *s_twL, const float2 *s_twAB,
                                            const float2 *s_twBC, const float2 *s_twS, const float *s_db)
{
    if (kHasFloor && !(fd_flags & kFrameNoFloor)) apply_floor(x, fy, s_db, ((fd_flags >> kFrameSkipShift) & kFrameSkipMask) < 4);
    if (kLong) {
        imdct2048_wave(x, reinterpret_cast<float2 *>(hbuf), s_twL, s_twAB, s_twBC, lane);
    } else {
        // all eight lane groups transform the same short block; group 0's copy lands at hbuf[0..128)
        imdct256_wave8(x, reinterpret_cast<float2 *>(hbuf), s_twS, s_twBC, lane);
    }
}

// this wave's channel of a planar packet: (X[2k], X[2k+1]) for the lane's 8 points (global memory or an LDS row)
// upper == false: the upper half of the row was not staged (ABI v4: beyond the residue's support) -- its points are +0.0
// (P: a pointer to float in whatever address space the caller knows it to be in -- an LDS row, or global memory)
template <class P>
__device__ __forceinline__ void load_spectrum(float2 (&x)[8], P base, int lpb, int lane, bool upper = true)
{
    using F2 = std::conditional_t<std::is_same<P, const VPZ_GLOBAL float *>::value, const VPZ_GLOBAL float2 *, const float2 *>;
    F2 s = reinterpret_cast<F2>(base);
    const int k0 = lane & (lpb - 1);
    const int st = lpb;
#pragma unroll
    for (int m = 0; m < 4; ++m) x[m] = s[k0 + st * m];
    if (upper) {
#pragma unroll
        for (int m = 4; m < 8; ++m) x[m] = s[k0 + st * m];
    } else {
#pragma unroll
        for (int m = 4; m < 8; ++m) x[m] = make_float2(0.0f, 0.0f);
    }
}

}  // namespace vpz
