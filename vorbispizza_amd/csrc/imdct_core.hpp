// Wavefront-level inverse MDCT cores for gfx950 (wave64).
//
// The reference computes the IMDCT with stb_vorbis' in-place radix-2 schedule (Mdct.cs:77-419).
// Here it is re-factored for a 64-lane wavefront (DESIGN.md "IMDCT kernel"):
//
//   tw[k] = exp(+2*pi*i*(k + 1/8)/N)                        k in [0, N/4)
//   z[k]  = (X[N/2-1-2k] + i*X[2k]) * tw[k]
//   Z     = unnormalised inverse DFT of z, length N/4        (512 points for N = 2048, 64 for 256)
//   W[j]  = Z[j] * tw[j]
//   h[2j] = Re W[j],  h[N/2-1-2j] = -Im W[j]                 h[m] = y[N/4 + m], m in [0, N/2)
//   y[n]  = -h[N/4-1-n] (n < N/4),  y[3N/4+m] = h[N/2-1-m] (m < N/4)     (Mdct.cs:378-381)
//
// N = 2048: one wavefront per channel-block, 8 complex points per lane, three radix-8 stages with
// two transposes through wave-private LDS.  N = 256: eight channel-blocks per wavefront (8 lanes x 8
// points each), two radix-8 stages, one transpose.  No MFMA: this is a butterfly network.
#pragma once

#include <hip/hip_runtime.h>

namespace vpz {

constexpr float kSqrtHalf = 0.70710678118654752440f;

// Complex arithmetic on (re, im) register pairs with the PACKED float32 instructions of gfx950 (two lanes of a 64-bit
// register pair per instruction, full rate): a complex add is one instruction, a complex multiply two -- the
// instructions' operand selectors (which half of a source feeds which half of the result) and sign modifiers do the
// swaps and negations a multiplication by i or a complex product needs, so none of them costs a move.  The compiler
// finds only part of this by itself (it packed 40 % of the transform), hence the explicit forms.
typedef float vpz_c2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vpz_c2 c2(float2 a) { return vpz_c2{a.x, a.y}; }
__device__ __forceinline__ float2 f2(vpz_c2 a) { return make_float2(a.x, a.y); }
// a + i*d = (a.x - d.y, a.y + d.x)
__device__ __forceinline__ vpz_c2 cadd_i(vpz_c2 a, vpz_c2 d)
{
    vpz_c2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
// a - i*d = (a.x + d.y, a.y - d.x)
__device__ __forceinline__ vpz_c2 csub_i(vpz_c2 a, vpz_c2 d)
{
    vpz_c2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
// a * b = a.x * (b.x, b.y) + a.y * (-b.y, b.x)
__device__ __forceinline__ vpz_c2 cmul2(vpz_c2 a, vpz_c2 b)
{
    vpz_c2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return f2(cmul2(c2(a), c2(b))); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by +i
__device__ __forceinline__ float2 cmul_i(float2 a) { return make_float2(-a.y, a.x); }

// value held by lane (63 - lane)
__device__ __forceinline__ float lane_mirror64(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute((63 - lane) << 2, __float_as_int(v)));
}
// Two values mirrored at once WITHOUT the LDS pipe (ds_bpermute goes through it; the stereo kernel keeps it ~70 % busy): lane bits 0..3
// by a row-mirroring DPP move, bit 4 by two v_permlane16_swap, bit 5 by two v_permlane32_swap (each swap pair hands both registers
// back with the rows / halves exchanged).  Pure data movement: the values of lane_mirror64.
__device__ __forceinline__ void lane_mirror64_x2(float a, float b, float &ma, float &mb)
{
    unsigned ua = __builtin_amdgcn_update_dpp(0u, __float_as_uint(a), 0x140, 0xF, 0xF, false);  // row_mirror
    unsigned ub = __builtin_amdgcn_update_dpp(0u, __float_as_uint(b), 0x140, 0xF, 0xF, false);
    auto r = __builtin_amdgcn_permlane16_swap(ua, ub, false, false);   // (A0 B0 A2 B2) (A1 B1 A3 B3)
    r = __builtin_amdgcn_permlane16_swap(r[1], r[0], false, false);    // (A1 A0 A3 A2) (B1 B0 B3 B2)
    r = __builtin_amdgcn_permlane32_swap(r[0], r[1], false, false);    // (Alo Blo) (Ahi Bhi)
    r = __builtin_amdgcn_permlane32_swap(r[1], r[0], false, false);    // (Ahi Alo) (Bhi Blo)
    ma = __uint_as_float(r[0]);
    mb = __uint_as_float(r[1]);
}
// value held by lane (lane ^ 7): mirror inside each group of 8 lanes
__device__ __forceinline__ float lane_mirror8(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 7) << 2, __float_as_int(v)));
}

typedef float vpz_f4v __attribute__((ext_vector_type(4)));
typedef float vpz_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_nt(float4 *p, float4 v)
{
    vpz_f4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vpz_f4v *>(p));
}
__device__ __forceinline__ void store_nt(float *p, float v) { __builtin_nontemporal_store(v, p); }

// In-register 8-point inverse DFT: v[p] <- sum_m v[m] * exp(+2*pi*i*p*m/8); 28 packed instructions
__device__ __forceinline__ void radix8_inverse(float2 (&vv)[8])
{
    vpz_c2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = c2(vv[i]);
    const vpz_c2 a0 = v[0] + v[4], a4 = v[0] - v[4];
    const vpz_c2 a1 = v[1] + v[5], d5 = v[1] - v[5];
    const vpz_c2 a2 = v[2] + v[6], a6 = v[2] - v[6];
    const vpz_c2 a3 = v[3] + v[7], d7 = v[3] - v[7];
    // odd branch twiddles exp(+i*pi*m/4), m = 1, 2, 3:
    //   d5 * (1 + i) / sqrt2 = (d5 + i*d5) / sqrt2;  d7 * (-1 + i) / sqrt2 = -(d7 - i*d7) / sqrt2;  a6 * i folded below
    const vpz_c2 a5 = cadd_i(d5, d5) * kSqrtHalf;
    const vpz_c2 a7 = csub_i(d7, d7) * -kSqrtHalf;

    const vpz_c2 b0 = a0 + a2, b2 = a0 - a2;
    const vpz_c2 b1 = a1 + a3, e3 = a1 - a3;
    const vpz_c2 c0 = cadd_i(a4, a6), c2_ = csub_i(a4, a6);
    const vpz_c2 c1 = a5 + a7, e7 = a5 - a7;

    vv[0] = f2(b0 + b1); vv[4] = f2(b0 - b1);
    vv[2] = f2(cadd_i(b2, e3)); vv[6] = f2(csub_i(b2, e3));
    vv[1] = f2(c0 + c1); vv[5] = f2(c0 - c1);
    vv[3] = f2(cadd_i(c2_, e7)); vv[7] = f2(csub_i(c2_, e7));
}

// Tuning builds only (-DVPZ_REG_TRANSPOSE): transpose 1 of dft512_wave without LDS -- register index bit b against lane
// bit 3 + b, b = 0, 1, 2: lane bit 5 and 4 with one `v_permlane32_swap` / `v_permlane16_swap` per register pair, lane bit 3
// with three row-rotating DPP moves.  Pure data movement: same bits.  Measured and not adopted, see DESIGN.md 4.7.
#ifdef VPZ_REG_TRANSPOSE
template <int kLaneBit>
__device__ __forceinline__ void lane_reg_exchange(float &a, float &b)
{
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    if (kLaneBit == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);  // a[32..63] <-> b[0..31]
        ua = r[0]; ub = r[1];
    } else if (kLaneBit == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(ua, ub, false, false);  // odd rows of a <-> even rows of b
        ua = r[0]; ub = r[1];
    } else {
        const unsigned t = __builtin_amdgcn_update_dpp(0u, ua, 0x128, 0xF, 0xF, false);  // row_ror:8: t[l] = a[l ^ 8]
        ua = __builtin_amdgcn_update_dpp(ua, ub, 0x128, 0xF, 0xC, false);                // lanes 8..15 of a row: b[l ^ 8]
        ub = __builtin_amdgcn_update_dpp(ub, t, 0xE4, 0xF, 0x3, false);                  // lanes 0..7 of a row: old a[l ^ 8]
    }
    a = __uint_as_float(ua); b = __uint_as_float(ub);
}
// (lane l0 + 8*l1, reg p) -> (lane l0 + 8*p, reg l1)
__device__ __forceinline__ void transpose_lane_hi(float2 (&z)[8])
{
#pragma unroll
    for (int p = 0; p < 8; ++p) if (!(p & 1)) { lane_reg_exchange<8>(z[p].x, z[p | 1].x); lane_reg_exchange<8>(z[p].y, z[p | 1].y); }
#pragma unroll
    for (int p = 0; p < 8; ++p) if (!(p & 2)) { lane_reg_exchange<16>(z[p].x, z[p | 2].x); lane_reg_exchange<16>(z[p].y, z[p | 2].y); }
#pragma unroll
    for (int p = 0; p < 8; ++p) if (!(p & 4)) { lane_reg_exchange<32>(z[p].x, z[p | 4].x); lane_reg_exchange<32>(z[p].y, z[p | 4].y); }
}
#endif

// LDS floats a wavefront needs for the transposes / the h staging area.
constexpr int kWaveScratchFloat2 = 576;  // 72*8 (stage A->B), 66*7+64 = 526 (stage B->C), 512 (h)

// -------------------------------------------------------------------------------------------
// 512-point unnormalised inverse DFT across a wavefront: z[m] holds point k = lane + 64*m on entry and
// Z[j], j = lane + 64*q, in z[q] on return.  Three radix-8 stages, two transposes through `scratch`
// (>= 576 float2, wave-private).
//   s_twAB : 512 float2  exp(2*pi*i*l*p/512) at [p*64 + l]
//   s_twBC :  64 float2  exp(2*pi*i*l0*q/64) at [l0*8 + q]
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void dft512_wave(float2 (&z)[8], float2 *scratch, const float2 *s_twAB,
                                            const float2 *s_twBC, int lane)
{
    // stage A: DFT over the top input digit (stride 64), output digit p
    radix8_inverse(z);
#pragma unroll
    for (int p = 1; p < 8; ++p) z[p] = cmul(z[p], s_twAB[p * 64 + lane]);
    // transpose 1: (lane l = l0 + 8*l1, reg p) -> (lane l0 + 8*p, reg l1); rows padded to 72
    const int l0 = lane & 7, pp = lane >> 3;
#ifdef VPZ_REG_TRANSPOSE
    transpose_lane_hi(z);
#else
#pragma unroll
    for (int p = 0; p < 8; ++p) scratch[72 * p + lane] = z[p];
#pragma unroll
    for (int l1 = 0; l1 < 8; ++l1) z[l1] = scratch[72 * pp + l0 + 8 * l1];
#endif
    // stage B: DFT over l1, output digit q1
    radix8_inverse(z);
#pragma unroll
    for (int q = 1; q < 8; ++q) z[q] = cmul(z[q], s_twBC[l0 * 8 + q]);
    // transpose 2: (lane l0 + 8*p, reg q1) -> (lane p + 8*q1, reg l0); row stride 66
#pragma unroll
    for (int q = 0; q < 8; ++q) scratch[66 * l0 + pp + 8 * q] = z[q];
#pragma unroll
    for (int r = 0; r < 8; ++r) z[r] = scratch[66 * r + lane];
    // stage C: DFT over l0, output digit q0
    radix8_inverse(z);
}

// -------------------------------------------------------------------------------------------
// N = 2048.  xa[m] = (X[2k], X[2k+1]) for k = lane + 64*m, already floor-multiplied if needed.
// On return the wave-private LDS area `h` holds h[0..1024) as floats in natural order.
//   s_tw   : 512 float2  tw[k]
//   s_twAB : 512 float2  exp(2*pi*i*l*p/512) at [p*64 + l]
//   s_twBC :  64 float2  exp(2*pi*i*l0*q/64) at [l0*8 + q]
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void imdct2048_wave(const float2 (&xa)[8], float2 *scratch,
                                               const float2 *s_tw, const float2 *s_twAB,
                                               const float2 *s_twBC, int lane)
{
    float2 z[8];
    float2 tw[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) tw[m] = s_tw[lane + 64 * m];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        // X[N/2-1-2k] sits in the .y half of the pair loaded by lane 63-lane for point 7-m
        float re = lane_mirror64(xa[7 - m].y, lane);
        z[m] = cmul(make_float2(re, xa[m].x), tw[m]);
    }
    dft512_wave(z, scratch, s_twAB, s_twBC, lane);  // lane now holds Z[j], j = lane + 64*q0
    float wim[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float2 w = cmul(z[q], tw[q]);
        z[q].x = w.x;
        wim[q] = -w.y;
    }
    // h[2j] = Re W[j];  h[2j+1] = -Im W[N/4-1-j], held by lane 63-lane in register 7-q0
    float *h = reinterpret_cast<float *>(scratch);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float hi = lane_mirror64(wim[7 - q], lane);
        reinterpret_cast<float2 *>(h)[lane + 64 * q] = make_float2(z[q].x, hi);
    }
}

// -------------------------------------------------------------------------------------------
// N = 256, eight channel-blocks per wavefront: group g = lane >> 3 owns one block, l = lane & 7.
// xa[m] = (X[2k], X[2k+1]) for k = l + 8*m of the group's block.
// On return h[g*128 .. g*128+128) holds the group's h in natural order.
//   s_tw   : 64 float2 tw[k] for N = 256
//   s_twBC : 64 float2 exp(2*pi*i*l*p/64) at [l*8 + p]   (shared with the 2048 path)
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void imdct256_wave8(const float2 (&xa)[8], float2 *scratch,
                                               const float2 *s_tw, const float2 *s_twBC, int lane)
{
    const int g = lane >> 3, l = lane & 7;
    float2 z[8];
    float2 tw[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) tw[m] = s_tw[l + 8 * m];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        float re = lane_mirror8(xa[7 - m].y, lane);
        z[m] = cmul(make_float2(re, xa[m].x), tw[m]);
    }
    radix8_inverse(z);  // over m (stride 8), output digit p
#pragma unroll
    for (int p = 1; p < 8; ++p) z[p] = cmul(z[p], s_twBC[l * 8 + p]);
    // transpose inside the group: (lane l, reg p) -> (lane p, reg l).  Row stride 9 (not 8) keeps both
    // the ds_write_b64 (16-lane groups) and the ds_read_b64 (32-lane groups) free of bank conflicts.
#pragma unroll
    for (int p = 0; p < 8; ++p) scratch[72 * g + 9 * p + l] = z[p];
#pragma unroll
    for (int r = 0; r < 8; ++r) z[r] = scratch[72 * g + 9 * l + r];
    radix8_inverse(z);  // over l, output digit q;  lane holds Z[j], j = l + 8*q
    float wim[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float2 w = cmul(z[q], tw[q]);
        z[q].x = w.x;
        wim[q] = -w.y;
    }
    float *h = reinterpret_cast<float *>(scratch);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float hi = lane_mirror8(wim[7 - q], lane);
        reinterpret_cast<float2 *>(h)[g * 64 + l + 8 * q] = make_float2(z[q].x, hi);
    }
}

// -------------------------------------------------------------------------------------------
// Two independent transforms by ONE wavefront, step by step side by side (synth_dual.hip: both channels of a stereo
// stream): every twiddle is read from LDS once for both, and each transform's LDS round trips (transposes, lane
// mirrors) are covered by the other's arithmetic -- the wave carries two dependency chains instead of one.  The
// operations per transform and their order are exactly those of the single versions above: same bits.
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void dft512_wave_x2(float2 (&za)[8], float2 (&zb)[8], float2 *sa, float2 *sb,
                                               const float2 *s_twAB, const float2 *s_twBC, int lane)
{
    radix8_inverse(za);
    radix8_inverse(zb);
#pragma unroll
    for (int p = 1; p < 8; ++p) {
        const float2 w = s_twAB[p * 64 + lane];
        za[p] = cmul(za[p], w);
        zb[p] = cmul(zb[p], w);
    }
    const int l0 = lane & 7, pp = lane >> 3;
#ifdef VPZ_REG_TRANSPOSE
    transpose_lane_hi(za);
    transpose_lane_hi(zb);
#else
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        sa[72 * p + lane] = za[p];
        sb[72 * p + lane] = zb[p];
    }
#pragma unroll
    for (int l1 = 0; l1 < 8; ++l1) {
        za[l1] = sa[72 * pp + l0 + 8 * l1];
        zb[l1] = sb[72 * pp + l0 + 8 * l1];
    }
#endif
    radix8_inverse(za);
    radix8_inverse(zb);
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        const float2 w = s_twBC[l0 * 8 + q];
        za[q] = cmul(za[q], w);
        zb[q] = cmul(zb[q], w);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        sa[66 * l0 + pp + 8 * q] = za[q];
        sb[66 * l0 + pp + 8 * q] = zb[q];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        za[r] = sa[66 * r + lane];
        zb[r] = sb[66 * r + lane];
    }
    radix8_inverse(za);
    radix8_inverse(zb);
}

// imdct2048_wave for two spectra (xa -> scratch_a, xb -> scratch_b)
__device__ __forceinline__ void imdct2048_wave_x2(const float2 (&xa)[8], const float2 (&xb)[8], float2 *scratch_a,
                                                  float2 *scratch_b, const float2 *s_tw, const float2 *s_twAB,
                                                  const float2 *s_twBC, int lane)
{
    float2 za[8], zb[8];
    float2 tw[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) tw[m] = s_tw[lane + 64 * m];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const float ra = lane_mirror64(xa[7 - m].y, lane), rb = lane_mirror64(xb[7 - m].y, lane);
        za[m] = cmul(make_float2(ra, xa[m].x), tw[m]);
        zb[m] = cmul(make_float2(rb, xb[m].x), tw[m]);
    }
    dft512_wave_x2(za, zb, scratch_a, scratch_b, s_twAB, s_twBC, lane);
    float wa[8], wb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float2 a = cmul(za[q], tw[q]), b = cmul(zb[q], tw[q]);
        za[q].x = a.x;
        wa[q] = -a.y;
        zb[q].x = b.x;
        wb[q] = -b.y;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float ha = lane_mirror64(wa[7 - q], lane), hb = lane_mirror64(wb[7 - q], lane);
        scratch_a[lane + 64 * q] = make_float2(za[q].x, ha);
        scratch_b[lane + 64 * q] = make_float2(zb[q].x, hb);
    }
}

// ... with the three twiddle sets in REGISTERS (a lane reads the same 22 entries for every 2048 block: tw[m] = s_tw[lane + 64 m],
// twAB[p] = s_twAB[64 p + lane], twBC[q] = s_twBC[8 (lane & 7) + q]; entries 0 of the last two are not used): the stereo kernel keeps
// them across the frames of a run -- 11 KB less through the CU's LDS pipe per pass.  Operations and their order: imdct2048_wave_x2's.
// kBCRegs false: the third set stays in LDS (s_twBC) -- for the variants that need its 14 registers for something worth more.
template <bool kBCRegs>
__device__ __forceinline__ void imdct2048_wave_x2_regs(const float2 (&xa)[8], const float2 (&xb)[8], float2 *scratch_a, float2 *scratch_b,
                                                       const float2 (&tw)[8], const float2 (&twAB)[8], const float2 (&twBC)[8],
                                                       const float2 *s_twBC, int lane)
{
    float2 za[8], zb[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
#ifndef VPZ_DUAL_MIRROR_VALU  // (-DVPZ_DUAL_MIRROR_VALU: lane_mirror64_x2, measured equal -- profiles/r5_ab_lds_diet.txt)
        const float ra = lane_mirror64(xa[7 - m].y, lane), rb = lane_mirror64(xb[7 - m].y, lane);
#else
        float ra, rb;
        lane_mirror64_x2(xa[7 - m].y, xb[7 - m].y, ra, rb);
#endif
        za[m] = cmul(make_float2(ra, xa[m].x), tw[m]);
        zb[m] = cmul(make_float2(rb, xb[m].x), tw[m]);
    }
    {   // dft512_wave_x2
        float2 *sa = scratch_a, *sb = scratch_b;
        radix8_inverse(za);
        radix8_inverse(zb);
#pragma unroll
        for (int p = 1; p < 8; ++p) {
            za[p] = cmul(za[p], twAB[p]);
            zb[p] = cmul(zb[p], twAB[p]);
        }
        const int l0 = lane & 7, pp = lane >> 3;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            sa[72 * p + lane] = za[p];
            sb[72 * p + lane] = zb[p];
        }
#pragma unroll
        for (int l1 = 0; l1 < 8; ++l1) {
            za[l1] = sa[72 * pp + l0 + 8 * l1];
            zb[l1] = sb[72 * pp + l0 + 8 * l1];
        }
        radix8_inverse(za);
        radix8_inverse(zb);
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            const float2 w = kBCRegs ? twBC[q] : s_twBC[l0 * 8 + q];
            za[q] = cmul(za[q], w);
            zb[q] = cmul(zb[q], w);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sa[66 * l0 + pp + 8 * q] = za[q];
            sb[66 * l0 + pp + 8 * q] = zb[q];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            za[r] = sa[66 * r + lane];
            zb[r] = sb[66 * r + lane];
        }
        radix8_inverse(za);
        radix8_inverse(zb);
    }
    float wa[8], wb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float2 a = cmul(za[q], tw[q]), b = cmul(zb[q], tw[q]);
        za[q].x = a.x;
        wa[q] = -a.y;
        zb[q].x = b.x;
        wb[q] = -b.y;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#ifndef VPZ_DUAL_MIRROR_VALU  // (-DVPZ_DUAL_MIRROR_VALU: lane_mirror64_x2, measured equal -- profiles/r5_ab_lds_diet.txt)
        const float ha = lane_mirror64(wa[7 - q], lane), hb = lane_mirror64(wb[7 - q], lane);
#else
        float ha, hb;
        lane_mirror64_x2(wa[7 - q], wb[7 - q], ha, hb);
#endif
        scratch_a[lane + 64 * q] = make_float2(za[q].x, ha);
        scratch_b[lane + 64 * q] = make_float2(zb[q].x, hb);
    }
}

// imdct256_wave8 for two sets of eight blocks (lane group g transforms block g of each set)
__device__ __forceinline__ void imdct256_wave8_x2(const float2 (&xa)[8], const float2 (&xb)[8], float2 *scratch_a,
                                                  float2 *scratch_b, const float2 *s_tw, const float2 *s_twBC, int lane)
{
    const int g = lane >> 3, l = lane & 7;
    float2 za[8], zb[8];
    float2 tw[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) tw[m] = s_tw[l + 8 * m];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const float ra = lane_mirror8(xa[7 - m].y, lane), rb = lane_mirror8(xb[7 - m].y, lane);
        za[m] = cmul(make_float2(ra, xa[m].x), tw[m]);
        zb[m] = cmul(make_float2(rb, xb[m].x), tw[m]);
    }
    radix8_inverse(za);
    radix8_inverse(zb);
#pragma unroll
    for (int p = 1; p < 8; ++p) {
        const float2 w = s_twBC[l * 8 + p];
        za[p] = cmul(za[p], w);
        zb[p] = cmul(zb[p], w);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        scratch_a[72 * g + 9 * p + l] = za[p];
        scratch_b[72 * g + 9 * p + l] = zb[p];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        za[r] = scratch_a[72 * g + 9 * l + r];
        zb[r] = scratch_b[72 * g + 9 * l + r];
    }
    radix8_inverse(za);
    radix8_inverse(zb);
    float wa[8], wb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float2 a = cmul(za[q], tw[q]), b = cmul(zb[q], tw[q]);
        za[q].x = a.x;
        wa[q] = -a.y;
        zb[q].x = b.x;
        wb[q] = -b.y;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float ha = lane_mirror8(wa[7 - q], lane), hb = lane_mirror8(wb[7 - q], lane);
        scratch_a[g * 64 + l + 8 * q] = make_float2(za[q].x, ha);
        scratch_b[g * 64 + l + 8 * q] = make_float2(zb[q].x, hb);
    }
}

// -------------------------------------------------------------------------------------------
// N = 4096: the 1024-point transform as two 512-point ones (even / odd input points) and one radix-2 step.
// xa[m] = X[4k' .. 4k'+3] for k' = lane + 64*m.  On return `h` (>= 1024 float2, wave-private; its first 576
// float2 double as the transposes' scratch) holds h[0..2048) in natural order.
//   s_tw   : 1024 float2  tw[k] = exp(2*pi*i*(k + 1/8)/4096)
//   s_twAB / s_twBC : the 512-point tables of dft512_wave
//   s_w    :  512 float2  exp(2*pi*i*j/1024)
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void imdct4096_wave(const float4 (&xa)[8], float2 *h, const float2 *s_tw,
                                               const float2 *s_twAB, const float2 *s_twBC, const float2 *s_w, int lane)
{
    float2 e[8], o[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        // point k = 2k' (even) and 2k'+1 (odd): z[k] = (X[N/2-1-2k] + i*X[2k]) * tw[k]; the mirrored X values sit
        // in the .w / .y components of the float4 loaded by lane 63-lane for its point 7-m
        const int kp = lane + 64 * m;
        const float re_e = lane_mirror64(xa[7 - m].w, lane), re_o = lane_mirror64(xa[7 - m].y, lane);
        e[m] = cmul(make_float2(re_e, xa[m].x), s_tw[2 * kp]);
        o[m] = cmul(make_float2(re_o, xa[m].z), s_tw[2 * kp + 1]);
    }
    dft512_wave(e, h, s_twAB, s_twBC, lane);
    dft512_wave(o, h, s_twAB, s_twBC, lane);
    // Z[j] = E[j] + w^j O[j], Z[j + 512] = E[j] - w^j O[j];  then W = Z * tw
    float2 wl[8], wu[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = lane + 64 * q;
        const float2 t = cmul(o[q], s_w[j]);
        wl[q] = cmul(cadd(e[q], t), s_tw[j]);
        wu[q] = cmul(csub(e[q], t), s_tw[j + 512]);
    }
    // h[2J] = Re W[J], h[2J+1] = -Im W[1023-J]: for J = j the partner is the UPPER value of lane 63-lane,
    // register 7-q (1023 - j = 512 + (511 - j)); for J = j + 512 it is that lane's LOWER value
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float from_upper = lane_mirror64(-wu[7 - q].y, lane);
        const float from_lower = lane_mirror64(-wl[7 - q].y, lane);
        const int j = lane + 64 * q;
        h[j] = make_float2(wl[q].x, from_upper);
        h[j + 512] = make_float2(wu[q].x, from_lower);
    }
}

// -------------------------------------------------------------------------------------------
// N = 8192: the 2048-point transform as four 512-point ones (input points k = 4k' + r) and two radix-2 levels.
// lo[m] = X[8k' .. 8k'+3], hi[m] = X[8k'+4 .. 8k'+7] for k' = lane + 64*m.  On return `h` (>= 2048 float2,
// wave-private; its first 576 float2 double as the transposes' scratch) holds h[0..4096) in natural order.
//   g_tw  : 2048 float2 tw[k] = exp(2*pi*i*(k + 1/8)/8192)   (global memory: read once per block and lane)
//   g_w2  : 1024 float2 exp(2*pi*i*J/2048), of which the lower 512 are read (global memory, or LDS)
//   s_w1  :  512 float2 exp(2*pi*i*j/1024), s_twAB / s_twBC: the 512-point tables (LDS)
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void imdct8192_wave(const float4 (&lo)[8], const float4 (&hi)[8], float2 *h,
                                               const float2 *g_tw, const float2 *g_w2, const float2 *s_w1,
                                               const float2 *s_twAB, const float2 *s_twBC, int lane)
{
    float2 a0[8], a1[8], a2[8], a3[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        // z[k] = (X[N/2-1-2k] + i*X[2k]) * tw[k], k = 4k' + r: X[2k] is component 2r of this lane's group of 8,
        // X[N/2-1-2k] component 7-2r of the group lane 63-lane loaded for its point 7-m
        const int k = 4 * (lane + 64 * m);
        const float r0 = lane_mirror64(hi[7 - m].w, lane), r1 = lane_mirror64(hi[7 - m].y, lane);
        const float r2 = lane_mirror64(lo[7 - m].w, lane), r3 = lane_mirror64(lo[7 - m].y, lane);
        a0[m] = cmul(make_float2(r0, lo[m].x), g_tw[k]);
        a1[m] = cmul(make_float2(r1, lo[m].z), g_tw[k + 1]);
        a2[m] = cmul(make_float2(r2, hi[m].x), g_tw[k + 2]);
        a3[m] = cmul(make_float2(r3, hi[m].z), g_tw[k + 3]);
    }
    dft512_wave(a0, h, s_twAB, s_twBC, lane);
    dft512_wave(a1, h, s_twAB, s_twBC, lane);
    dft512_wave(a2, h, s_twAB, s_twBC, lane);
    dft512_wave(a3, h, s_twAB, s_twBC, lane);
    // level 1: E = DFT1024 of the even points (a0 | a2), O of the odd ones (a1 | a3);
    // level 2: Z[J] = E[J] + w2^J O[J], Z[J + 1024] = E[J] - w2^J O[J]
    float2 z0[8], z1[8], z2[8], z3[8];  // Z[j], Z[j + 512], Z[j + 1024], Z[j + 1536]
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = lane + 64 * q;
        const float2 te = cmul(a2[q], s_w1[j]), to = cmul(a3[q], s_w1[j]);
        const float2 el = cadd(a0[q], te), eu = csub(a0[q], te);
        const float2 ol = cadd(a1[q], to), ou = csub(a1[q], to);
        // (w2[J + 512] = i w2[J]: the table's upper half IS the lower one turned -- vpz_context.hip builds it so --, one read serves both)
        const float2 w2l = g_w2[j], w2u = make_float2(-w2l.y, w2l.x);
        const float2 tl = cmul(ol, w2l), tu = cmul(ou, w2u);
        z0[q] = cmul(cadd(el, tl), g_tw[j]);
        z1[q] = cmul(cadd(eu, tu), g_tw[j + 512]);
        z2[q] = cmul(csub(el, tl), g_tw[j + 1024]);
        z3[q] = cmul(csub(eu, tu), g_tw[j + 1536]);
    }
    // h[2J] = Re W[J], h[2J+1] = -Im W[2047-J]: the partner of quarter s is quarter 3-s of lane 63-lane, register 7-q
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = lane + 64 * q;
        h[j] = make_float2(z0[q].x, lane_mirror64(-z3[7 - q].y, lane));
        h[j + 512] = make_float2(z1[q].x, lane_mirror64(-z2[7 - q].y, lane));
        h[j + 1024] = make_float2(z2[q].x, lane_mirror64(-z1[7 - q].y, lane));
        h[j + 1536] = make_float2(z3[q].x, lane_mirror64(-z0[7 - q].y, lane));
    }
}

// value held by lane (lane ^ MASK)
template <int MASK>
__device__ __forceinline__ float lane_xor(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ MASK) << 2, __float_as_int(v)));
}
template <int MASK>
__device__ __forceinline__ float2 lane_xor2(float2 v, int lane)
{
    return make_float2(lane_xor<MASK>(v.x, lane), lane_xor<MASK>(v.y, lane));
}

// -------------------------------------------------------------------------------------------
// N = 512 (R = 2) and N = 1024 (R = 4): the N/4 = 64*R point transform over L = 8*R lanes, 8 points per
// lane, so a wavefront holds 8/R channel-blocks (block b = lane / L, l = lane % L = l0 + 8*l1).
//   input  k = l0 + 8*l1 + L*m        (m = register)
//   output j = p + 8*q1 + L*q0        p: 8-point DFT over m (registers), q1: R-point DFT over l1 (lane
//                                     exchanges, no LDS), q0: 8-point DFT over l0 (one LDS transpose)
// The R-point stage leaves its output digit bit-reversed in the lane index: after it lane l0 + 8*l1 holds
// q1 = rev(l1); s_twBC and the final indices account for that.
//   s_tw   : 64*R float2  tw[k] = exp(2*pi*i*(k + 1/8)/N)
//   s_twAB : 64*R float2  exp(2*pi*i*p*l/(64*R)) at [p*L + l]
//   s_twBC :  8*R float2  exp(2*pi*i*rev(l1)*l0/(8*R)) at [l]
// On return h[b*(N/2) .. (b+1)*(N/2)) holds block b's h in natural order.
// -------------------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void imdct_mid_wave(const float2 (&xa)[8], float2 *scratch, const float2 *s_tw,
                                               const float2 *s_twAB, const float2 *s_twBC, int lane)
{
    static_assert(R == 2 || R == 4, "N = 512 or 1024");
    constexpr int L = 8 * R, M = 64 * R;
    const int l = lane & (L - 1), b = lane / L, l1 = l >> 3;
    float2 z[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        // X[N/2-1-2k] sits in the .y half of the pair loaded by lane l ^ (L-1) of the block for point 7-m
        const float re = lane_xor<L - 1>(xa[7 - m].y, lane);
        z[m] = cmul(make_float2(re, xa[m].x), s_tw[l + L * m]);
    }
    radix8_inverse(z);  // over m, output digit p
#pragma unroll
    for (int p = 1; p < 8; ++p) z[p] = cmul(z[p], s_twAB[p * L + l]);
    // R-point inverse DFT over l1, across lanes
    if (R == 2) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float2 o = lane_xor2<8>(z[p], lane);
            z[p] = l1 == 0 ? cadd(z[p], o) : csub(o, z[p]);
        }
    } else {
        const bool hi = l1 & 2, lo = l1 & 1;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            // a[l1] -> e_t = a_t + a_{t+2} (lanes l1 = t), o_t = a_t - a_{t+2} (lanes l1 = t + 2)
            const float2 o = lane_xor2<16>(z[p], lane);
            const float2 s1 = hi ? csub(o, z[p]) : cadd(z[p], o);
            // even outputs from the e lanes, odd ones from the o lanes: lane l1 ends up with q1 = rev(l1)
            const float2 t = lane_xor2<8>(s1, lane);
            if (!hi) z[p] = lo ? csub(t, s1) : cadd(s1, t);                 // b[2] = e0 - e1 ; b[0] = e0 + e1
            else z[p] = lo ? csub(t, cmul_i(s1)) : cadd(s1, cmul_i(t));     // b[3] = o0 - i*o1 ; b[1] = o0 + i*o1
        }
    }
    {
        const float2 w = s_twBC[l];
#pragma unroll
        for (int p = 0; p < 8; ++p) z[p] = cmul(z[p], w);
    }
    // transpose inside each group of 8 lanes: (lane l0, reg p) -> (lane p, reg l0); row stride 9 as in the
    // N = 256 version
    const int g = lane >> 3, l0 = lane & 7;
#pragma unroll
    for (int p = 0; p < 8; ++p) scratch[72 * g + 9 * p + l0] = z[p];
#pragma unroll
    for (int r = 0; r < 8; ++r) z[r] = scratch[72 * g + 9 * l0 + r];
    radix8_inverse(z);  // over l0, output digit q0;  lane holds Z[j], j = jl + L*q0
    const int q1 = R == 2 ? l1 : (((l1 & 1) << 1) | (l1 >> 1));
    const int jl = l0 + 8 * q1;
    float wim[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float2 w = cmul(z[q], s_tw[jl + L * q]);
        z[q].x = w.x;
        wim[q] = -w.y;
    }
    // h[2j] = Re W[j];  h[2j+1] = -Im W[M-1-j]: digits (7-p, R-1-q1, 7-q0) = lane ^ (L-1), register 7-q0
    float2 *h2 = scratch + b * M;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float hi2 = lane_xor<L - 1>(wim[7 - q], lane);
        h2[jl + L * q] = make_float2(z[q].x, hi2);
    }
}

// Value of the full IMDCT output y[pos] given h (natural order, N/2 floats), N = 4*n4.
__device__ __forceinline__ float y_from_h(const float *h, int pos, int n4)
{
    if (pos < n4) return -h[n4 - 1 - pos];
    if (pos < 3 * n4) return h[pos - n4];
    return h[5 * n4 - 1 - pos];
}

}  // namespace vpz
