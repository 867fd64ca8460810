// Floor0.Apply (Floor0.cs:164-225) -- the LSP floor the reference itself calls "virtually unused".
// One workgroup per (packet, channel) record that has a type-0 floor: the spectrum in the planar temp is
// multiplied in place; floor1_unwrap_kernel gives the record the one-post curve "index 255" (table[255] == 1.0),
// so the synthesis kernels downstream need no floor-0 variant.  Compiled with -ffp-contract=off.  The LSP
// product is ill conditioned next to a root (w - c_j cancels), where a 1-ulp difference between two cosf
// implementations shows up as 1e-4 of the output; cos / sqrt / exp / divide are therefore evaluated in
// double and rounded once, which reproduces a correctly rounded host libm (the reference's MathF).
#include "floor0_math.hpp"
#include "synth_desc.hpp"
#include "vpz_internal.hpp"

namespace vpz {

struct Floor0Dev {
    int32_t order, bark_map_size, amp_ofs, reserved;
    int64_t bark_off[2];  // offset of the short / long bark map (n+1 ints each) in the maps buffer
};

struct Floor0Rec {        // one record to process
    int64_t spec_off;     // float offset of this channel's spectrum in the temp
    int32_t rec;          // channel record (index into amp / coeff rows)
    int32_t floor;        // index into Floor0Dev table
    int32_t half;         // blocksize / 2
    int32_t is_long;
};

__global__ __launch_bounds__(256) void floor0_apply_kernel(const Floor0Rec *__restrict__ recs,
                                                          const Floor0Dev *__restrict__ floors,
                                                          const int32_t *__restrict__ bark_maps,
                                                          const float *__restrict__ amp,
                                                          const float *__restrict__ coeff, int coeff_stride,
                                                          float *__restrict__ spec)
{
    __shared__ float s_c[256];
    const Floor0Rec r = recs[blockIdx.x];
    const Floor0Dev f = floors[r.floor];
    float *x = spec + r.spec_off;
    const float a = amp[r.rec];
    if (a <= 0.0f) {  // :169-173
        for (int i = threadIdx.x; i < r.half; i += 256) x[i] = 0.0f;
        return;
    }
    for (int j = threadIdx.x; j < f.order; j += 256)
        s_c[j] = 2.0f * cos_rounded_once(coeff[(size_t)r.rec * coeff_stride + j]);
    __syncthreads();
    const int32_t *bark = bark_maps + f.bark_off[r.is_long];
    const float wdel = (float)(3.14159265358979323846 / (double)f.bark_map_size);
    const float amp_ofs = (float)f.amp_ofs;
    for (int i = threadIdx.x; i < r.half; i += 256) {
        const int k = bark[i];
        float p = .5f, q = .5f;
        const float w = 2.0f * cos_rounded_once(wdel * (float)k);
        int j;
        for (j = 1; j < f.order; j += 2) {
            q *= w - s_c[j - 1];
            p *= w - s_c[j];
        }
        if (j == f.order) {  // odd order
            q *= w - s_c[j - 1];
            p *= p * (4.0f - w * w);
            q *= q;
        } else {
            p *= p * (2.0f - w);
            q *= q * (2.0f + w);
        }
        // (amp / sqrt(p + q) in float32, both correctly rounded -- hipcc's default for float division and square root --, which is
        // what rounding the double-precision quotient of the double-precision root gives: 53 bits are more than 2 * 24 + 2)
        q = a / sqrtf(p + q) - amp_ofs;
        q = exp_rounded_once(q * 0.11512925f);
        x[i] *= q;
    }
}

// The fused route (stereo fast path): Floor0's multiplier is a function of the bin's BARK INDEX only -- the reference computes
// q for a run of bins with the same barkMap value and multiplies them all by it (Floor0.cs:188-222: `while (barkMap[++i] == k)
// residue[i] *= q`) -- so a record's curve is bark_map_size values, not blocksize/2: this kernel evaluates them once per
// (packet, channel) record into curve[rec][k] (the same arithmetic, per k instead of per bin), and synth_dual_kernel looks its
// bins' values up by the bark map (floor0_multiply in synth_dual.hip).  No planar temp, no extra pass over the spectrum.
// rec_info[rec]: floor index | 0x40 for a type-0 floor (else the record is skipped) | 0x80 long block.
// wtab[floor * k_stride + k] = 2 cos(pi k / bark_map_size) (Floor0.cs:103-111 `SynthesizeWDelMap`): a setup-time table, filled
// once per decoder with the expression floor0_apply_kernel evaluates per bin
__global__ __launch_bounds__(256) void floor0_wtab_kernel(const Floor0Dev *__restrict__ floors, int k_stride, float *__restrict__ wtab)
{
    const Floor0Dev f = floors[blockIdx.x];
    if (f.bark_map_size <= 0) return;
    const float wdel = (float)(3.14159265358979323846 / (double)f.bark_map_size);
    for (int k = threadIdx.x; k < f.bark_map_size && k < k_stride; k += 256)
        wtab[(size_t)blockIdx.x * k_stride + k] = 2.0f * cos_rounded_once(wdel * (float)k);
}

hipError_t launch_floor0_wtab(const void *floors, int n_floors, int k_stride, float *wtab, hipStream_t stream)
{
    if (n_floors <= 0) return hipSuccess;
    hipLaunchKernelGGL(floor0_wtab_kernel, dim3(n_floors), dim3(256), 0, stream, static_cast<const Floor0Dev *>(floors), k_stride, wtab);
    return hipGetLastError();
}

// A wavefront per record, four wavefronts per workgroup, and a grid that stays: every wavefront takes records wave, wave + W,
// wave + 2 W, ... with the next record's header, amplitude and coefficients in flight while the current one is evaluated.  (A
// workgroup per record -- 256 threads for 256 values -- ran at the dispatcher's workgroup rate: 65 536 of them took 175 us; a
// wavefront per record in 16 384 workgroups 87 us, still mostly launch and the chain of dependent loads at its head.)
constexpr int kCurveWaves = 4;
constexpr int kCurveGroupsMax = 256 * 8;
__global__ __launch_bounds__(64 * kCurveWaves) void floor0_curve_kernel(int n_rec, const uint8_t *__restrict__ rec_info,
                                                                       const Floor0Dev *__restrict__ floors, const float *__restrict__ amp,
                                                                       const float *__restrict__ coeff, int coeff_stride, int k_stride,
                                                                       const float *__restrict__ wtab, float *__restrict__ curve,
                                                                       const uint8_t *__restrict__ post_counts, uint8_t *__restrict__ ccount,
                                                                       int32_t *__restrict__ cposts)
{
    __shared__ __attribute__((aligned(16))) float s_c_all[kCurveWaves][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *s_c = s_c_all[wave];
    const float4 *s_c4 = reinterpret_cast<const float4 *>(s_c);
    const int n_waves = gridDim.x * kCurveWaves;
    int rec = blockIdx.x * kCurveWaves + wave;
    if (rec >= n_rec) return;
    const int pre = lane < coeff_stride ? lane : 0;  // (the first 64 coefficients are prefetched; an order beyond that reads the rest late)
    int info_n = rec_info[rec], cnt_n = post_counts[rec];
    float a_n = amp[rec], c_n = coeff[(size_t)rec * coeff_stride + pre];
    int cur_floor = -1;
    Floor0Dev f = {};
    for (; rec < n_rec; rec += n_waves) {
        const int info = __builtin_amdgcn_readfirstlane(info_n);
        // (every lane has read the record's header from the same address: say so, or the whole pass runs under a lane mask)
        const float a = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a_n))), c0 = c_n;
        const int cnt = __builtin_amdgcn_readfirstlane(cnt_n);
        const int nxt = rec + n_waves < n_rec ? rec + n_waves : rec;  // (the last pass reads its own record again)
        info_n = rec_info[nxt];
        cnt_n = post_counts[nxt];
        a_n = amp[nxt];
        c_n = coeff[(size_t)nxt * coeff_stride + pre];
        if (!(info & 0x40)) continue;
        // what floor1_unwrap_kernel leaves for a type-0 record, so that a batch without type-1 floors needs no unwrap launch: the
        // marker count (0: ExecuteChannel false -- Floor0.Data: Amp > 0 --, the channel's block is zeros and nobody reads its
        // curve) and, where the first active post would be, which floor it is
        const bool live = cnt != 0 && a > 0.0f;
        if (lane == 0) {
            ccount[rec] = live ? (uint8_t)kFloor0Marker : (uint8_t)0;
            cposts[(size_t)rec * 64] = info & 0x3F;
        }
        if (!live) continue;
        if ((info & 0x3F) != cur_floor) {
            cur_floor = info & 0x3F;
            f = floors[cur_floor];
        }
        const int order = __builtin_amdgcn_readfirstlane(f.order), bark_size = __builtin_amdgcn_readfirstlane(f.bark_map_size);
        if (lane < order) s_c[lane] = 2.0f * cos_rounded_once(c0);
        for (int j = lane + 64; j < order; j += 64) s_c[j] = 2.0f * cos_rounded_once(coeff[(size_t)rec * coeff_stride + j]);
        __builtin_amdgcn_wave_barrier();
        // (the factors' coefficients come four to an LDS read -- every lane reads the same address, a broadcast.  Parking them in a
        // register's lanes and fetching them with v_readlane_b32 was measured slower)
        const float amp_ofs = (float)__builtin_amdgcn_readfirstlane(f.amp_ofs);
        const float *wt = wtab + (size_t)cur_floor * k_stride;
        // four bark indices per lane at a time (k = k0 + lane + 64 m): their table values are asked for together, every coefficient
        // read serves all four, and the eight products are independent chains
        const int kmax = bark_size < k_stride ? bark_size : k_stride;
        for (int k0 = 0; k0 < kmax; k0 += 256) {
            float w[4], p[4], q[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int k = k0 + lane + 64 * m;
                w[m] = wt[k < kmax ? k : 0];
                p[m] = .5f;
                q[m] = .5f;
            }
            int j = 1;
            for (; j + 2 < order; j += 4) {  // factors j - 1, j, j + 1, j + 2: the reference's loop body twice (Floor0.cs:195-199)
                const float4 c = s_c4[(j - 1) >> 2];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    q[m] *= w[m] - c.x;
                    p[m] *= w[m] - c.y;
                    q[m] *= w[m] - c.z;
                    p[m] *= w[m] - c.w;
                }
            }
            for (; j < order; j += 2) {
                const float c0j = s_c[j - 1], c1j = s_c[j];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    q[m] *= w[m] - c0j;
                    p[m] *= w[m] - c1j;
                }
            }
            const bool odd = j == order;  // (wave-uniform)
            float e[4];
            if (odd) {
                const float clast = s_c[j - 1];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    q[m] *= w[m] - clast;
                    p[m] *= p[m] * (4.0f - w[m] * w[m]);
                    q[m] *= q[m];
                }
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    p[m] *= p[m] * (2.0f - w[m]);
                    q[m] *= q[m] * (2.0f + w[m]);
                }
            }
            // (amp / sqrt(p + q) in float32, both correctly rounded -- hipcc's default for float division and square root --, which
            // is what rounding the double-precision quotient of the double-precision root gives: 53 bits are more than 2 * 24 + 2)
#pragma unroll
            for (int m = 0; m < 4; ++m) e[m] = (a / sqrtf(p[m] + q[m]) - amp_ofs) * 0.11512925f;
            float v[4];
            exp_rounded_once_x4(e, v);
#pragma unroll
            // (k_stride is a multiple of 256 -- the host's word --, so a round is stored whole: a value under `if (k < kmax)` would be
            // COMPUTED under it too, four times one after the other instead of side by side.  Nobody reads beyond bark_map_size)
            for (int m = 0; m < 4; ++m) curve[(size_t)rec * k_stride + k0 + lane + 64 * m] = v[m];
        }
        __builtin_amdgcn_wave_barrier();  // (the next record's coefficients go where this one's are read)
    }
}

hipError_t launch_floor0_curves(int n_rec, const uint8_t *rec_info, const void *floors, const float *amp, const float *coeff,
                                int coeff_stride, int k_stride, const float *wtab, float *curve, const uint8_t *post_counts,
                                uint8_t *ccount, int32_t *cposts, hipStream_t stream)
{
    if (n_rec <= 0) return hipSuccess;
    if (k_stride <= 0 || (k_stride & 255)) return hipErrorInvalidValue;
    int groups = (n_rec + kCurveWaves - 1) / kCurveWaves;
    if (groups > kCurveGroupsMax) groups = kCurveGroupsMax;
    hipLaunchKernelGGL(floor0_curve_kernel, dim3(groups), dim3(64 * kCurveWaves), 0, stream, n_rec, rec_info,
                       static_cast<const Floor0Dev *>(floors), amp, coeff, coeff_stride, k_stride, wtab, curve, post_counts, ccount, cposts);
    return hipGetLastError();
}

hipError_t launch_floor0_apply(const void *recs, int n_recs, const void *floors, const int32_t *bark_maps,
                               const float *amp, const float *coeff, int coeff_stride, float *spec,
                               hipStream_t stream)
{
    if (n_recs <= 0) return hipSuccess;
    hipLaunchKernelGGL(floor0_apply_kernel, dim3(n_recs), dim3(256), 0, stream,
                       static_cast<const Floor0Rec *>(recs), static_cast<const Floor0Dev *>(floors), bark_maps, amp,
                       coeff, coeff_stride, spec);
    return hipGetLastError();
}

size_t floor0_dev_size() { return sizeof(Floor0Dev); }
size_t floor0_rec_size() { return sizeof(Floor0Rec); }
void fill_floor0_dev(void *dst, int order, int bark_map_size, int amp_ofs, int64_t off_short, int64_t off_long)
{
    Floor0Dev *d = static_cast<Floor0Dev *>(dst);
    d->order = order;
    d->bark_map_size = bark_map_size;
    d->amp_ofs = amp_ofs;
    d->reserved = 0;
    d->bark_off[0] = off_short;
    d->bark_off[1] = off_long;
}
void fill_floor0_rec(void *dst, int64_t spec_off, int rec, int floor, int half, int is_long)
{
    Floor0Rec *r = static_cast<Floor0Rec *>(dst);
    r->spec_off = spec_off;
    r->rec = rec;
    r->floor = floor;
    r->half = half;
    r->is_long = is_long;
}

}  // namespace vpz
