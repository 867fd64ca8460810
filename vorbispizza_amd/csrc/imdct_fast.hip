// Batched inverse MDCT kernels (vpz_imdct_batch, VPZ_IMDCT_FAST): `Mdct.Reverse` semantics
// (Mdct.cs:15-19) -- N/2 spectral values in, all N time-domain values out, per channel-block.
//
// HBM-bound: 4*N/2 bytes read + 4*N bytes written per channel-block (12 288 B at N = 2048).
// One wavefront per channel-block (N = 2048) or per 8 channel-blocks (N = 256); persistent
// workgroups of 4 wavefronts grid-stride over the batch with the next block's loads in flight
// while the current one is transformed.  All global accesses are 8 or 16 bytes per lane and each
// wave-instruction covers one contiguous 512 B / 1 KiB span.
#include <cstdlib>

#include "imdct_core.hpp"
#include "vpz_internal.hpp"

namespace vpz {

constexpr int kWavesPerGroup = 4;
constexpr int kThreads = 64 * kWavesPerGroup;

// Write y (N floats) from h (N/2 floats in LDS): y = [-rev(h[0:N/4]), h, rev(h[N/4:N/2])].
// `lanes` lanes cooperate, lane index `l`; float4 granularity, N/8 float4 of h.
template <int N, int LANES>
__device__ __forceinline__ void store_full_block(const float *h, float *out, int l)
{
    constexpr int H4 = N / 8;       // float4 count of h
    constexpr int Q4 = N / 16;      // float4 count of a quarter block
    const float4 *h4 = reinterpret_cast<const float4 *>(h);
    float4 *o4 = reinterpret_cast<float4 *>(out);
#pragma unroll
    for (int r = 0; r < H4 / LANES; ++r) {
        const int f = l + LANES * r;
        float4 v = h4[f];
        // streaming (non-temporal) stores: the PCM is written once and never re-read here
        store_nt(&o4[Q4 + f], v);
        float4 rv = make_float4(v.w, v.z, v.y, v.x);
        if (r < (H4 / LANES) / 2) {  // f < Q4: first quarter, negated mirror
            store_nt(&o4[Q4 - 1 - f], make_float4(-rv.x, -rv.y, -rv.z, -rv.w));
        } else {                     // last quarter, mirror
            store_nt(&o4[3 * Q4 + (H4 - 1 - f)], rv);
        }
    }
}

// kGather: block b reads at spectra + src_off[b] and writes at out + dst_off[b] (the three-pass decoder path);
// the plain instantiation is the dense batch of vpz_imdct_batch, untouched by this option
template <bool kGather>
__global__ __launch_bounds__(kThreads) void imdct2048_kernel(const float *__restrict__ spectra,
                                                             float *__restrict__ out,
                                                             long count,
                                                             const float2 *__restrict__ tables,
                                                             const int64_t *__restrict__ src_off,
                                                             const int64_t *__restrict__ dst_off)
{
    __shared__ float2 s_tw[512];
    __shared__ float2 s_twAB[512];
    __shared__ float2 s_twBC[64];
    __shared__ float2 s_scratch[kWavesPerGroup][kWaveScratchFloat2];

    for (int i = threadIdx.x; i < 512; i += kThreads) {
        s_tw[i] = tables[kFastTwOffset + i];
        s_twAB[i] = tables[kFastTwABOffset + i];
    }
    if (threadIdx.x < 64) s_twBC[threadIdx.x] = tables[kFastTwBCOffset + threadIdx.x];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float2 *scratch = s_scratch[wave];
    const long stride = (long)gridDim.x * kWavesPerGroup;
    long blk = (long)blockIdx.x * kWavesPerGroup + wave;
    if (blk >= count) return;

    float2 cur[8];
    {
        const float2 *src = reinterpret_cast<const float2 *>(spectra + (kGather ? src_off[blk] : blk * 1024));
#pragma unroll
        for (int m = 0; m < 8; ++m) cur[m] = src[lane + 64 * m];
    }
    // The first block has to be there before the loop is entered, and the loads of the next one are unconditional (the
    // last pass re-reads its own block): with loads pending at the loop header, or under a condition, the compiler
    // waits for EVERYTHING at the first use of `cur` -- i.e. for the prefetch it has just issued -- and the transform
    // never overlaps a load of its own wave (found in round 2 with the ISA listing; the fused kernel had the same).
#pragma unroll
    for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(cur[m].x), "v"(cur[m].y));
    while (true) {
        const long nxt = blk + stride;
        float2 pre[8];
        {
            const long ld = nxt < count ? nxt : blk;
            const float2 *src = reinterpret_cast<const float2 *>(spectra + (kGather ? src_off[ld] : ld * 1024));
#pragma unroll
            for (int m = 0; m < 8; ++m) pre[m] = src[lane + 64 * m];
        }
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the loads into the transform to save registers)
        imdct2048_wave(cur, scratch, s_tw, s_twAB, s_twBC, lane);
        // (vmcnt counts stores too, in issue order: the wait for the prefetch is placed ahead of this block's stores)
#pragma unroll
        for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(pre[m].x), "v"(pre[m].y));
        store_full_block<2048, 64>(reinterpret_cast<const float *>(scratch), out + (kGather ? dst_off[blk] : blk * 2048), lane);
        if (nxt >= count) break;
#pragma unroll
        for (int m = 0; m < 8; ++m) cur[m] = pre[m];
        blk = nxt;
    }
}

template <bool kGather>
__global__ __launch_bounds__(kThreads) void imdct256_kernel(const float *__restrict__ spectra,
                                                            float *__restrict__ out,
                                                            long count,
                                                            const float2 *__restrict__ tables,
                                                            const int64_t *__restrict__ src_off,
                                                            const int64_t *__restrict__ dst_off)
{
    __shared__ float2 s_tw[64];
    __shared__ float2 s_twBC[64];
    __shared__ float2 s_scratch[kWavesPerGroup][kWaveScratchFloat2];
    if (threadIdx.x < 64) {
        s_tw[threadIdx.x] = tables[kFastTwOffset + threadIdx.x];
        s_twBC[threadIdx.x] = tables[kFastTwBCOffset + threadIdx.x];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 3, l = lane & 7;
    float2 *scratch = s_scratch[wave];
    const long stride = (long)gridDim.x * kWavesPerGroup * 8;
    // every wavefront owns 8 consecutive channel-blocks per step; tail groups clamp and skip stores
    // (software pipeline as in imdct2048_kernel: the next step's loads are in flight during the transform)
    long base = ((long)blockIdx.x * kWavesPerGroup + wave) * 8;
    if (base >= count) return;
    auto load_step = [&](long at, float2 (&x)[8]) {
        const long ld = at + g < count ? at + g : count - 1;
        const float2 *src = reinterpret_cast<const float2 *>(spectra + (kGather ? src_off[ld] : ld * 128));
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = src[l + 8 * m];
    };
    float2 xa[8];
    load_step(base, xa);
#pragma unroll
    for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(xa[m].x), "v"(xa[m].y));
    while (true) {
        const long nxt = base + stride;
        float2 pre[8];
        load_step(nxt < count ? nxt : base, pre);
        __builtin_amdgcn_sched_barrier(0);
        const long blk = base + g;
        imdct256_wave8(xa, scratch, s_tw, s_twBC, lane);
#pragma unroll
        for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(pre[m].x), "v"(pre[m].y));
        if (blk < count)
            store_full_block<256, 8>(reinterpret_cast<const float *>(scratch) + g * 128,
                                     out + (kGather ? dst_off[blk] : blk * 256), l);
        if (nxt >= count) break;
#pragma unroll
        for (int m = 0; m < 8; ++m) xa[m] = pre[m];
        base = nxt;
    }
}

// N = 512 (R = 2) / N = 1024 (R = 4): 8/R channel-blocks per wavefront
template <int R>
__global__ __launch_bounds__(kThreads) void imdct_mid_kernel(const float *__restrict__ spectra, float *__restrict__ out,
                                                            long count, const float2 *__restrict__ tables,
                                                            const int64_t *__restrict__ src_off,
                                                            const int64_t *__restrict__ dst_off)
{
    constexpr int L = 8 * R, M = 64 * R, N = 256 * R, B = 8 / R;
    __shared__ float2 s_tw[M];
    __shared__ float2 s_twAB[M];
    __shared__ float2 s_twBC[L];
    __shared__ float2 s_scratch[kWavesPerGroup][kWaveScratchFloat2];
    for (int i = threadIdx.x; i < M; i += kThreads) {
        s_tw[i] = tables[kFastTwOffset + i];
        s_twAB[i] = tables[kFastTwABOffset + i];
    }
    if (threadIdx.x < L) s_twBC[threadIdx.x] = tables[kFastTwBCOffset + threadIdx.x];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = lane / L, l = lane & (L - 1);
    float2 *scratch = s_scratch[wave];
    const long stride = (long)gridDim.x * kWavesPerGroup * B;
    long base = ((long)blockIdx.x * kWavesPerGroup + wave) * B;
    if (base >= count) return;
    auto load_step = [&](long at, float2 (&x)[8]) {
        const long ld = at + b < count ? at + b : count - 1;
        const float2 *src = reinterpret_cast<const float2 *>(spectra + (src_off ? src_off[ld] : ld * (N / 2)));
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = src[l + L * m];
    };
    float2 xa[8];
    load_step(base, xa);
#pragma unroll
    for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(xa[m].x), "v"(xa[m].y));
    while (true) {
        const long nxt = base + stride;
        float2 pre[8];
        load_step(nxt < count ? nxt : base, pre);
        __builtin_amdgcn_sched_barrier(0);
        const long blk = base + b;
        imdct_mid_wave<R>(xa, scratch, s_tw, s_twAB, s_twBC, lane);
#pragma unroll
        for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(pre[m].x), "v"(pre[m].y));
        if (blk < count)
            store_full_block<N, L>(reinterpret_cast<const float *>(scratch) + b * (N / 2),
                                   out + (dst_off ? dst_off[blk] : blk * N), l);
        if (nxt >= count) break;
#pragma unroll
        for (int m = 0; m < 8; ++m) xa[m] = pre[m];
        base = nxt;
    }
}

// N = 4096: one wavefront per channel-block, 16-byte loads (X[4k' .. 4k'+3] per lane and point)
__global__ __launch_bounds__(kThreads) void imdct4096_kernel(const float *__restrict__ spectra, float *__restrict__ out,
                                                            long count, const float2 *__restrict__ tables,
                                                            const int64_t *__restrict__ src_off,
                                                            const int64_t *__restrict__ dst_off)
{
    __shared__ float2 s_tw[1024];
    __shared__ float2 s_twAB[512];
    __shared__ float2 s_twBC[64];
    __shared__ float2 s_w[512];
    __shared__ float2 s_h[kWavesPerGroup][1024];
    for (int i = threadIdx.x; i < 1024; i += kThreads) s_tw[i] = tables[kFast4096TwOffset + i];
    for (int i = threadIdx.x; i < 512; i += kThreads) {
        s_twAB[i] = tables[kFast4096TwABOffset + i];
        s_w[i] = tables[kFast4096WOffset + i];
    }
    if (threadIdx.x < 64) s_twBC[threadIdx.x] = tables[kFast4096TwBCOffset + threadIdx.x];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long stride = (long)gridDim.x * kWavesPerGroup;
    for (long blk = (long)blockIdx.x * kWavesPerGroup + wave; blk < count; blk += stride) {
        const float4 *src = reinterpret_cast<const float4 *>(spectra + (src_off ? src_off[blk] : blk * 2048));
        float4 xa[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) xa[m] = src[lane + 64 * m];
        imdct4096_wave(xa, s_h[wave], s_tw, s_twAB, s_twBC, s_w, lane);
        store_full_block<4096, 64>(reinterpret_cast<const float *>(s_h[wave]), out + (dst_off ? dst_off[blk] : blk * 4096), lane);
    }
}

// N = 8192: one wavefront per channel-block, two 16-byte loads per lane and point.  Every table is in LDS -- the 2048-entry
// twiddle table and the read half of the level-2 table too: read from global memory inside the butterflies they were some twenty
// trips to the L2 per block, one behind the other (synth_big.hip has the measurements) --, at the price of one wave: seven waves
// of 16 KB beside 28.5 KB of tables, one workgroup per CU.
constexpr int kWaves8192 = 7;
constexpr int kThreads8192 = 64 * kWaves8192;
__global__ __launch_bounds__(kThreads8192) void imdct8192_kernel(const float *__restrict__ spectra, float *__restrict__ out,
                                                                long count, const float2 *__restrict__ tables,
                                                                const int64_t *__restrict__ src_off,
                                                                const int64_t *__restrict__ dst_off)
{
    __shared__ float2 s_twAB[512];
    __shared__ float2 s_twBC[64];
    __shared__ float2 s_w1[512];
    __shared__ float2 s_tw[2048];
    __shared__ float2 s_w2[512];
    __shared__ float2 s_h[kWaves8192][2048];
    for (int i = threadIdx.x; i < 512; i += kThreads8192) {
        s_twAB[i] = tables[kFast8192TwABOffset + i];
        s_w1[i] = tables[kFast8192W1Offset + i];
        s_w2[i] = tables[kFast8192W2Offset + i];
    }
    for (int i = threadIdx.x; i < 2048; i += kThreads8192) s_tw[i] = tables[kFast8192TwOffset + i];
    if (threadIdx.x < 64) s_twBC[threadIdx.x] = tables[kFast8192TwBCOffset + threadIdx.x];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long stride = (long)gridDim.x * kWaves8192;
    for (long blk = (long)blockIdx.x * kWaves8192 + wave; blk < count; blk += stride) {
        const float4 *src = reinterpret_cast<const float4 *>(spectra + (src_off ? src_off[blk] : blk * 4096));
        float4 lo[8], hi[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            lo[m] = src[2 * (lane + 64 * m)];
            hi[m] = src[2 * (lane + 64 * m) + 1];
        }
        imdct8192_wave(lo, hi, s_h[wave], s_tw, s_w2, s_w1, s_twAB, s_twBC, lane);
        store_full_block<8192, 64>(reinterpret_cast<const float *>(s_h[wave]), out + (dst_off ? dst_off[blk] : blk * 8192), lane);
    }
}

// Persistent grid: exactly as many workgroups as the chip keeps resident (CUs x measured
// occupancy), so that every workgroup gets the same share of the batch and there is no partial
// last round of workgroups.
template <typename K>
static int resident_groups(K kernel, int num_cu, int threads = kThreads)
{
    int per_cu = 0;
    if (const char *e = getenv("VPZ_IMDCT_GROUPS_PER_CU")) {  // tuning experiments only
        per_cu = atoi(e);
        if (per_cu > 0) return num_cu * per_cu;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 2;
    }
    // Measured on MI355X (profiles/r1_grid_sweep.txt): 2 workgroups (8 wavefronts) per CU stream
    // HBM best for this 1:2 read:write shape; more resident waves only add DRAM page conflicts.
    if (per_cu > 2) per_cu = 2;
    return num_cu * per_cu;
}

static int grid_for(int64_t work_groups, int resident)
{
    int64_t g = work_groups < resident ? work_groups : resident;
    return (int)(g < 1 ? 1 : g);
}

hipError_t launch_imdct_fast_2048(const float *spectra, float *out, int64_t count,
                                  const float2 *tw, Context *ctx, hipStream_t stream, const int64_t *src_off,
                                  const int64_t *dst_off)
{
    if (count <= 0) return hipSuccess;
    int &resident = ctx->resident[kResident2048];  // (per context: the contexts of one process may sit on different devices)
    if (!resident) resident = resident_groups(imdct2048_kernel<false>, ctx->num_cu);
    int grid = grid_for((count + kWavesPerGroup - 1) / kWavesPerGroup, resident);
    if (src_off && dst_off)
        hipLaunchKernelGGL(imdct2048_kernel<true>, dim3(grid), dim3(kThreads), 0, stream, spectra, out, (long)count, tw,
                           src_off, dst_off);
    else
        hipLaunchKernelGGL(imdct2048_kernel<false>, dim3(grid), dim3(kThreads), 0, stream, spectra, out, (long)count, tw,
                           nullptr, nullptr);
    return hipGetLastError();
}

hipError_t launch_imdct_fast_256(const float *spectra, float *out, int64_t count,
                                 const float2 *tw, Context *ctx, hipStream_t stream, const int64_t *src_off,
                                 const int64_t *dst_off)
{
    if (count <= 0) return hipSuccess;
    int &resident = ctx->resident[kResident256];  // (per context: the contexts of one process may sit on different devices)
    if (!resident) resident = resident_groups(imdct256_kernel<false>, ctx->num_cu);
    int64_t per_group = kWavesPerGroup * 8;
    int grid = grid_for((count + per_group - 1) / per_group, resident);
    if (src_off && dst_off)
        hipLaunchKernelGGL(imdct256_kernel<true>, dim3(grid), dim3(kThreads), 0, stream, spectra, out, (long)count, tw,
                           src_off, dst_off);
    else
        hipLaunchKernelGGL(imdct256_kernel<false>, dim3(grid), dim3(kThreads), 0, stream, spectra, out, (long)count, tw,
                           nullptr, nullptr);
    return hipGetLastError();
}

hipError_t launch_imdct_fast_4096(const float *spectra, float *out, int64_t count, const float2 *tw, Context *ctx,
                                  hipStream_t stream, const int64_t *src_off, const int64_t *dst_off)
{
    if (count <= 0) return hipSuccess;
    int &resident = ctx->resident[kResident4096];  // (per context: the contexts of one process may sit on different devices)
    if (!resident) resident = resident_groups(imdct4096_kernel, ctx->num_cu);
    const int grid = grid_for((count + kWavesPerGroup - 1) / kWavesPerGroup, resident);
    hipLaunchKernelGGL(imdct4096_kernel, dim3(grid), dim3(kThreads), 0, stream, spectra, out, (long)count, tw, src_off,
                       dst_off);
    return hipGetLastError();
}

hipError_t launch_imdct_fast_8192(const float *spectra, float *out, int64_t count, const float2 *tw, Context *ctx,
                                  hipStream_t stream, const int64_t *src_off, const int64_t *dst_off)
{
    if (count <= 0) return hipSuccess;
    int &resident = ctx->resident[kResident8192];  // (per context: the contexts of one process may sit on different devices)
    if (!resident) resident = resident_groups(imdct8192_kernel, ctx->num_cu, kThreads8192);
    const int grid = grid_for((count + kWaves8192 - 1) / kWaves8192, resident);
    hipLaunchKernelGGL(imdct8192_kernel, dim3(grid), dim3(kThreads8192), 0, stream, spectra, out, (long)count, tw, src_off,
                       dst_off);
    return hipGetLastError();
}

hipError_t launch_imdct_fast_mid(int n, const float *spectra, float *out, int64_t count, const float2 *tw, Context *ctx,
                                 hipStream_t stream, const int64_t *src_off, const int64_t *dst_off)
{
    if (count <= 0) return hipSuccess;
    if (n == 512) {
        int &resident = ctx->resident[kResident512];  // (per context: the contexts of one process may sit on different devices)
        if (!resident) resident = resident_groups(imdct_mid_kernel<2>, ctx->num_cu);
        const int64_t per_group = kWavesPerGroup * 4;
        hipLaunchKernelGGL(imdct_mid_kernel<2>, dim3(grid_for((count + per_group - 1) / per_group, resident)),
                           dim3(kThreads), 0, stream, spectra, out, (long)count, tw, src_off, dst_off);
    } else if (n == 1024) {
        int &resident = ctx->resident[kResident1024];  // (per context: the contexts of one process may sit on different devices)
        if (!resident) resident = resident_groups(imdct_mid_kernel<4>, ctx->num_cu);
        const int64_t per_group = kWavesPerGroup * 2;
        hipLaunchKernelGGL(imdct_mid_kernel<4>, dim3(grid_for((count + per_group - 1) / per_group, resident)),
                           dim3(kThreads), 0, stream, spectra, out, (long)count, tw, src_off, dst_off);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace vpz
