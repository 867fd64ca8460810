// What the memory system of THIS box delivers -- the denominator of every "fraction of what the memory delivers" in
// DESIGN.md / BASELINE.md.  Four probes, each swept over the launch shape (workgroups per CU x threads) and over plain /
// non-temporal accesses; the best line of each is the ceiling quoted:
//   copy      float4 copy, 1:1, UNROLLED: `U` independent 16-byte loads in flight per lane before the first store
//             (MI355X_MICROARCH.md measures 6.29 TB/s for a float4 copy; a grid-stride loop without unrolling -- round 2's
//             probe -- keeps one load in flight per lane and reaches 4.8-5.2)
//   read      16-byte loads only (sum folded into one store per lane at the end)
//   write     16-byte stores only
//   imdct     the I/O shape of imdct2048_kernel with the arithmetic removed: per wavefront and block 8 x 512 B loads
//             (8 B/lane) and 8 x 1 KiB stores (16 B/lane), 1:2 read:write, next block's loads ahead of the stores
//   fused     the I/O shape of synth_kernel's steady state (all-long stereo): per wavefront and block 4 KiB in, 4 KiB out
// Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/io_ceiling.hip -o /tmp/io_ceiling && /tmp/io_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define GLOBAL __attribute__((address_space(1)))

template <bool NT> __device__ __forceinline__ f4v ld4(const f4v *p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT> __device__ __forceinline__ void st4(f4v *p, f4v v)
{
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// every workgroup owns a contiguous slab (its CU's L2 / channel locality does not depend on the grid), walks it in
// tiles of U x blockDim float4: U loads, then U stores
template <int U, bool NTL, bool NTS>
__global__ void copy_unrolled(const f4v *__restrict__ in, f4v *__restrict__ out, long n4)
{
    const long per_wg = (n4 + gridDim.x - 1) / gridDim.x;
    const long lo = per_wg * blockIdx.x, hi = lo + per_wg < n4 ? lo + per_wg : n4;
    const long tile = (long)U * blockDim.x;
    for (long base = lo; base < hi; base += tile) {
        f4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + (long)u * blockDim.x + threadIdx.x;
            v[u] = ld4<NTL>(in + (i < hi ? i : hi - 1));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + (long)u * blockDim.x + threadIdx.x;
            if (i < hi) st4<NTS>(out + i, v[u]);
        }
    }
}

// the same with the grid-stride walk round 2's probe used (kept for comparison)
__global__ void copy_gridstride(const f4v *__restrict__ in, f4v *__restrict__ out, long n4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}

template <int U, bool NT>
__global__ void read_only(const f4v *__restrict__ in, f4v *__restrict__ out, long n4)
{
    const long per_wg = (n4 + gridDim.x - 1) / gridDim.x;
    const long lo = per_wg * blockIdx.x, hi = lo + per_wg < n4 ? lo + per_wg : n4;
    const long tile = (long)U * blockDim.x;
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    for (long base = lo; base < hi; base += tile) {
        f4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + (long)u * blockDim.x + threadIdx.x;
            v[u] = ld4<NT>(in + (i < hi ? i : hi - 1));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int U, bool NT>
__global__ void write_only(f4v *__restrict__ out, long n4)
{
    const long per_wg = (n4 + gridDim.x - 1) / gridDim.x;
    const long lo = per_wg * blockIdx.x, hi = lo + per_wg < n4 ? lo + per_wg : n4;
    const long tile = (long)U * blockDim.x;
    const f4v v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (long base = lo; base < hi; base += tile) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + (long)u * blockDim.x + threadIdx.x;
            if (i < hi) st4<NT>(out + i, v);
        }
    }
}

// imdct2048_kernel's traffic: block b reads in[b*1024 .. +1024) as 8 x float2 per lane, writes out[b*2048 .. +2048) as
// 8 x float4 per lane; the next block's loads are issued before this block's stores (persistent waves)
template <bool NTS>
__global__ __launch_bounds__(256) void io_imdct(const float *__restrict__ in, float *__restrict__ out, long count)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long stride = (long)gridDim.x * 4;
    long blk = (long)blockIdx.x * 4 + wave;
    if (blk >= count) return;
    float2 x[8];
    {
        const float2 *src = reinterpret_cast<const float2 *>(in + blk * 1024);
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = src[lane + 64 * m];
    }
    for (; blk < count; blk += stride) {
        float2 y[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) y[m] = x[m];
#pragma unroll
        for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(y[m].x), "v"(y[m].y));
        const long nb = blk + stride < count ? blk + stride : blk;
        const float2 *src = reinterpret_cast<const float2 *>(in + nb * 1024);
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = src[lane + 64 * m];
        __builtin_amdgcn_sched_barrier(0);
        f4v *o4 = reinterpret_cast<f4v *>(out + blk * 2048);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f4v v = {y[2 * r].x, y[2 * r].y, y[2 * r + 1].x, y[2 * r + 1].y};
            const f4v w = {-v.w, -v.z, -v.y, -v.x};
            st4<NTS>(o4 + lane + 64 * r, v);
            st4<NTS>(o4 + 256 + lane + 64 * r, w);
        }
    }
}

// synth_kernel's steady state, all-long stereo: a wavefront walks a RUN of consecutive blocks of one channel (4 KiB in,
// 4 KiB out each), the next block's loads ahead of this block's stores.  W16: loads of 16 bytes per lane (the stereo
// fast path reads the Residue2 vector that way) instead of 8.
template <bool W16, bool NTS>
__global__ __launch_bounds__(512) void io_fused(const float *__restrict__ in, float *__restrict__ out, long runs, int run_len)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long run = (long)blockIdx.x * 8 + wave;
    if (run >= runs) return;
    const float *src = in + run * run_len * 1024;
    float *dst = out + run * run_len * 1024;
    f4v x[4];
    auto load = [&](const float *p) {
        if (W16) {
#pragma unroll
            for (int m = 0; m < 4; ++m) x[m] = reinterpret_cast<const f4v *>(p)[lane + 64 * m];
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f2v a = reinterpret_cast<const f2v *>(p)[lane + 64 * (2 * m)];
                const f2v b = reinterpret_cast<const f2v *>(p)[lane + 64 * (2 * m + 1)];
                x[m] = f4v{a.x, a.y, b.x, b.y};
            }
        }
    };
    load(src);
    for (int f = 0; f < run_len; ++f) {
        f4v y[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) y[m] = x[m];
#pragma unroll
        for (int m = 0; m < 4; ++m) asm volatile("" ::"v"(y[m]));
        load(src + (long)(f + 1 < run_len ? f + 1 : f) * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) st4<NTS>(reinterpret_cast<f4v *>(dst + (long)f * 1024) + lane + 64 * m, y[m]);
    }
}

template <typename F> static float time_us(F f, int reps = 11)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<float> t;
    for (int i = 0; i < 2; ++i) f();
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(a);
        f();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        t.push_back(ms * 1e3f);
    }
    hipEventDestroy(a);
    hipEventDestroy(b);
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

struct Best {
    double gbs = 0;
    char what[128] = "";
    void take(double g, const char *fmt, int a, int b, const char *c)
    {
        if (g > gbs) { gbs = g; snprintf(what, sizeof what, fmt, a, b, c); }
    }
};

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const long n4 = 1L << 26;  // 1 GiB in, 1 GiB out: far beyond the 256 MiB Infinity Cache
    f4v *in, *out;
    hipMalloc(&in, n4 * 16);
    hipMalloc(&out, n4 * 16 + 4096);
    hipMemset(in, 0x11, n4 * 16);
    printf("device: %s, %d CUs; buffers 2 x %.0f MiB\n", prop.gcnArchName, cus, n4 * 16.0 / (1 << 20));
    Best copy, rd, wr;
#define SWEEP_COPY(U, NTL, NTS, tag)                                                                                     \
    for (int threads : {256, 512})                                                                                      \
        for (int per_cu : {2, 4, 8, 16}) {                                                                              \
            if (threads * per_cu > 2048) continue;                                                                      \
            const int grid = cus * per_cu;                                                                              \
            const float us = time_us([&] { hipLaunchKernelGGL((copy_unrolled<U, NTL, NTS>), dim3(grid), dim3(threads), 0, 0, in, out, n4); }); \
            const double g = n4 * 32.0 / us / 1e3;                                                                      \
            printf("copy  U=%d %-14s %4d thr x %2d WG/CU: %7.1f us  %5.0f GB/s\n", U, tag, threads, per_cu, us, g);     \
            copy.take(g, "U=" #U " %d threads x %d WG/CU %s", threads, per_cu, tag);                                    \
        }
    SWEEP_COPY(1, false, false, "plain")
    SWEEP_COPY(4, false, false, "plain")
    SWEEP_COPY(4, false, true, "nt-store")
    SWEEP_COPY(4, true, true, "nt-load+store")
    SWEEP_COPY(8, false, false, "plain")
    SWEEP_COPY(8, false, true, "nt-store")
    for (int per_cu : {2, 4, 8}) {
        const float us = time_us([&] { hipLaunchKernelGGL(copy_gridstride, dim3(cus * per_cu), dim3(256), 0, 0, in, out, n4); });
        printf("copy  grid-stride, no unrolling (round 2's probe) 256 thr x %d WG/CU: %7.1f us  %5.0f GB/s\n", per_cu, us, n4 * 32.0 / us / 1e3);
    }
#define SWEEP_RD(U, NT, tag)                                                                                             \
    for (int per_cu : {2, 4, 8}) {                                                                                      \
        const int grid = cus * per_cu;                                                                                  \
        const float us = time_us([&] { hipLaunchKernelGGL((read_only<U, NT>), dim3(grid), dim3(256), 0, 0, in, out, n4); }); \
        const double g = n4 * 16.0 / us / 1e3;                                                                          \
        printf("read  U=%d %-10s 256 thr x %2d WG/CU: %7.1f us  %5.0f GB/s\n", U, tag, per_cu, us, g);                  \
        rd.take(g, "U=" #U " %d threads x %d WG/CU %s", 256, per_cu, tag);                                              \
    }
    SWEEP_RD(4, false, "plain")
    SWEEP_RD(8, false, "plain")
    SWEEP_RD(8, true, "nt")
#define SWEEP_WR(U, NT, tag)                                                                                             \
    for (int per_cu : {2, 4, 8}) {                                                                                      \
        const int grid = cus * per_cu;                                                                                  \
        const float us = time_us([&] { hipLaunchKernelGGL((write_only<U, NT>), dim3(grid), dim3(256), 0, 0, out, n4); }); \
        const double g = n4 * 16.0 / us / 1e3;                                                                          \
        printf("write U=%d %-10s 256 thr x %2d WG/CU: %7.1f us  %5.0f GB/s\n", U, tag, per_cu, us, g);                  \
        wr.take(g, "U=" #U " %d threads x %d WG/CU %s", 256, per_cu, tag);                                              \
    }
    SWEEP_WR(4, false, "plain")
    SWEEP_WR(4, true, "nt")
    // ---- the two kernel shapes
    const long count = 131072;  // config[1]: 512 MiB in, 1 GiB out
    Best im, fu;
    for (int per_cu : {1, 2, 3, 4}) {
        const int grid = cus * per_cu;
        const float u0 = time_us([&] { hipLaunchKernelGGL(io_imdct<false>, dim3(grid), dim3(256), 0, 0, (const float *)in, (float *)out, count); });
        const float u1 = time_us([&] { hipLaunchKernelGGL(io_imdct<true>, dim3(grid), dim3(256), 0, 0, (const float *)in, (float *)out, count); });
        printf("imdct shape (1:2) %d WG/CU: plain %7.1f us %5.0f GB/s | nt-store %7.1f us %5.0f GB/s\n", per_cu, u0,
               count * 12288.0 / u0 / 1e3, u1, count * 12288.0 / u1 / 1e3);
        im.take(count * 12288.0 / u0 / 1e3, "%d WG/CU x %d waves %s", per_cu, 4, "plain");
        im.take(count * 12288.0 / u1 / 1e3, "%d WG/CU x %d waves %s", per_cu, 4, "nt-store");
    }
    for (int run_len : {16, 32}) {
        const long runs = 131072 / run_len;  // 131 072 channel-blocks: 512 MiB in, 512 MiB out
        const int grid = (int)((runs + 7) / 8);
        const double bytes = 131072.0 * 8192.0;
        const float a = time_us([&] { hipLaunchKernelGGL((io_fused<false, false>), dim3(grid), dim3(512), 0, 0, (const float *)in, (float *)out, runs, run_len); });
        const float b = time_us([&] { hipLaunchKernelGGL((io_fused<false, true>), dim3(grid), dim3(512), 0, 0, (const float *)in, (float *)out, runs, run_len); });
        const float c = time_us([&] { hipLaunchKernelGGL((io_fused<true, false>), dim3(grid), dim3(512), 0, 0, (const float *)in, (float *)out, runs, run_len); });
        const float d = time_us([&] { hipLaunchKernelGGL((io_fused<true, true>), dim3(grid), dim3(512), 0, 0, (const float *)in, (float *)out, runs, run_len); });
        printf("fused shape (1:1) runs of %2d, %d workgroups: 8B loads %7.1f us %5.0f GB/s | + nt-store %7.1f us %5.0f | 16B loads %7.1f us %5.0f | + nt-store %7.1f us %5.0f\n",
               run_len, grid, a, bytes / a / 1e3, b, bytes / b / 1e3, c, bytes / c / 1e3, d, bytes / d / 1e3);
        fu.take(bytes / a / 1e3, "runs of %d, %d-byte loads %s", run_len, 8, "plain");
        fu.take(bytes / b / 1e3, "runs of %d, %d-byte loads %s", run_len, 8, "nt-store");
        fu.take(bytes / c / 1e3, "runs of %d, %d-byte loads %s", run_len, 16, "plain");
        fu.take(bytes / d / 1e3, "runs of %d, %d-byte loads %s", run_len, 16, "nt-store");
    }
    printf("\nCEILINGS on this box (GB/s, fraction of the 8000 GB/s spec peak):\n");
    printf("  float4 copy (1:1)      %5.0f  %.3f   [%s]   (guide: 6290)\n", copy.gbs, copy.gbs / 8000, copy.what);
    printf("  read only              %5.0f  %.3f   [%s]\n", rd.gbs, rd.gbs / 8000, rd.what);
    printf("  write only             %5.0f  %.3f   [%s]\n", wr.gbs, wr.gbs / 8000, wr.what);
    printf("  imdct2048 I/O (1:2)    %5.0f  %.3f   [%s]\n", im.gbs, im.gbs / 8000, im.what);
    printf("  fused synth I/O (1:1)  %5.0f  %.3f   [%s]\n", fu.gbs, fu.gbs / 8000, fu.what);
    return 0;
}
