// Measures what HBM delivers for the I/O shape of the IMDCT kernel with the arithmetic removed:
// per wavefront and block, 8 x 512 B loads (8 B/lane) and 8 x 1 KiB stores (16 B/lane), 1:2 read:write.
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/io_ceiling.hip -o /tmp/io_ceiling && /tmp/io_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(256) void io_shape(const float *__restrict__ in, float *__restrict__ out, long count)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long stride = (long)gridDim.x * 4;
    for (long blk = (long)blockIdx.x * 4 + wave; blk < count; blk += stride) {
        const float2 *src = reinterpret_cast<const float2 *>(in + blk * 1024);
        float2 x[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (VARIANT & 2) { f2v t = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(&src[lane + 64 * m])); x[m] = make_float2(t.x, t.y); }
            else x[m] = src[lane + 64 * m];
        }
        float4 *o4 = reinterpret_cast<float4 *>(out + blk * 2048);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 v = make_float4(x[2 * r].x, x[2 * r].y, x[2 * r + 1].x, x[2 * r + 1].y);
            float4 w = make_float4(-v.w, -v.z, -v.y, -v.x);
            if (VARIANT & 1) {
                { f4v t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(&o4[lane + 64 * r])); }
                { f4v t = {w.x, w.y, w.z, w.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(&o4[256 + lane + 64 * r])); }
            } else {
                o4[lane + 64 * r] = v;
                o4[256 + lane + 64 * r] = w;
            }
        }
    }
}

// the same traffic with the NEXT block's loads issued before this block's stores (what imdct2048_kernel does since
// round 2): `depth` blocks of loads in flight per wavefront
template <int DEPTH>
__global__ __launch_bounds__(256) void io_pipelined(const float *__restrict__ in, float *__restrict__ out, long count)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long stride = (long)gridDim.x * 4;
    long blk = (long)blockIdx.x * 4 + wave;
    if (blk >= count) return;
    float2 x[DEPTH][8];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const long b = blk + d * stride < count ? blk + d * stride : blk;
        const float2 *src = reinterpret_cast<const float2 *>(in + b * 1024);
#pragma unroll
        for (int m = 0; m < 8; ++m) x[d][m] = src[lane + 64 * m];
    }
    while (true) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long cur = blk + d * stride;
            if (cur >= count) return;
            float2 y[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) y[m] = x[d][m];
            const long nb = cur + DEPTH * stride < count ? cur + DEPTH * stride : cur;
            const float2 *src = reinterpret_cast<const float2 *>(in + nb * 1024);
#pragma unroll
            for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(y[m].x), "v"(y[m].y));
#pragma unroll
            for (int m = 0; m < 8; ++m) x[d][m] = src[lane + 64 * m];
            __builtin_amdgcn_sched_barrier(0);
            float4 *o4 = reinterpret_cast<float4 *>(out + cur * 2048);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float4 v = make_float4(y[2 * r].x, y[2 * r].y, y[2 * r + 1].x, y[2 * r + 1].y);
                { f4v t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(&o4[lane + 64 * r])); }
                { f4v t = {-v.w, -v.z, -v.y, -v.x}; __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(&o4[256 + lane + 64 * r])); }
            }
        }
        blk += DEPTH * stride;
    }
}

__global__ __launch_bounds__(256) void copy4(const float4 *__restrict__ in, float4 *__restrict__ out, long n4)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) out[i] = in[i];
}

template <typename F> static float time_us(F f, int reps = 15)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> t;
    for (int i = 0; i < 3; ++i) f();
    for (int i = 0; i < reps; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms * 1e3f); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const long count = 131072;
    float *in, *out;
    hipMalloc(&in, count * 1024 * 4); hipMalloc(&out, count * 2048 * 4);
    hipMemset(in, 0x11, count * 1024 * 4);
    for (int per_cu : {1, 2, 3, 4}) {
        int grid = 256 * per_cu;
        float us0 = time_us([&] { hipLaunchKernelGGL(io_shape<0>, dim3(grid), dim3(256), 0, 0, in, out, count); });
        float us1 = time_us([&] { hipLaunchKernelGGL(io_shape<1>, dim3(grid), dim3(256), 0, 0, in, out, count); });
        float us2 = time_us([&] { hipLaunchKernelGGL(io_shape<2>, dim3(grid), dim3(256), 0, 0, in, out, count); });
        float us3 = time_us([&] { hipLaunchKernelGGL(io_shape<3>, dim3(grid), dim3(256), 0, 0, in, out, count); });
        printf("io_shape  %d WG/CU: plain %.1f us %.0f GB/s | nt-store %.1f us %.0f | nt-load %.1f us %.0f | both %.1f us %.0f\n", per_cu,
               us0, count * 12288.0 / us0 / 1e3, us1, count * 12288.0 / us1 / 1e3, us2, count * 12288.0 / us2 / 1e3, us3, count * 12288.0 / us3 / 1e3);
    }
    for (int per_cu : {1, 2, 3, 4}) {
        int grid = 256 * per_cu;
        float us1 = time_us([&] { hipLaunchKernelGGL(io_pipelined<1>, dim3(grid), dim3(256), 0, 0, in, out, count); });
        float us2 = time_us([&] { hipLaunchKernelGGL(io_pipelined<2>, dim3(grid), dim3(256), 0, 0, in, out, count); });
        printf("pipelined %d WG/CU: 1 block ahead %.1f us %.0f GB/s | 2 blocks ahead %.1f us %.0f GB/s\n", per_cu, us1,
               count * 12288.0 / us1 / 1e3, us2, count * 12288.0 / us2 / 1e3);
    }
    for (int per_cu : {2, 4, 8}) {
        int grid = 256 * per_cu;
        long n4 = count * 1024 / 4;  // copy 512 MiB -> 512 MiB
        float us = time_us([&] { hipLaunchKernelGGL(copy4, dim3(grid), dim3(256), 0, 0, (const float4 *)in, (float4 *)out, n4); });
        printf("copy4     %d WG/CU: %.1f us  %.0f GB/s\n", per_cu, us, n4 * 32.0 / us / 1e3);
    }
    return 0;
}
