// Experiment (not product code): does ONE wavefront that synthesises BOTH channels of a stereo stream -- two
// independent transforms per iteration, 8 waves per CU -- beat two wavefronts with one channel each (16 waves per CU)?
// All-long N = 2048 stereo IMDCT + window + OLA, planar output: north_star's literal workload, stripped to its
// steady state.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -I vorbispizza_amd/csrc tools/dual_proto.hip -o /tmp/dual_proto && /tmp/dual_proto
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#include "imdct_core.hpp"
using namespace vpz;

constexpr int kRow = 1160, kTail = 512, kRun = 32;

__device__ __forceinline__ float ola2(float v, float wl, float t, float wr)
{
#pragma clang fp contract(off)
    const float a = v * wl;
    const float b = t * wr;
    return a + b;
}

// long after long: out[0..512) = -mirror(h[0..512)) over the straight tail, out[512..1024) = h[0..512) over the mirrored tail
__device__ __forceinline__ void emit_long(const float *h, const float *tail, const float *slope, float *dst, int lane)
{
    const float4 *h4 = reinterpret_cast<const float4 *>(h), *t4 = reinterpret_cast<const float4 *>(tail);
    const float4 *s4 = reinterpret_cast<const float4 *>(slope);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int g = lane + 64 * r;
        const float4 wl = s4[g], wr = s4[255 - g];
        float4 o;
        if (r < 2) {
            const float4 hv = h4[127 - g], pv = t4[g];
            o = make_float4(ola2(-hv.w, wl.x, pv.x, wr.w), ola2(-hv.z, wl.y, pv.y, wr.z), ola2(-hv.y, wl.z, pv.z, wr.y), ola2(-hv.x, wl.w, pv.w, wr.x));
        } else {
            const float4 hv = h4[g - 128], pv = t4[255 - g];
            o = make_float4(ola2(hv.x, wl.x, pv.w, wr.w), ola2(hv.y, wl.y, pv.z, wr.z), ola2(hv.z, wl.z, pv.y, wr.y), ola2(hv.w, wl.w, pv.x, wr.x));
        }
        store_nt(reinterpret_cast<float4 *>(dst) + g, o);
    }
}
__device__ __forceinline__ void save_tail(const float *h, float *tail, int lane)
{
    const float4 *src = reinterpret_cast<const float4 *>(h + 512);
    float4 *dt = reinterpret_cast<float4 *>(tail);
    const float4 t0 = src[lane], t1 = src[lane + 64];
    dt[lane] = t0;
    dt[lane + 64] = t1;
}

// CH = channels per wavefront (1 or 2); workgroup = 8 / CH waves, all 8 rows of LDS in use either way
template <int CH>
__global__ __launch_bounds__(64 * (8 / CH), 4 / CH) void synth_proto(const float *__restrict__ spec, float *__restrict__ out,
                                                                    const float2 *__restrict__ tw, const float *__restrict__ slope_g,
                                                                    int n_runs, long frames)
{
    constexpr int kWaves = 8 / CH, kThreads = 64 * kWaves;
    __shared__ float2 s_tw[512], s_twAB[512], s_twBC[64];
    __shared__ float s_slope[1024];
    __shared__ float s_work[8][kRow], s_tail[8][kTail];
    for (int i = threadIdx.x; i < 512; i += kThreads) { s_tw[i] = tw[i]; s_twAB[i] = tw[512 + i]; }
    if (threadIdx.x < 64) s_twBC[threadIdx.x] = tw[1024 + threadIdx.x];
    for (int i = threadIdx.x; i < 1024; i += kThreads) s_slope[i] = slope_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * kWaves + wave;
    const int run = CH == 2 ? item : item >> 1, ch0 = CH == 2 ? 0 : item & 1;
    if (run >= n_runs) return;
    const long f0 = (long)run * kRun;
    float2 x[CH][8], xn[CH][8];
    auto load = [&](long f, float2 (&dst)[CH][8]) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float2 *s = reinterpret_cast<const float2 *>(spec + (f * 2 + ch0 + c) * 1024);
#pragma unroll
            for (int m = 0; m < 8; ++m) dst[c][m] = s[lane + 64 * m];
        }
    };
    load(f0, x);
    for (int i = 0; i < kRun; ++i) {
        if (i + 1 < kRun) load(f0 + i + 1, xn);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            imdct2048_wave(x[c], reinterpret_cast<float2 *>(s_work[wave * CH + c]), s_tw, s_twAB, s_twBC, lane);
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(xn[c][m].x), "v"(xn[c][m].y));
        if (i > 0) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                emit_long(s_work[wave * CH + c], s_tail[wave * CH + c], s_slope,
                          out + (long)(ch0 + c) * frames * 1024 + (f0 + i - 1) * 1024, lane);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) save_tail(s_work[wave * CH + c], s_tail[wave * CH + c], lane);
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int m = 0; m < 8; ++m) x[c][m] = xn[c][m];
    }
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
    const long frames = 65536;
    const int n_runs = frames / kRun;
    float *spec, *out, *slope; float2 *tw;
    hipMalloc(&spec, frames * 2 * 1024 * 4); hipMalloc(&out, frames * 2 * 1024 * 4 + 4096);
    hipMalloc(&tw, 1088 * 8); hipMalloc(&slope, 4096);
    std::vector<float> hs(frames * 2 * 1024);
    for (size_t i = 0; i < hs.size(); ++i) hs[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f / 256.0f;
    hipMemcpy(spec, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    std::vector<float2> htw(1088);
    const double two_pi = 6.283185307179586;
    for (int k = 0; k < 512; ++k) htw[k] = make_float2((float)cos(two_pi * (k + 0.125) / 2048), (float)sin(two_pi * (k + 0.125) / 2048));
    for (int p = 0; p < 8; ++p) for (int l = 0; l < 64; ++l) htw[512 + p * 64 + l] = make_float2((float)cos(two_pi * l * p / 512), (float)sin(two_pi * l * p / 512));
    for (int l0 = 0; l0 < 8; ++l0) for (int q = 0; q < 8; ++q) htw[1024 + l0 * 8 + q] = make_float2((float)cos(two_pi * l0 * q / 64), (float)sin(two_pi * l0 * q / 64));
    hipMemcpy(tw, htw.data(), 1088 * 8, hipMemcpyHostToDevice);
    std::vector<float> hsl(1024);
    for (int i = 0; i < 1024; ++i) { double s = sin(0.5 * M_PI * (i + 0.5) / 1024); hsl[i] = (float)sin(0.5 * M_PI * s * s); }
    hipMemcpy(slope, hsl.data(), 4096, hipMemcpyHostToDevice);
    const double bytes = (double)frames * 2 * 1024 * 4 * 2;
    float us1 = time_us([&] { hipLaunchKernelGGL(synth_proto<1>, dim3((n_runs * 2 + 7) / 8), dim3(512), 0, 0, spec, out, tw, slope, n_runs, frames); });
    std::vector<float> o1(1 << 20), o2(1 << 20);
    hipMemcpy(o1.data(), out + 5 * (1 << 20), 4 << 20, hipMemcpyDeviceToHost);
    float us2 = time_us([&] { hipLaunchKernelGGL(synth_proto<2>, dim3((n_runs + 3) / 4), dim3(256), 0, 0, spec, out, tw, slope, n_runs, frames); });
    hipMemcpy(o2.data(), out + 5 * (1 << 20), 4 << 20, hipMemcpyDeviceToHost);
    double diff = 0; for (size_t i = 0; i < o1.size(); ++i) diff = std::max(diff, (double)fabs(o1[i] - o2[i]));
    printf("one channel per wave (16 waves/CU): %.1f us  %.0f GB/s  frac %.3f\n", us1, bytes / us1 / 1e3, bytes / us1 / 1e3 / 8000);
    printf("two channels per wave ( 8 waves/CU): %.1f us  %.0f GB/s  frac %.3f   (outputs differ by %g)\n", us2, bytes / us2 / 1e3, bytes / us2 / 1e3 / 8000, diff);
    return 0;
}
