// How the MAPPING of frames to wavefronts changes what the memory system delivers for the stereo fast path's traffic
// (round-4 review, item 3): the loads and stores of synth_dual_kernel<false, false, 0, false> on north_star's workload --
// per frame 8 KiB in (a planar packet [2][1024]: 8-byte loads, each wave-load one contiguous 512 B span; or 16-byte loads)
// and two planar PCM rows of 4 KiB out (16-byte non-temporal stores) -- with the arithmetic removed, under three mappings:
//   runs      today: a wavefront walks its own run of R consecutive frames (2048 waves, each streaming its own 256 KB)
//   wg-cyclic the 4 waves of a workgroup take consecutive frames of ONE run of 4 R frames (frame f -> wave f mod 4):
//             a workgroup streams one contiguous region (in the real kernel the overlap tail would pass wave to wave
//             through LDS)
//   grid-cyclic frame f -> wave f mod (all waves): what a block-cyclic copy does; not realisable for the fused kernel
//             (the overlap-add chains consecutive frames), the upper bound of the mapping question
// Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/io_shapes.hip -o /tmp/io_shapes && /tmp/io_shapes
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

// MODE 0 runs, 1 wg-cyclic, 2 grid-cyclic.  W16: 16-byte loads.  frames: all frames; waves = gridDim.x * 4
template <int MODE, bool W16, bool NTS>
__global__ __launch_bounds__(256, 2) void io_dual(const float *__restrict__ in, float *__restrict__ out, long frames, long ch_stride)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long n_waves = (long)gridDim.x * 4;
    const long gw = (long)blockIdx.x * 4 + wave;
    long f, step, end;
    if (MODE == 0) {
        const long R = (frames + n_waves - 1) / n_waves;
        f = gw * R; step = 1; end = std::min(frames, f + R);
    } else if (MODE == 1) {
        const long R4 = (frames + gridDim.x - 1) / gridDim.x;
        f = (long)blockIdx.x * R4 + wave; step = 4; end = std::min(frames, (long)(blockIdx.x + 1) * R4);
    } else {
        f = gw; step = n_waves; end = frames;
    }
    if (f >= end) return;
    f4v x[8];
    auto load = [&](long fr) {
        const float *p = in + fr * 2048;
        if (W16) {
#pragma unroll
            for (int m = 0; m < 8; ++m) x[m] = reinterpret_cast<const f4v *>(p)[lane + 64 * m];
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const f2v a = reinterpret_cast<const f2v *>(p)[lane + 64 * m];
                const f2v b = reinterpret_cast<const f2v *>(p)[lane + 64 * m + 512];
                x[m] = f4v{a.x, a.y, b.x, b.y};
            }
        }
    };
    load(f);
    for (; f < end; f += step) {
        f4v y[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) y[m] = x[m];
#pragma unroll
        for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(y[m]));
        load(f + step < end ? f + step : f);
        __builtin_amdgcn_sched_barrier(0);
        f4v *l4 = reinterpret_cast<f4v *>(out + f * 1024), *r4 = reinterpret_cast<f4v *>(out + ch_stride + f * 1024);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (NTS) {
                __builtin_nontemporal_store(y[2 * m], l4 + lane + 64 * m);
                __builtin_nontemporal_store(y[2 * m + 1], r4 + lane + 64 * m);
            } else {
                l4[lane + 64 * m] = y[2 * m];
                r4[lane + 64 * m] = y[2 * m + 1];
            }
        }
    }
}

template <typename F> static float time_us(F f, int reps = 15)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<float> t;
    for (int i = 0; i < 3; ++i) f();
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

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const long frames = 65536;
    float *in, *out;
    hipMalloc(&in, frames * 2048 * 4);
    hipMalloc(&out, frames * 2048 * 4 + 4096);
    hipMemset(in, 0x11, frames * 2048 * 4);
    const long ch_stride = frames * 1024;
    const double bytes = frames * 2048.0 * 8;
    printf("device: %s, %d CUs; %ld stereo frames, %.0f MiB in + %.0f MiB out per launch\n", prop.gcnArchName, cus, frames,
           frames * 8192.0 / (1 << 20), frames * 8192.0 / (1 << 20));
    const char *names[3] = {"runs (today)", "wg-cyclic", "grid-cyclic"};
    for (int rep = 0; rep < 2; ++rep)
        for (int per_cu : {2, 3, 4}) {
            const int grid = cus * per_cu;
#define ONE(MODE, W16, NTS)                                                                                                 \
    {                                                                                                                       \
        const float us = time_us([&] { hipLaunchKernelGGL((io_dual<MODE, W16, NTS>), dim3(grid), dim3(256), 0, 0, (const float *)in, out, frames, ch_stride); }); \
        printf("%-13s %2d-byte loads %-8s %d WG/CU (%5d waves): %7.1f us  %5.0f GB/s  %.3f of 8 TB/s\n", names[MODE], W16 ? 16 : 8,    \
               NTS ? "nt-store" : "plain", per_cu, grid * 4, us, bytes / us / 1e3, bytes / us / 1e3 / 8000);                   \
    }
            ONE(0, false, true) ONE(1, false, true) ONE(2, false, true)
            ONE(0, true, true) ONE(1, true, true) ONE(2, true, true)
            ONE(0, false, false) ONE(1, false, false) ONE(2, false, false)
        }
    return 0;
}
