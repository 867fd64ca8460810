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

// MODE 0 runs, 1 wg-cyclic, 2 grid-cyclic, 3 chunked: chunks of `chunk` consecutive frames, chunk c -> workgroup c mod grid,
// inside a chunk frame f -> wave f mod W (what a kernel with an LDS hand-off of the overlap tail and one recomputed block per
// chunk could do).  W16: 16-byte loads.  W = waves per workgroup = blockDim.x / 64
template <int MODE, bool W16, bool NTS>
__global__ __launch_bounds__(512) void io_dual(const float *__restrict__ in, float *__restrict__ out, long frames, long ch_stride, int chunk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = blockDim.x >> 6;
    const long n_waves = (long)gridDim.x * W;
    const long gw = (long)blockIdx.x * W + wave;
    long f, step, end;
    if (MODE == 3) {
        // the frame sequence of this wave: chunk index c = blockIdx.x + k * gridDim.x, frames c * chunk + wave + W * j
        const long n_chunks = (frames + chunk - 1) / chunk;
        auto frame_of = [&](long i) -> long {  // i-th frame of this wave, -1 past the end
            const int per = chunk / W;         // (chunk is a multiple of W)
            const long c = blockIdx.x + (i / per) * (long)gridDim.x;
            if (c >= n_chunks) return -1;
            const long fr = c * chunk + wave + (long)W * (i % per);
            return fr < frames ? fr : -1;
        };
        long i = 0;
        long fr = frame_of(0);
        if (fr < 0) return;
        f4v x[8];
        auto load = [&](long q) {
            const float *p = in + q * 2048;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (W16) x[m] = reinterpret_cast<const f4v *>(p)[lane + 64 * m];
                else {
                    const f2v a = reinterpret_cast<const f2v *>(p)[lane + 64 * m];
                    const f2v b = reinterpret_cast<const f2v *>(p)[lane + 64 * m + 512];
                    x[m] = f4v{a.x, a.y, b.x, b.y};
                }
            }
        };
        load(fr);
        while (fr >= 0) {
            f4v y[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) y[m] = x[m];
#pragma unroll
            for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(y[m]));
            const long nx = frame_of(++i);
            load(nx >= 0 ? nx : fr);
            __builtin_amdgcn_sched_barrier(0);
            f4v *l4 = reinterpret_cast<f4v *>(out + fr * 1024), *r4 = reinterpret_cast<f4v *>(out + ch_stride + fr * 1024);
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
            fr = nx;
        }
        return;
    }
    if (MODE == 0 && chunk > 0) {
        // runs of `chunk` frames, run r -> wave r mod (all waves): a wave walks its runs one after the other
        f4v x[8];
        for (long r = gw; r * chunk < frames; r += n_waves) {
            const long lo = r * chunk, hi = std::min(frames, lo + chunk);
            auto load = [&](long q) {
                const float *p = in + q * 2048;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const f2v a = reinterpret_cast<const f2v *>(p)[lane + 64 * m];
                    const f2v b = reinterpret_cast<const f2v *>(p)[lane + 64 * m + 512];
                    x[m] = f4v{a.x, a.y, b.x, b.y};
                }
            };
            load(lo);
            for (long q = lo; q < hi; ++q) {
                f4v y[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) y[m] = x[m];
#pragma unroll
                for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(y[m]));
                load(q + 1 < hi ? q + 1 : q);
                __builtin_amdgcn_sched_barrier(0);
                f4v *l4 = reinterpret_cast<f4v *>(out + q * 1024), *r4 = reinterpret_cast<f4v *>(out + ch_stride + q * 1024);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    __builtin_nontemporal_store(y[2 * m], l4 + lane + 64 * m);
                    __builtin_nontemporal_store(y[2 * m + 1], r4 + lane + 64 * m);
                }
            }
        }
        return;
    }
    if (MODE == 4) {
        // today's runs (R = frames / waves), every wave starting at its own phase inside its run and wrapping round
        const long R = (frames + n_waves - 1) / n_waves;
        const long lo = gw * R, hi = std::min(frames, lo + R);
        if (lo >= hi) return;
        const long n = hi - lo;
        const long s0 = (gw * chunk) % n;  // chunk: the phase step between neighbouring waves
        f4v x[8];
        auto load = [&](long q) {
            const float *p = in + q * 2048;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const f2v a = reinterpret_cast<const f2v *>(p)[lane + 64 * m];
                const f2v b = reinterpret_cast<const f2v *>(p)[lane + 64 * m + 512];
                x[m] = f4v{a.x, a.y, b.x, b.y};
            }
        };
        load(lo + s0);
        for (long i = 0; i < n; ++i) {
            const long q = lo + (s0 + i) % n, qn = lo + (s0 + i + 1) % n;
            f4v y[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) y[m] = x[m];
#pragma unroll
            for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(y[m]));
            load(qn);
            __builtin_amdgcn_sched_barrier(0);
            f4v *l4 = reinterpret_cast<f4v *>(out + q * 1024), *r4 = reinterpret_cast<f4v *>(out + ch_stride + q * 1024);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                __builtin_nontemporal_store(y[2 * m], l4 + lane + 64 * m);
                __builtin_nontemporal_store(y[2 * m + 1], r4 + lane + 64 * m);
            }
        }
        return;
    }
    if (MODE == 0) {
        const long R = (frames + n_waves - 1) / n_waves;
        f = gw * R; step = 1; end = std::min(frames, f + R);
    } else if (MODE == 1) {
        const long R4 = (frames + gridDim.x - 1) / gridDim.x;
        f = (long)blockIdx.x * R4 + wave; step = W; end = std::min(frames, (long)(blockIdx.x + 1) * R4);
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
    const char *names[5] = {"runs (today)", "wg-cyclic", "grid-cyclic", "chunked", "runs, phased"};
    for (int rep = 0; rep < 2; ++rep) {
        for (int per_cu : {2, 4}) {
            const int grid = cus * per_cu;
#define ONE(MODE, W16, NTS, THREADS, CHUNK)                                                                                 \
    {                                                                                                                       \
        const float us = time_us([&] { hipLaunchKernelGGL((io_dual<MODE, W16, NTS>), dim3(grid), dim3(THREADS), 0, 0, (const float *)in, out, frames, ch_stride, CHUNK); }); \
        printf("%-13s chunk %4d %2d-byte loads %-8s %d WG/CU x %d waves (%5d waves): %7.1f us  %5.0f GB/s  %.3f of 8 TB/s\n", names[MODE], CHUNK, W16 ? 16 : 8, \
               NTS ? "nt-store" : "plain", per_cu, THREADS / 64, grid * (THREADS / 64), us, bytes / us / 1e3, bytes / us / 1e3 / 8000);  \
    }
            ONE(0, false, true, 256, 0) ONE(1, false, true, 256, 0) ONE(2, false, true, 256, 0)
            ONE(0, true, true, 256, 0) ONE(1, true, true, 256, 0) ONE(2, true, true, 256, 0)
            ONE(3, false, true, 256, 4) ONE(3, false, true, 256, 8) ONE(3, false, true, 256, 16) ONE(3, false, true, 256, 32)
            ONE(3, false, true, 256, 64) ONE(3, false, true, 256, 128)
            ONE(3, true, true, 256, 8) ONE(3, true, true, 256, 16) ONE(3, true, true, 256, 32)
        }
        {   // today's launch shape: run lengths other than frames / waves (run r -> wave r mod waves), and phased starts
            const int per_cu = 2;
            const int grid = cus * per_cu;
            ONE(0, false, true, 256, 8) ONE(0, false, true, 256, 15) ONE(0, false, true, 256, 16) ONE(0, false, true, 256, 17)
            ONE(0, false, true, 256, 24) ONE(0, false, true, 256, 28) ONE(0, false, true, 256, 30) ONE(0, false, true, 256, 31)
            ONE(0, false, true, 256, 32) ONE(0, false, true, 256, 33) ONE(0, false, true, 256, 34) ONE(0, false, true, 256, 36)
            ONE(0, false, true, 256, 40) ONE(0, false, true, 256, 48) ONE(0, false, true, 256, 63)
            ONE(4, false, true, 256, 1) ONE(4, false, true, 256, 3) ONE(4, false, true, 256, 5) ONE(4, false, true, 256, 7)
            ONE(4, false, true, 256, 11) ONE(4, false, true, 256, 13)
        }
        {   // one workgroup of 8 waves per CU
            const int per_cu = 1;
            const int grid = cus;
            ONE(0, false, true, 512, 0) ONE(1, false, true, 512, 0) ONE(2, false, true, 512, 0)
            ONE(3, false, true, 512, 8) ONE(3, false, true, 512, 16) ONE(3, false, true, 512, 32) ONE(3, false, true, 512, 64)
            ONE(3, false, true, 512, 256)
        }
    }
    return 0;
}
