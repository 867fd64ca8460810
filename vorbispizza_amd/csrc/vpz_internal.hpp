// Internal declarations shared by the HIP translation units of libvorbispizza_synth.so.
// gfx950 (MI355X) only -- no other back end exists or is dispatched to.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/vorbispizza_synth.h"

namespace vpz {

// ---------------------------------------------------------------------------------------------
// Device tables for one block size.
//   fast path : unit twiddles of the N/4-point complex FFT factorisation (DESIGN.md "IMDCT kernel")
//   exact path: the reference's A/B/C/bitrev tables (Mdct.cs:29-66), evaluated in the same f32 order
//   window    : rising slope of the Vorbis window (BlocksizeDerivedCache.cs:25-36), f32 order
// ---------------------------------------------------------------------------------------------
struct BlockTables {
    int n = 0;
    // fast-path twiddles, float2 each: [tw n/4][twAB 512][twBC 64] (twAB only for n == 2048)
    float2 *d_fast = nullptr;
    // exact-path tables
    float *d_A = nullptr, *d_B = nullptr, *d_C = nullptr;
    uint16_t *d_bitrev = nullptr;
    int ld = 0;
    // window slope, n/2 floats
    float *d_slope = nullptr;
    std::vector<float> h_slope;
};

struct Context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    int num_cu = 256;
    std::string last_error;
    std::map<int, BlockTables> tables;  // keyed by block size (Mdct._setupCache analogue)
    float *d_inv_db = nullptr;          // 256-entry floor1 inverse dB table (Floor1.cs:407-473)
    // staging buffers for VPZ_MEM_HOST calls (grown on demand, reused)
    void *stage_in = nullptr;  size_t stage_in_bytes = 0;
    void *stage_out = nullptr; size_t stage_out_bytes = 0;
    void *stage_aux = nullptr; size_t stage_aux_bytes = 0;
    // fork-join pool of the host-side state machine (host_pool.hpp), created by the first large synth batch
    void *host_pool = nullptr;
    void (*host_pool_free)(void *) = nullptr;
    // persistent grids of the IMDCT kernels on THIS context's device (workgroups the chip keeps resident), filled by the first
    // launch of each; per context, not per process: a host with several GPUs holds one context per device
    int resident[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
enum { kResident2048 = 0, kResident256, kResident4096, kResident8192, kResident512, kResident1024 };

int set_error(Context *ctx, int status, const char *what, hipError_t e = hipSuccess);
int get_tables(Context *ctx, int n, BlockTables **out);
int ensure_stage(Context *ctx, void **buf, size_t *have, size_t need);

#define VPZ_HIP_TRY(ctx, expr)                                                   \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) return vpz::set_error((ctx), VPZ_E_HIP, #expr, _e); \
    } while (0)

// ---- kernel launchers (imdct_fast.hip / imdct_exact.hip / synth_kernels.hip) ----
// spectra [count][n/2] -> out [count][n], device pointers, asynchronous on `stream`.
// optional gather lists (float offsets of each block's spectrum / output) serve the three-pass decoder path
hipError_t launch_imdct_fast_2048(const float *spectra, float *out, int64_t count,
                                  const float2 *tw, Context *ctx, hipStream_t stream,
                                  const int64_t *src_off = nullptr, const int64_t *dst_off = nullptr);
hipError_t launch_imdct_fast_256(const float *spectra, float *out, int64_t count,
                                 const float2 *tw, Context *ctx, hipStream_t stream,
                                 const int64_t *src_off = nullptr, const int64_t *dst_off = nullptr);
// N = 4096; optional gather lists (float offsets of each block's spectrum / output) for the three-pass decoder path
hipError_t launch_imdct_fast_4096(const float *spectra, float *out, int64_t count, const float2 *tw, Context *ctx,
                                  hipStream_t stream, const int64_t *src_off = nullptr, const int64_t *dst_off = nullptr);
hipError_t launch_imdct_fast_8192(const float *spectra, float *out, int64_t count, const float2 *tw, Context *ctx,
                                  hipStream_t stream, const int64_t *src_off = nullptr, const int64_t *dst_off = nullptr);
hipError_t launch_imdct_fast_mid(int n, const float *spectra, float *out, int64_t count, const float2 *tw, Context *ctx,
                                 hipStream_t stream, const int64_t *src_off = nullptr,
                                 const int64_t *dst_off = nullptr);  // n = 512 or 1024
hipError_t launch_imdct_exact(int n, int ld, const float *spectra, float *out, int64_t count,
                              const float *A, const float *B, const float *C,
                              const uint16_t *bitrev, int num_cu, hipStream_t stream,
                              const int64_t *src_off = nullptr, const int64_t *dst_off = nullptr);

// offsets (in float2 units) inside BlockTables::d_fast
constexpr int kFastTwOffset = 0;       // tw[k] = exp(+2*pi*i*(k + 1/8)/n), k < n/4   (<= 512 entries)
constexpr int kFastTwABOffset = 512;   // twAB[p*64 + l] = exp(+2*pi*i*l*p/512)
constexpr int kFastTwBCOffset = 1024;  // twBC[l0*8 + q] = exp(+2*pi*i*l0*q/64)
constexpr int kFastTableCount = 1024 + 64;
// N = 4096 keeps its own layout: tw[1024] | twAB[512] | twBC[64] | w[512] = exp(2*pi*i*j/1024)
constexpr int kFast4096TwOffset = 0, kFast4096TwABOffset = 1024, kFast4096TwBCOffset = 1536, kFast4096WOffset = 1600;
constexpr int kFast4096TableCount = 2112;
// N = 8192: tw[2048] | twAB[512] | twBC[64] | w1[512] = exp(2*pi*i*j/1024) | w2[1024] = exp(2*pi*i*J/2048)
constexpr int kFast8192TwOffset = 0, kFast8192TwABOffset = 2048, kFast8192TwBCOffset = 2560, kFast8192W1Offset = 2624,
              kFast8192W2Offset = 3136;
constexpr int kFast8192TableCount = 4160;

}  // namespace vpz

struct vpz_context {
    vpz::Context impl;
};
