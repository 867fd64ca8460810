// Context, per-block-size tables and vpz_imdct_batch of the C ABI (include/vorbispizza_synth.h).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

#include "vpz_internal.hpp"

namespace vpz {

int set_error(Context *ctx, int status, const char *what, hipError_t e)
{
    if (ctx) {
        char buf[512];
        if (e != hipSuccess)
            snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
        else
            snprintf(buf, sizeof buf, "%s", what);
        ctx->last_error = buf;
    }
    return status;
}

int ensure_stage(Context *ctx, void **buf, size_t *have, size_t need)
{
    if (*have >= need) return VPZ_OK;
    if (*buf) {
        VPZ_HIP_TRY(ctx, hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    size_t want = need + need / 4;
    hipError_t e = hipMalloc(buf, want);
    if (e != hipSuccess) return set_error(ctx, VPZ_E_NOMEM, "hipMalloc(staging)", e);
    *have = want;
    return VPZ_OK;
}

static int ilog(int x)  // Utils.cs:19-28
{
    int cnt = 0;
    while (x > 0) { ++cnt; x >>= 1; }
    return cnt;
}

static uint32_t bit_reverse(uint32_t v, int bits)  // Utils.cs:35-42
{
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}

template <typename T>
static int upload(Context *ctx, T **dst, const std::vector<T> &src)
{
    size_t bytes = sizeof(T) * (src.empty() ? 1 : src.size());
    hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), bytes);
    if (e != hipSuccess) return set_error(ctx, VPZ_E_NOMEM, "hipMalloc(table)", e);
    if (!src.empty()) VPZ_HIP_TRY(ctx, hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
    return VPZ_OK;
}

int get_tables(Context *ctx, int n, BlockTables **out)
{
    auto it = ctx->tables.find(n);
    if (it != ctx->tables.end()) {
        *out = &it->second;
        return VPZ_OK;
    }
    if (n < 64 || n > 8192 || (n & (n - 1)) != 0)
        return set_error(ctx, VPZ_E_UNSUPPORTED, "block size must be a power of two in [64, 8192]");

    BlockTables t;
    t.n = n;
    t.ld = ilog(n) - 1;
    const int n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
    const float pi_f = 3.14159274101257324219f;  // MathF.PI

    // reference twiddles, same f32 evaluation order as Mdct.cs:43-58
    std::vector<float> A(n2), B(n2), C(n4);
    for (int k = 0, k2 = 0; k < n4; ++k, k2 += 2) {
        float arg_a = (float)(4 * k) * pi_f / (float)n;
        A[k2] = cosf(arg_a);
        A[k2 + 1] = -sinf(arg_a);
        float arg_b = (float)(k2 + 1) * pi_f / (float)n / 2.0f;
        B[k2] = cosf(arg_b) * .5f;
        B[k2 + 1] = sinf(arg_b) * .5f;
    }
    for (int k = 0, k2 = 0; k < n8; ++k, k2 += 2) {
        float arg_c = (float)(2 * (k2 + 1)) * pi_f / (float)n;
        C[k2] = cosf(arg_c);
        C[k2 + 1] = -sinf(arg_c);
    }
    std::vector<uint16_t> rev(n8);
    for (int i = 0; i < n8; ++i) rev[i] = (uint16_t)(bit_reverse((uint32_t)i, t.ld - 3) << 2);

    // window slope, f32 order of BlocksizeDerivedCache.cs:34-35
    t.h_slope.resize(n2);
    for (int x = 0; x < n2; ++x) {
        float v = sinf(0.5f * pi_f * ((float)x + 0.5f) / (float)n2);
        t.h_slope[x] = sinf(0.5f * pi_f * v * v);
    }

    // fast-path unit twiddles, evaluated in double and rounded once
    std::vector<float2> fast;
    if (n == 2048 || n == 256) {
        fast.assign(kFastTableCount, make_float2(0.f, 0.f));
        const double two_pi = 6.283185307179586476925286766559;
        for (int k = 0; k < n4; ++k) {
            double a = two_pi * ((double)k + 0.125) / (double)n;
            fast[kFastTwOffset + k] = make_float2((float)cos(a), (float)sin(a));
        }
        for (int p = 0; p < 8; ++p)
            for (int l = 0; l < 64; ++l) {
                double a = two_pi * (double)(l * p) / 512.0;
                fast[kFastTwABOffset + p * 64 + l] = make_float2((float)cos(a), (float)sin(a));
            }
        for (int l0 = 0; l0 < 8; ++l0)
            for (int q = 0; q < 8; ++q) {
                double a = two_pi * (double)(l0 * q) / 64.0;
                fast[kFastTwBCOffset + l0 * 8 + q] = make_float2((float)cos(a), (float)sin(a));
            }
    }

    if (n == 4096) {
        // imdct4096_wave: tw[1024] | the 512-point stage tables | w[j] = exp(2*pi*i*j/1024)
        fast.assign(kFast4096TableCount, make_float2(0.f, 0.f));
        const double two_pi = 6.283185307179586476925286766559;
        for (int k = 0; k < 1024; ++k) {
            double a = two_pi * ((double)k + 0.125) / 4096.0;
            fast[kFast4096TwOffset + k] = make_float2((float)cos(a), (float)sin(a));
        }
        for (int p = 0; p < 8; ++p)
            for (int l = 0; l < 64; ++l) {
                double a = two_pi * (double)(l * p) / 512.0;
                fast[kFast4096TwABOffset + p * 64 + l] = make_float2((float)cos(a), (float)sin(a));
            }
        for (int l0 = 0; l0 < 8; ++l0)
            for (int q = 0; q < 8; ++q) {
                double a = two_pi * (double)(l0 * q) / 64.0;
                fast[kFast4096TwBCOffset + l0 * 8 + q] = make_float2((float)cos(a), (float)sin(a));
            }
        for (int j = 0; j < 512; ++j) {
            double a = two_pi * (double)j / 1024.0;
            fast[kFast4096WOffset + j] = make_float2((float)cos(a), (float)sin(a));
        }
    }
    if (n == 8192) {
        fast.assign(kFast8192TableCount, make_float2(0.f, 0.f));
        const double two_pi = 6.283185307179586476925286766559;
        for (int k = 0; k < 2048; ++k) {
            double a = two_pi * ((double)k + 0.125) / 8192.0;
            fast[kFast8192TwOffset + k] = make_float2((float)cos(a), (float)sin(a));
        }
        for (int p = 0; p < 8; ++p)
            for (int l = 0; l < 64; ++l) {
                double a = two_pi * (double)(l * p) / 512.0;
                fast[kFast8192TwABOffset + p * 64 + l] = make_float2((float)cos(a), (float)sin(a));
            }
        for (int l0 = 0; l0 < 8; ++l0)
            for (int q = 0; q < 8; ++q) {
                double a = two_pi * (double)(l0 * q) / 64.0;
                fast[kFast8192TwBCOffset + l0 * 8 + q] = make_float2((float)cos(a), (float)sin(a));
            }
        for (int j = 0; j < 512; ++j) {
            double a = two_pi * (double)j / 1024.0;
            fast[kFast8192W1Offset + j] = make_float2((float)cos(a), (float)sin(a));
        }
        // (the upper half is the lower one times i, to the bit: imdct8192_wave reads the lower half only and turns the value)
        for (int j = 0; j < 512; ++j) {
            double a = two_pi * (double)j / 2048.0;
            const float2 w = make_float2((float)cos(a), (float)sin(a));
            fast[kFast8192W2Offset + j] = w;
            fast[kFast8192W2Offset + 512 + j] = make_float2(-w.y, w.x);
        }
    }
    if (n == 512 || n == 1024) {
        // imdct_mid_wave<R>: tw[k], twAB[p*L + l] = exp(2*pi*i*p*l/(64R)), twBC[l] = exp(2*pi*i*rev(l1)*l0/(8R))
        const int R = n / 256, L = 8 * R, M = 64 * R;
        fast.assign(kFastTableCount, make_float2(0.f, 0.f));
        const double two_pi = 6.283185307179586476925286766559;
        for (int k = 0; k < M; ++k) {
            double a = two_pi * ((double)k + 0.125) / (double)n;
            fast[kFastTwOffset + k] = make_float2((float)cos(a), (float)sin(a));
        }
        for (int p = 0; p < 8; ++p)
            for (int l = 0; l < L; ++l) {
                double a = two_pi * (double)(p * l) / (double)M;
                fast[kFastTwABOffset + p * L + l] = make_float2((float)cos(a), (float)sin(a));
            }
        for (int l = 0; l < L; ++l) {
            const int l0 = l & 7, l1 = l >> 3;
            const int q1 = R == 2 ? l1 : (((l1 & 1) << 1) | (l1 >> 1));
            double a = two_pi * (double)(q1 * l0) / (double)L;
            fast[kFastTwBCOffset + l] = make_float2((float)cos(a), (float)sin(a));
        }
    }

    int rc;
    if ((rc = upload(ctx, &t.d_A, A)) != VPZ_OK) return rc;
    if ((rc = upload(ctx, &t.d_B, B)) != VPZ_OK) return rc;
    if ((rc = upload(ctx, &t.d_C, C)) != VPZ_OK) return rc;
    if ((rc = upload(ctx, &t.d_bitrev, rev)) != VPZ_OK) return rc;
    if ((rc = upload(ctx, &t.d_slope, t.h_slope)) != VPZ_OK) return rc;
    if (!fast.empty() && (rc = upload(ctx, &t.d_fast, fast)) != VPZ_OK) return rc;

    auto ins = ctx->tables.emplace(n, std::move(t));
    *out = &ins.first->second;
    return VPZ_OK;
}

static void free_tables(BlockTables &t)
{
    if (t.d_fast) (void)hipFree(t.d_fast);
    if (t.d_A) (void)hipFree(t.d_A);
    if (t.d_B) (void)hipFree(t.d_B);
    if (t.d_C) (void)hipFree(t.d_C);
    if (t.d_bitrev) (void)hipFree(t.d_bitrev);
    if (t.d_slope) (void)hipFree(t.d_slope);
}

}  // namespace vpz

using vpz::Context;

extern "C" {

int vpz_abi_version(void) { return VPZ_ABI_VERSION; }

const char *vpz_error_string(int status)
{
    switch (status) {
    case VPZ_OK: return "ok";
    case VPZ_E_INVALID_ARG: return "invalid argument";
    case VPZ_E_UNSUPPORTED: return "unsupported block size or layout";
    case VPZ_E_HIP: return "HIP runtime error";
    case VPZ_E_NOMEM: return "out of device memory";
    case VPZ_E_WINDOW_MISMATCH: return "previous packet tail longer than the window slope";
    case VPZ_E_NO_DEVICE: return "no usable gfx950 device";
    case VPZ_E_CAPACITY: return "output buffer too small";
    default: return "unknown status";
    }
}

int vpz_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int vpz_context_create(int device_id, vpz_context **out)
{
    if (!out) return VPZ_E_INVALID_ARG;
    *out = nullptr;
    int n = vpz_device_count();
    if (n <= 0) return VPZ_E_NO_DEVICE;
    if (device_id < 0 || device_id >= n) return VPZ_E_INVALID_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return VPZ_E_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return VPZ_E_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VPZ_E_NO_DEVICE;  // MI355X only

    vpz_context *c = new (std::nothrow) vpz_context();
    if (!c) return VPZ_E_NOMEM;
    Context *ctx = &c->impl;
    ctx->device = device_id;
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess) {
        vpz_context_destroy(c);
        return VPZ_E_HIP;
    }
    *out = c;
    return VPZ_OK;
}

void vpz_context_destroy(vpz_context *c)
{
    if (!c) return;
    Context *ctx = &c->impl;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->tables) vpz::free_tables(kv.second);
    if (ctx->host_pool && ctx->host_pool_free) ctx->host_pool_free(ctx->host_pool);
    if (ctx->d_inv_db) (void)hipFree(ctx->d_inv_db);
    if (ctx->stage_in) (void)hipFree(ctx->stage_in);
    if (ctx->stage_out) (void)hipFree(ctx->stage_out);
    if (ctx->stage_aux) (void)hipFree(ctx->stage_aux);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete c;
}

int vpz_context_synchronize(vpz_context *c)
{
    if (!c) return VPZ_E_INVALID_ARG;
    VPZ_HIP_TRY(&c->impl, hipSetDevice(c->impl.device));
    VPZ_HIP_TRY(&c->impl, hipStreamSynchronize(c->impl.stream));
    return VPZ_OK;
}

const char *vpz_context_last_error(vpz_context *c) { return c ? c->impl.last_error.c_str() : ""; }

void *vpz_context_stream(vpz_context *c) { return c ? (void *)c->impl.stream : nullptr; }

int vpz_context_timer_start(vpz_context *c)
{
    if (!c) return VPZ_E_INVALID_ARG;
    VPZ_HIP_TRY(&c->impl, hipEventRecord(c->impl.ev_start, c->impl.stream));
    return VPZ_OK;
}

int vpz_context_timer_stop(vpz_context *c, float *elapsed_ms)
{
    if (!c || !elapsed_ms) return VPZ_E_INVALID_ARG;
    VPZ_HIP_TRY(&c->impl, hipEventRecord(c->impl.ev_stop, c->impl.stream));
    VPZ_HIP_TRY(&c->impl, hipEventSynchronize(c->impl.ev_stop));
    VPZ_HIP_TRY(&c->impl, hipEventElapsedTime(elapsed_ms, c->impl.ev_start, c->impl.ev_stop));
    return VPZ_OK;
}

int vpz_device_alloc(vpz_context *c, uint64_t bytes, void **dev_ptr)
{
    if (!c || !dev_ptr) return VPZ_E_INVALID_ARG;
    VPZ_HIP_TRY(&c->impl, hipSetDevice(c->impl.device));
    hipError_t e = hipMalloc(dev_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) return vpz::set_error(&c->impl, VPZ_E_NOMEM, "hipMalloc", e);
    return VPZ_OK;
}

// page-locked host memory: what a host-memory synth call copies from / to at the link's full rate (pageable memory goes
// through the runtime's own staging buffers)
int vpz_host_alloc(vpz_context *c, uint64_t bytes, void **host_ptr)
{
    if (!c || !host_ptr) return VPZ_E_INVALID_ARG;
    *host_ptr = nullptr;
    // (may be called while another thread is inside a synth call of the same context -- the dispatcher's workers get their
    // page-locked arrays this way --: nothing of the context is written, not even its error text; the status says it all)
    if (hipSetDevice(c->impl.device) != hipSuccess) { (void)hipGetLastError(); return VPZ_E_HIP; }
    hipError_t e = hipHostMalloc(host_ptr, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); *host_ptr = nullptr; return VPZ_E_NOMEM; }
    return VPZ_OK;
}

int vpz_host_free(vpz_context *c, void *host_ptr)
{
    if (!c) return VPZ_E_INVALID_ARG;
    if (!host_ptr) return VPZ_OK;
    if (hipSetDevice(c->impl.device) != hipSuccess || hipHostFree(host_ptr) != hipSuccess) { (void)hipGetLastError(); return VPZ_E_HIP; }
    return VPZ_OK;
}

int vpz_device_free(vpz_context *c, void *dev_ptr)
{
    if (!c) return VPZ_E_INVALID_ARG;
    VPZ_HIP_TRY(&c->impl, hipSetDevice(c->impl.device));
    VPZ_HIP_TRY(&c->impl, hipStreamSynchronize(c->impl.stream));
    VPZ_HIP_TRY(&c->impl, hipFree(dev_ptr));
    return VPZ_OK;
}

int vpz_memcpy_h2d(vpz_context *c, void *dev_dst, const void *host_src, uint64_t bytes)
{
    if (!c || (!dev_dst && bytes) || (!host_src && bytes)) return VPZ_E_INVALID_ARG;
    VPZ_HIP_TRY(&c->impl, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, c->impl.stream));
    VPZ_HIP_TRY(&c->impl, hipStreamSynchronize(c->impl.stream));
    return VPZ_OK;
}

int vpz_memcpy_d2h(vpz_context *c, void *host_dst, const void *dev_src, uint64_t bytes)
{
    if (!c || (!host_dst && bytes) || (!dev_src && bytes)) return VPZ_E_INVALID_ARG;
    VPZ_HIP_TRY(&c->impl, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, c->impl.stream));
    VPZ_HIP_TRY(&c->impl, hipStreamSynchronize(c->impl.stream));
    return VPZ_OK;
}

int vpz_imdct_batch(vpz_context *c, int n, int64_t count, const float *spectra, float *out,
                    int mem_space, int mode)
{
    if (!c) return VPZ_E_INVALID_ARG;
    Context *ctx = &c->impl;
    if (count < 0 || (count > 0 && (!spectra || !out)))
        return vpz::set_error(ctx, VPZ_E_INVALID_ARG, "vpz_imdct_batch: null buffer or negative count");
    if (mem_space != VPZ_MEM_HOST && mem_space != VPZ_MEM_DEVICE)
        return vpz::set_error(ctx, VPZ_E_INVALID_ARG, "vpz_imdct_batch: bad mem_space");
    if (mode != VPZ_IMDCT_FAST && mode != VPZ_IMDCT_EXACT)
        return vpz::set_error(ctx, VPZ_E_INVALID_ARG, "vpz_imdct_batch: bad mode");
    VPZ_HIP_TRY(ctx, hipSetDevice(ctx->device));
    vpz::BlockTables *t = nullptr;
    int rc = vpz::get_tables(ctx, n, &t);
    if (rc != VPZ_OK) return rc;
    if (count == 0) return VPZ_OK;

    const size_t in_bytes = sizeof(float) * (size_t)count * (size_t)(n / 2);
    const size_t out_bytes = sizeof(float) * (size_t)count * (size_t)n;
    const float *d_in = spectra;
    float *d_out = out;
    if (mem_space == VPZ_MEM_HOST) {
        if ((rc = vpz::ensure_stage(ctx, &ctx->stage_in, &ctx->stage_in_bytes, in_bytes)) != VPZ_OK) return rc;
        if ((rc = vpz::ensure_stage(ctx, &ctx->stage_out, &ctx->stage_out_bytes, out_bytes)) != VPZ_OK) return rc;
        VPZ_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_in, spectra, in_bytes, hipMemcpyHostToDevice, ctx->stream));
        d_in = static_cast<const float *>(ctx->stage_in);
        d_out = static_cast<float *>(ctx->stage_out);
    }

    hipError_t e;
    const bool fast = (mode == VPZ_IMDCT_FAST) && t->d_fast != nullptr;
    if (fast && n == 2048)
        e = vpz::launch_imdct_fast_2048(d_in, d_out, count, t->d_fast, ctx, ctx->stream);
    else if (fast && n == 256)
        e = vpz::launch_imdct_fast_256(d_in, d_out, count, t->d_fast, ctx, ctx->stream);
    else if (fast && (n == 512 || n == 1024))
        e = vpz::launch_imdct_fast_mid(n, d_in, d_out, count, t->d_fast, ctx, ctx->stream);
    else if (fast && n == 4096)
        e = vpz::launch_imdct_fast_4096(d_in, d_out, count, t->d_fast, ctx, ctx->stream);
    else if (fast && n == 8192)
        e = vpz::launch_imdct_fast_8192(d_in, d_out, count, t->d_fast, ctx, ctx->stream);
    else
        e = vpz::launch_imdct_exact(n, t->ld, d_in, d_out, count, t->d_A, t->d_B, t->d_C, t->d_bitrev,
                                    ctx->num_cu, ctx->stream);
    if (e != hipSuccess) return vpz::set_error(ctx, VPZ_E_HIP, "imdct kernel launch", e);

    if (mem_space == VPZ_MEM_HOST) {
        VPZ_HIP_TRY(ctx, hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        VPZ_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return VPZ_OK;
}

}  // extern "C"
