// VPZ_IMDCT_EXACT: the reference's own inverse-MDCT butterfly schedule (Mdct.cs:98-414), run by one
// workgroup per channel-block with the two work arrays in LDS.  Every float result is produced by
// the same sequence of separately rounded f32 multiplies / adds as the reference's scalar code
// (this translation unit is compiled with -ffp-contract=off), so the output is bit-identical to
// `Mdct.Reverse` for every Vorbis block size 64..8192 -- including the literal (non-IMDCT) result
// the reference gives for N = 64/128.  It is the any-block-size path and the
// bit-exact parity anchor; the headline path is imdct_fast.hip.
//
// Each pass of the schedule is a set of independent butterflies, so a pass is spread over the
// workgroup's threads and passes are separated by barriers.
#include "vpz_internal.hpp"

namespace vpz {

constexpr int kExactThreads = 256;

// One radix-2 pass `l` of step 3 (Mdct.cs:202-238: iter0, iter1, the r-loops and the s-loops are
// all this butterfly with k0 = n >> (l+2), twiddle stride k1 = 8 << l; the reference merely
// interchanges the two loops).  pairs_per_call = 4 * ((n >> (l+4)) >> 2), exactly as the unrolled
// loops execute (0 for n = 64, l = 1).
__device__ __forceinline__ void step3_pass(float *u, const float *A, int n, int l, int tid)
{
    const int n2 = n >> 1;
    const int k0 = n >> (l + 2);
    const int k0_2 = k0 >> 1;
    const int k1 = 8 << l;
    const int calls = 2 << l;
    const int pairs = ((n >> (l + 4)) >> 2) << 2;
    const int total = calls * pairs;
    for (int w = tid; w < total; w += kExactThreads) {
        const int i = w / pairs, t = w - i * pairs;
        const int e0 = n2 - 1 - k0 * i - 2 * t;
        const int e2 = e0 - k0_2;
        const float a0 = A[k1 * t], a1 = A[k1 * t + 1];
        const float x0 = u[e0], x1 = u[e0 - 1], y0 = u[e2], y1 = u[e2 - 1];
        const float k00 = x0 - y0;
        const float k01 = x1 - y1;
        u[e0] = x0 + y0;
        u[e0 - 1] = x1 + y1;
        u[e2] = k00 * a0 - k01 * a1;
        u[e2 - 1] = k01 * a0 + k00 * a1;
    }
}

// Mdct.cs:695-726 on z[0], z[-1] .. z[-7]
__device__ __forceinline__ void iter_54(float *z)
{
    float k00 = z[0] - z[-4];
    float y0 = z[0] + z[-4];
    float y2 = z[-2] + z[-6];
    float k22 = z[-2] - z[-6];
    z[0] = y0 + y2;
    z[-2] = y0 - y2;
    float k33 = z[-3] - z[-7];
    z[-4] = k00 + k33;
    z[-6] = k00 - k33;
    float k11 = z[-1] - z[-5];
    float y1 = z[-1] + z[-5];
    float y3 = z[-3] + z[-7];
    z[-1] = y1 + y3;
    z[-3] = y1 - y3;
    z[-5] = k11 - k22;
    z[-7] = k11 + k22;
}

__global__ __launch_bounds__(kExactThreads) void imdct_exact_kernel(
    int n, int ld, const float *__restrict__ spectra, float *__restrict__ out, long count,
    const float *__restrict__ A, const float *__restrict__ B, const float *__restrict__ C,
    const uint16_t *__restrict__ bitrev, const int64_t *__restrict__ src_off, const int64_t *__restrict__ dst_off)
{
    extern __shared__ float s_mem[];
    float *buffer = s_mem;       // n floats  (the reference's `buffer`, alias u)
    float *buf2 = s_mem + n;     // n/2 floats (`buf2`, alias v)
    const int tid = threadIdx.x;
    const int n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;

    for (long blk = blockIdx.x; blk < count; blk += gridDim.x) {
        const float *src = src_off ? spectra + src_off[blk] : spectra + blk * n2;
        for (int i = tid; i < n2; i += kExactThreads) buffer[i] = src[i];
        __syncthreads();

        // Step0 (Mdct.cs:98-125): buffer -> buf2
        for (int it = tid; it < n8; it += kExactThreads) {
            {
                const int d = n2 - 2 - 2 * it, aa = 2 * it, e = 4 * it;
                buf2[d + 1] = buffer[e] * A[aa] - buffer[e + 2] * A[aa + 1];
                buf2[d] = buffer[e] * A[aa + 1] + buffer[e + 2] * A[aa];
            }
            {
                const int d = n4 - 2 - 2 * it, aa = 2 * (n8 + it), e = n2 - 3 - 4 * it;
                buf2[d + 1] = -buffer[e + 2] * A[aa] - -buffer[e] * A[aa + 1];
                buf2[d] = -buffer[e + 2] * A[aa + 1] + -buffer[e] * A[aa];
            }
        }
        __syncthreads();

        // Step2 (Mdct.cs:140-178): v = buf2 -> u = buffer
        for (int it = tid; it < (n2 >> 3); it += kExactThreads) {
            const float *AA = A + (n2 - 8 - 8 * it);
            const float *e0 = buf2 + n4 + 4 * it;
            const float *e1 = buf2 + 4 * it;
            float *d0 = buffer + n4 + 4 * it;
            float *d1 = buffer + 4 * it;
            float v41_21 = e0[1] - e1[1];
            float v40_20 = e0[0] - e1[0];
            d0[1] = e0[1] + e1[1];
            d0[0] = e0[0] + e1[0];
            d1[1] = v41_21 * AA[4] - v40_20 * AA[5];
            d1[0] = v40_20 * AA[4] + v41_21 * AA[5];
            v41_21 = e0[3] - e1[3];
            v40_20 = e0[2] - e1[2];
            d0[3] = e0[3] + e1[3];
            d0[2] = e0[2] + e1[2];
            d1[3] = v41_21 * AA[0] - v40_20 * AA[1];
            d1[2] = v40_20 * AA[0] + v41_21 * AA[1];
        }
        __syncthreads();

        // Step3 (Mdct.cs:184-246): iteration 0 and 1 always run, then l = 2 .. ld-7, then the
        // fused last three passes
        step3_pass(buffer, A, n, 0, tid);
        __syncthreads();
        step3_pass(buffer, A, n, 1, tid);
        __syncthreads();
        for (int l = 2; l < ld - 6; ++l) {
            step3_pass(buffer, A, n, l, tid);
            __syncthreads();
        }
        {   // step3_inner_s_loop_ld654 (Mdct.cs:651-693): n>>5 independent groups of 16 floats
            const float A2 = A[n >> 3];
            for (int gi = tid; gi < (n >> 5); gi += kExactThreads) {
                float *z = buffer + (n2 - 1) - 16 * gi;
                float k00 = z[-0] - z[-8];
                float k11 = z[-1] - z[-9];
                float l00 = z[-2] - z[-10];
                float l11 = z[-3] - z[-11];
                z[-0] = z[-0] + z[-8];
                z[-1] = z[-1] + z[-9];
                z[-2] = z[-2] + z[-10];
                z[-3] = z[-3] + z[-11];
                z[-8] = k00;
                z[-9] = k11;
                z[-10] = (l00 + l11) * A2;
                z[-11] = (l11 - l00) * A2;

                k00 = z[-4] - z[-12];
                k11 = z[-5] - z[-13];
                l00 = z[-6] - z[-14];
                l11 = z[-7] - z[-15];
                z[-4] = z[-4] + z[-12];
                z[-5] = z[-5] + z[-13];
                z[-6] = z[-6] + z[-14];
                z[-7] = z[-7] + z[-15];
                z[-12] = k11;
                z[-13] = -k00;
                z[-14] = (l11 - l00) * A2;
                z[-15] = (l00 + l11) * -A2;

                iter_54(z);
                iter_54(z - 8);
            }
        }
        __syncthreads();

        // Step4_5_6 (Mdct.cs:256-288): bit-reverse gather u -> v
        for (int it = tid; it < (n4 >> 2); it += kExactThreads) {
            float *d0 = buf2 + n4 - 4 - 4 * it;
            float *d1 = buf2 + n2 - 4 - 4 * it;
            int k4 = bitrev[2 * it];
            d1[3] = buffer[k4 + 0];
            d1[2] = buffer[k4 + 1];
            d0[3] = buffer[k4 + 2];
            d0[2] = buffer[k4 + 3];
            k4 = bitrev[2 * it + 1];
            d1[1] = buffer[k4 + 0];
            d1[0] = buffer[k4 + 1];
            d0[1] = buffer[k4 + 2];
            d0[0] = buffer[k4 + 3];
        }
        __syncthreads();

        // Step7 (Mdct.cs:302-345): in place on v, pairs from both ends
        for (int it = tid; it < (n2 >> 3); it += kExactThreads) {
            float *d = buf2 + 4 * it;
            float *e = buf2 + n2 - 4 - 4 * it;
            const float *Cc = C + 4 * it;
            float a02 = d[0] - e[2];
            float a11 = d[1] + e[3];
            float b0 = Cc[1] * a02 + Cc[0] * a11;
            float b1 = Cc[1] * a11 - Cc[0] * a02;
            float b2 = d[0] + e[2];
            float b3 = d[1] - e[3];
            d[0] = b2 + b0;
            d[1] = b3 + b1;
            e[2] = b2 - b0;
            e[3] = b1 - b3;

            a02 = d[2] - e[0];
            a11 = d[3] + e[1];
            b0 = Cc[3] * a02 + Cc[2] * a11;
            b1 = Cc[3] * a11 - Cc[2] * a02;
            b2 = d[2] + e[0];
            b3 = d[3] - e[1];
            d[2] = b2 + b0;
            d[3] = b3 + b1;
            e[0] = b2 - b0;
            e[1] = b1 - b3;
        }
        __syncthreads();

        // Step8 (Mdct.cs:360-414): v -> global output, mirrored stores
        float *dst = dst_off ? out + dst_off[blk] : out + blk * n;
        for (int it = tid; it < (n2 >> 3); it += kExactThreads) {
            const float *Bb = B + n2 - 8 - 8 * it;
            const float *e = buf2 + n2 - 8 - 8 * it;
            float *d0 = dst + 4 * it;
            float *d1 = dst + n2 - 4 - 4 * it;
            float *d2 = dst + n2 + 4 * it;
            float *d3 = dst + n - 4 - 4 * it;
            float p3 = e[6] * Bb[7] - e[7] * Bb[6];
            float p2 = -e[6] * Bb[6] - e[7] * Bb[7];
            d0[0] = p3; d1[3] = -p3; d2[0] = p2; d3[3] = p2;
            float p1 = e[4] * Bb[5] - e[5] * Bb[4];
            float p0 = -e[4] * Bb[4] - e[5] * Bb[5];
            d0[1] = p1; d1[2] = -p1; d2[1] = p0; d3[2] = p0;
            p3 = e[2] * Bb[3] - e[3] * Bb[2];
            p2 = -e[2] * Bb[2] - e[3] * Bb[3];
            d0[2] = p3; d1[1] = -p3; d2[2] = p2; d3[1] = p2;
            p1 = e[0] * Bb[1] - e[1] * Bb[0];
            p0 = -e[0] * Bb[0] - e[1] * Bb[1];
            d0[3] = p1; d1[0] = -p1; d2[3] = p0; d3[0] = p0;
        }
        __syncthreads();
    }
}

hipError_t launch_imdct_exact(int n, int ld, const float *spectra, float *out, int64_t count,
                              const float *A, const float *B, const float *C,
                              const uint16_t *bitrev, int num_cu, hipStream_t stream,
                              const int64_t *src_off, const int64_t *dst_off)
{
    if (count <= 0) return hipSuccess;
    const size_t lds = sizeof(float) * (size_t)(n + n / 2);
    int64_t cap = (int64_t)num_cu * 4;
    int grid = (int)(count < cap ? count : cap);
    hipLaunchKernelGGL(imdct_exact_kernel, dim3(grid), dim3(kExactThreads), lds, stream, n, ld,
                       spectra, out, (long)count, A, B, C, bitrev, src_off, dst_off);
    return hipGetLastError();
}

}  // namespace vpz
