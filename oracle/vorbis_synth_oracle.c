/*
 * oracle/vorbis_synth_oracle.c -- TEST INFRASTRUCTURE ONLY (see vorbis_synth_oracle.h).
 *
 * Scalar C99 restatement of the reference's PCM-synthesis arithmetic.  Build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math   (see oracle/Makefile)
 * so that every float32 multiply/add is rounded separately, as the .NET JIT does (it never
 * contracts a*b+c; the AVX2/SSE variants in Mdct.cs:481-553 are op-for-op the scalar branch).
 * PARITY: "parity unpinned" by reference goldens (none exist); see header.
 */
#include "vorbis_synth_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI_F 3.14159274101257324219f /* MathF.PI */

/* ------------------------------------------------------------------ Utils.cs:19-42 */
int orc_ilog(int x)
{
    int cnt = 0;
    while (x > 0) { ++cnt; x >>= 1; }
    return cnt;
}

uint32_t orc_bit_reverse(uint32_t n, int bits)
{
    n = ((n & 0xAAAAAAAAu) >> 1) | ((n & 0x55555555u) << 1);
    n = ((n & 0xCCCCCCCCu) >> 2) | ((n & 0x33333333u) << 2);
    n = ((n & 0xF0F0F0F0u) >> 4) | ((n & 0x0F0F0F0Fu) << 4);
    n = ((n & 0xFF00FF00u) >> 8) | ((n & 0x00FF00FFu) << 8);
    n = (n >> 16) | (n << 16);
    /* C# masks the shift count of a 32-bit operand to 5 bits; bits==0 never happens for n>=16 */
    return n >> ((32 - bits) & 31);
}

/* ------------------------------------------------------------------ Mdct.cs:29-66 */
orc_mdct *orc_mdct_create(int n)
{
    orc_mdct *m = (orc_mdct *)calloc(1, sizeof *m);
    int n2 = n >> 1, n4 = n2 >> 1, n8 = n4 >> 1;
    int k, k2, i;
    m->n = n;
    m->ld = orc_ilog(n) - 1;
    m->a = (float *)malloc(sizeof(float) * (size_t)n2);
    m->b = (float *)malloc(sizeof(float) * (size_t)n2);
    m->c = (float *)malloc(sizeof(float) * (size_t)(n4 > 0 ? n4 : 1));
    m->bitrev = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(n8 > 0 ? n8 : 1));
    for (k = k2 = 0; k < n4; ++k, k2 += 2) {
        /* :45  MathF.SinCos(4 * k * MathF.PI / n) -- int product, then float ops left to right */
        float arg_a = (float)(4 * k) * ORC_PI_F / (float)n;
        m->a[k2] = cosf(arg_a);
        m->a[k2 + 1] = -sinf(arg_a);
        /* :49  (k2 + 1) * MathF.PI / n / 2 */
        float arg_b = (float)(k2 + 1) * ORC_PI_F / (float)n / 2.0f;
        m->b[k2] = cosf(arg_b) * .5f;
        m->b[k2 + 1] = sinf(arg_b) * .5f;
    }
    for (k = k2 = 0; k < n8; ++k, k2 += 2) {
        /* :55  2 * (k2 + 1) * MathF.PI / n */
        float arg_c = (float)(2 * (k2 + 1)) * ORC_PI_F / (float)n;
        m->c[k2] = cosf(arg_c);
        m->c[k2 + 1] = -sinf(arg_c);
    }
    for (i = 0; i < n8; ++i) /* :61-65 */
        m->bitrev[i] = (uint16_t)(orc_bit_reverse((uint32_t)i, m->ld - 3) << 2);
    return m;
}

void orc_mdct_destroy(orc_mdct *m)
{
    if (!m) return;
    free(m->a); free(m->b); free(m->c); free(m->bitrev); free(m);
}

/* Mdct.cs:424-470 */
static void step3_iter0_loop(int n, float *e, int i_off, int k_off, const float *A)
{
    float *ee0 = e + i_off;
    float *ee2 = ee0 + k_off;
    int i, j;
    for (i = n >> 2; i > 0; --i) {
        for (j = 0; j < 8; j += 2) { /* four unrolled pairs, A advances 8 per pair */
            float k00_20 = ee0[-j] - ee2[-j];
            float k01_21 = ee0[-j - 1] - ee2[-j - 1];
            ee0[-j] += ee2[-j];
            ee0[-j - 1] += ee2[-j - 1];
            ee2[-j] = k00_20 * A[0] - k01_21 * A[1];
            ee2[-j - 1] = k01_21 * A[0] + k00_20 * A[1];
            A += 8;
        }
        ee0 -= 8;
        ee2 -= 8;
    }
}

/* Mdct.cs:472-598, scalar branch :554-597 */
static void step3_inner_r_loop(int lim, float *e, int d0, int k_off, const float *A, int k1)
{
    float *e0 = e + d0;
    float *e2 = e0 + k_off;
    int i, j;
    for (i = lim >> 2; i > 0; --i) {
        for (j = 0; j < 8; j += 2) {
            float k00_20 = e0[-j] - e2[-j];
            float k01_21 = e0[-j - 1] - e2[-j - 1];
            e0[-j] += e2[-j];
            e0[-j - 1] += e2[-j - 1];
            e2[-j] = k00_20 * A[0] - k01_21 * A[1];
            e2[-j - 1] = k01_21 * A[0] + k00_20 * A[1];
            A += k1;
        }
        e0 -= 8;
        e2 -= 8;
    }
}

/* Mdct.cs:600-649 */
static void step3_inner_s_loop(int n, float *e, int i_off, int k_off, const float *A, int a_off, int k0)
{
    float A0 = A[0], A1 = A[1];
    float A2 = A[a_off], A3 = A[a_off + 1];
    float A4 = A[a_off * 2], A5 = A[a_off * 2 + 1];
    float A6 = A[a_off * 3], A7 = A[a_off * 3 + 1];
    float *ee0 = e + i_off;
    float *ee2 = ee0 + k_off;
    int i;
    for (i = n; i > 0; --i) {
        float k00, k11;
        k00 = ee0[0] - ee2[0];
        k11 = ee0[-1] - ee2[-1];
        ee0[0] = ee0[0] + ee2[0];
        ee0[-1] = ee0[-1] + ee2[-1];
        ee2[0] = k00 * A0 - k11 * A1;
        ee2[-1] = k11 * A0 + k00 * A1;

        k00 = ee0[-2] - ee2[-2];
        k11 = ee0[-3] - ee2[-3];
        ee0[-2] = ee0[-2] + ee2[-2];
        ee0[-3] = ee0[-3] + ee2[-3];
        ee2[-2] = k00 * A2 - k11 * A3;
        ee2[-3] = k11 * A2 + k00 * A3;

        k00 = ee0[-4] - ee2[-4];
        k11 = ee0[-5] - ee2[-5];
        ee0[-4] = ee0[-4] + ee2[-4];
        ee0[-5] = ee0[-5] + ee2[-5];
        ee2[-4] = k00 * A4 - k11 * A5;
        ee2[-5] = k11 * A4 + k00 * A5;

        k00 = ee0[-6] - ee2[-6];
        k11 = ee0[-7] - ee2[-7];
        ee0[-6] = ee0[-6] + ee2[-6];
        ee0[-7] = ee0[-7] + ee2[-7];
        ee2[-6] = k00 * A6 - k11 * A7;
        ee2[-7] = k11 * A6 + k00 * A7;

        ee0 -= k0;
        ee2 -= k0;
    }
}

/* Mdct.cs:695-726 */
static void iter_54(float *z)
{
    float k00, k11, k22, k33;
    float y0, y1, y2, y3;

    k00 = z[0] - z[-4];
    y0 = z[0] + z[-4];
    y2 = z[-2] + z[-6];
    k22 = z[-2] - z[-6];

    z[-0] = y0 + y2;
    z[-2] = y0 - y2;

    k33 = z[-3] - z[-7];

    z[-4] = k00 + k33;
    z[-6] = k00 - k33;

    k11 = z[-1] - z[-5];
    y1 = z[-1] + z[-5];
    y3 = z[-3] + z[-7];

    z[-1] = y1 + y3;
    z[-3] = y1 - y3;
    z[-5] = k11 - k22;
    z[-7] = k11 + k22;
}

/* Mdct.cs:651-693 */
static void step3_inner_s_loop_ld654(int n, float *e, int i_off, const float *A, int base_n)
{
    int a_off = base_n >> 3;
    float A2 = A[a_off];
    float *z = e + i_off;
    float *base = z - 16 * n;

    while (z > base) {
        float k00, k11, l00, l11;

        k00 = z[-0] - z[-8];
        k11 = z[-1] - z[-9];
        l00 = z[-2] - z[-10];
        l11 = z[-3] - z[-11];
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
        z -= 16;
    }
}

/* Mdct.cs:77-419 */
void orc_mdct_reverse(const orc_mdct *m, float *buffer, float *buf2)
{
    const int n = m->n, n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
    const int ld = m->ld;
    const float *A = m->a;
    float *u = buffer, *v = buf2;

    /* Step0 :98-125 */
    {
        float *d = &buf2[n2 - 2];
        const float *AA = A;
        const float *e = &buffer[0];
        const float *e_stop = &buffer[n2];
        while (e != e_stop) {
            d[1] = e[0] * AA[0] - e[2] * AA[1];
            d[0] = e[0] * AA[1] + e[2] * AA[0];
            d -= 2; AA += 2; e += 4;
        }
        e = &buffer[n2 - 3];
        while (d >= buf2) {
            d[1] = -e[2] * AA[0] - -e[0] * AA[1];
            d[0] = -e[2] * AA[1] + -e[0] * AA[0];
            d -= 2; AA += 2; e -= 4;
        }
    }

    /* Step2 :140-178 */
    {
        const float *AA = &A[n2 - 8];
        const float *e0 = &v[n4];
        const float *e1 = &v[0];
        float *d0 = &u[n4];
        float *d1 = &u[0];
        while (AA >= A) {
            float v40_20, v41_21;

            v41_21 = e0[1] - e1[1];
            v40_20 = e0[0] - e1[0];
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

            AA -= 8;
            d0 += 4; d1 += 4; e0 += 4; e1 += 4;
        }
    }

    /* Step3 :184-246 */
    {
        int l, i, r;
        step3_iter0_loop(n >> 4, u, n2 - 1 - n4 * 0, -(n >> 3), A);
        step3_iter0_loop(n >> 4, u, n2 - 1 - n4 * 1, -(n >> 3), A);

        step3_inner_r_loop(n >> 5, u, n2 - 1 - n8 * 0, -(n >> 4), A, 16);
        step3_inner_r_loop(n >> 5, u, n2 - 1 - n8 * 1, -(n >> 4), A, 16);
        step3_inner_r_loop(n >> 5, u, n2 - 1 - n8 * 2, -(n >> 4), A, 16);
        step3_inner_r_loop(n >> 5, u, n2 - 1 - n8 * 3, -(n >> 4), A, 16);

        l = 2;
        for (; l < (ld - 3) >> 1; ++l) {
            int k0 = n >> (l + 2);
            int k0_2 = k0 >> 1;
            int lim = 1 << (l + 1);
            for (i = 0; i < lim; ++i)
                step3_inner_r_loop(n >> (l + 4), u, n2 - 1 - k0 * i, -k0_2, A, 1 << (l + 3));
        }
        for (; l < ld - 6; ++l) {
            int k0 = n >> (l + 2);
            int k1 = 1 << (l + 3);
            int k0_2 = k0 >> 1;
            int rlim = n >> (l + 6);
            int lim = 1 << (l + 1);
            const float *A0 = A;
            int i_off = n2 - 1;
            for (r = rlim; r > 0; --r) {
                step3_inner_s_loop(lim, u, i_off, -k0_2, A0, k1, k0);
                A0 += k1 * 4;
                i_off -= 8;
            }
        }
        step3_inner_s_loop_ld654(n >> 5, u, n2 - 1, A, n);
    }

    /* Step4_5_6 :256-288 */
    {
        const uint16_t *bitrev = m->bitrev;
        float *d0 = &v[n4 - 4];
        float *d1 = &v[n2 - 4];
        while (d0 >= v) {
            int k4;
            k4 = bitrev[0];
            d1[3] = u[k4 + 0];
            d1[2] = u[k4 + 1];
            d0[3] = u[k4 + 2];
            d0[2] = u[k4 + 3];

            k4 = bitrev[1];
            d1[1] = u[k4 + 0];
            d1[0] = u[k4 + 1];
            d0[1] = u[k4 + 2];
            d0[0] = u[k4 + 3];

            d0 -= 4; d1 -= 4; bitrev += 2;
        }
    }

    /* Step7 :302-345 */
    {
        const float *C = m->c;
        float *d = v;
        float *e = v + n2 - 4;
        while (d < e) {
            float a02, a11, b0, b1, b2, b3;

            a02 = d[0] - e[2];
            a11 = d[1] + e[3];
            b0 = C[1] * a02 + C[0] * a11;
            b1 = C[1] * a11 - C[0] * a02;
            b2 = d[0] + e[2];
            b3 = d[1] - e[3];
            d[0] = b2 + b0;
            d[1] = b3 + b1;
            e[2] = b2 - b0;
            e[3] = b1 - b3;

            a02 = d[2] - e[0];
            a11 = d[3] + e[1];
            b0 = C[3] * a02 + C[2] * a11;
            b1 = C[3] * a11 - C[2] * a02;
            b2 = d[2] + e[0];
            b3 = d[3] - e[1];
            d[2] = b2 + b0;
            d[3] = b3 + b1;
            e[0] = b2 - b0;
            e[1] = b1 - b3;

            C += 4; d += 4; e -= 4;
        }
    }

    /* Step8 :360-414 */
    {
        const float *B = m->b + n2 - 8;
        const float *e = buf2 + n2 - 8;
        float *d0 = &buffer[0];
        float *d1 = &buffer[n2 - 4];
        float *d2 = &buffer[n2];
        float *d3 = &buffer[n - 4];
        while (e >= v) {
            float p0, p1, p2, p3;

            p3 = e[6] * B[7] - e[7] * B[6];
            p2 = -e[6] * B[6] - e[7] * B[7];
            d0[0] = p3; d1[3] = -p3; d2[0] = p2; d3[3] = p2;

            p1 = e[4] * B[5] - e[5] * B[4];
            p0 = -e[4] * B[4] - e[5] * B[5];
            d0[1] = p1; d1[2] = -p1; d2[1] = p0; d3[2] = p0;

            p3 = e[2] * B[3] - e[3] * B[2];
            p2 = -e[2] * B[2] - e[3] * B[3];
            d0[2] = p3; d1[1] = -p3; d2[2] = p2; d3[1] = p2;

            p1 = e[0] * B[1] - e[1] * B[0];
            p0 = -e[0] * B[0] - e[1] * B[1];
            d0[3] = p1; d1[0] = -p1; d2[3] = p0; d3[0] = p0;

            B -= 8; e -= 8;
            d0 += 4; d2 += 4; d1 -= 4; d3 -= 4;
        }
    }
}

void orc_mdct_reverse_batch(int n, long count, const float *spectra, float *out)
{
    orc_mdct *m = orc_mdct_create(n);
    float *buf2 = (float *)malloc(sizeof(float) * (size_t)n);
    long r;
    for (r = 0; r < count; ++r) {
        float *row = out + (size_t)r * (size_t)n;
        memcpy(row, spectra + (size_t)r * (size_t)(n / 2), sizeof(float) * (size_t)(n / 2));
        orc_mdct_reverse(m, row, buf2);
    }
    free(buf2);
    orc_mdct_destroy(m);
}

/* ------------------------------------------------------------------ BlocksizeDerivedCache.cs:25-36 */
void orc_window_slope(int half, float *slope)
{
    int x;
    for (x = 0; x < half; ++x) {
        float v = sinf(0.5f * ORC_PI_F * ((float)x + 0.5f) / (float)half);
        slope[x] = sinf(0.5f * ORC_PI_F * v * v);
    }
}

/* ------------------------------------------------------------------ Mode.cs:30-66 */
void orc_get_packet_info(int size0, int size1, int block_flag, int prev_flag, int next_flag,
                         orc_packet_info *info)
{
    int size = block_flag ? size1 : size0;
    int prev = block_flag ? prev_flag : 1; /* flags?.prev ?? true */
    int next = block_flag ? next_flag : 1;
    int center = size / 2;
    if (prev) {
        info->LeftStart = 0;
        info->LeftEnd = center;
        info->Length = size / 2;
        info->LeftUseSize1 = block_flag ? 1 : 0;
    } else {
        info->LeftStart = (size - size0) / 4;
        info->LeftEnd = (size + size0) / 4;
        info->Length = size0 / 2;
        info->LeftUseSize1 = 0;
    }
    if (next) {
        info->RightStart = center;
        info->RightEnd = size;
    } else {
        info->RightStart = (size * 3 - size0) / 4;
        info->RightEnd = (size * 3 + size0) / 4;
    }
}

/* ------------------------------------------------------------------ Mapping.cs:198-269 */
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

void orc_apply_coupling(float *mag, float *ang, int n, int vector_form)
{
    int j;
    for (j = 0; j < n; ++j) {
        float oldM = mag[j], oldA = ang[j];
        float newM, newA;
        if (vector_form) {
            /* :209-225 restated lane-wise with the same bit operations */
            uint32_t posM = (oldM > 0.0f) ? 0xFFFFFFFFu : 0u;
            uint32_t posA = (oldA > 0.0f) ? 0xFFFFFFFFu : 0u;
            uint32_t signMask = 0x80000000u & posM;
            uint32_t signedA = f2u(oldA) ^ signMask;
            newM = oldM - u2f(signedA & ~posA);
            newA = oldM + u2f(signedA & posA);
        } else {
            /* :235-268 */
            newM = oldM;
            newA = oldM;
            if (oldM > 0) {
                if (oldA > 0) newA = oldM - oldA;
                else          newM = oldM + oldA;
            } else {
                if (oldA > 0) newA = oldM + oldA;
                else          newM = oldM - oldA;
            }
        }
        mag[j] = newM;
        ang[j] = newA;
    }
}

/* ------------------------------------------------------------------ Residue2.cs:42-51 */
void orc_residue2_deinterleave(const float *src, int half, int channels, float *dst, int stride)
{
    int ch, i;
    for (ch = 0; ch < channels; ++ch)
        for (i = 0; i < half; ++i)
            dst[(size_t)ch * (size_t)stride + (size_t)i] = src[(size_t)i * (size_t)channels + (size_t)ch];
}

/* ------------------------------------------------------------------ Floor1.cs */
static const uint32_t k_inverse_db_bits[256] = {
#include "floor1_inverse_db_bits.inc"
};
static float g_inverse_db[256];
static int g_inverse_db_ready = 0;

const float *orc_floor1_inverse_db_table(void)
{
    if (!g_inverse_db_ready) {
        int i;
        for (i = 0; i < 256; ++i) g_inverse_db[i] = u2f(k_inverse_db_bits[i]);
        g_inverse_db_ready = 1;
    }
    return g_inverse_db;
}

static const int k_range_lookup[4] = {128, 64, 43, 32}; /* Floor1.cs:36 */

/* Floor1.cs:108-149 (+ :80 for range) */
int orc_floor1_init(orc_floor1 *f, const int *xlist, int count, int multiplier)
{
    int i, j;
    memset(f, 0, sizeof *f);
    if (count < 2 || count > 65 || multiplier < 1 || multiplier > 4) return -1;
    f->count = count;
    f->multiplier = multiplier;
    f->range = k_range_lookup[multiplier - 1] * 2;
    for (i = 0; i < count; ++i) f->xlist[i] = xlist[i];
    f->sortidx[0] = 0;
    f->sortidx[1] = 1;
    for (i = 2; i < count; ++i) {
        f->lneigh[i] = 0;
        f->hneigh[i] = 1;
        f->sortidx[i] = i;
        for (j = 2; j < i; ++j) {
            int temp = f->xlist[j];
            if (temp < f->xlist[i]) {
                if (temp > f->xlist[f->lneigh[i]]) f->lneigh[i] = j;
            } else {
                if (temp < f->xlist[f->hneigh[i]]) f->hneigh[i] = j;
            }
        }
    }
    for (i = 0; i < count - 1; ++i) {
        for (j = i + 1; j < count; ++j) {
            if (f->xlist[i] == f->xlist[j]) return -1;
            if (f->xlist[f->sortidx[i]] > f->xlist[f->sortidx[j]]) {
                int t = f->sortidx[i];
                f->sortidx[i] = f->sortidx[j];
                f->sortidx[j] = t;
            }
        }
    }
    return 0;
}

static int iabs(int x) { int sign = x >> 31; return (x ^ sign) - sign; } /* :399-403 */

/* Floor1.cs:355-370 */
static int render_point(int x0, int y0, int x1, int y1, int X)
{
    int dy = y1 - y0;
    int adx = x1 - x0;
    int ady = iabs(dy);
    int err = ady * (X - x0);
    int off = err / adx;
    return dy < 0 ? y0 - off : y0 + off;
}

/* Floor1.cs:270-353 */
void orc_floor1_unwrap_posts(const orc_floor1 *f, int *posts, int post_count, uint8_t *step_flags)
{
    int finalY[64];
    int i;
    memset(finalY, 0, sizeof finalY);
    step_flags[0] = 1;
    step_flags[1] = 1;
    finalY[0] = posts[0];
    finalY[1] = posts[1];
    for (i = 2; i < post_count; ++i) {
        int lowOfs = f->lneigh[i];
        int highOfs = f->hneigh[i];
        int predicted = render_point(f->xlist[lowOfs], finalY[lowOfs],
                                     f->xlist[highOfs], finalY[highOfs], f->xlist[i]);
        int val = posts[i];
        int highroom = f->range - predicted;
        int lowroom = predicted;
        int room = (highroom < lowroom) ? highroom * 2 : lowroom * 2;
        int result;
        if (val != 0) {
            step_flags[lowOfs] = 1;
            step_flags[highOfs] = 1;
            step_flags[i] = 1;
            if (val >= room) {
                if (highroom > lowroom) result = val - lowroom + predicted;
                else                    result = predicted - val + highroom - 1;
            } else {
                if ((val % 2) == 1) result = predicted - ((val + 1) / 2);
                else                result = predicted + (val / 2);
            }
        } else {
            step_flags[i] = 0;
            result = predicted;
        }
        finalY[i] = result;
    }
    memcpy(posts, finalY, sizeof finalY); /* finalY.CopyTo(data.Posts) :352 -- all 64 entries */
}

/* Floor1.cs:372-397 */
static void render_line_multi(int x0, int y0, int x1, int y1, const float *db, float *v)
{
    int dy = y1 - y0;
    int adx = x1 - x0;
    int ady = iabs(dy);
    int sy = 1 - (((dy >> 31) & 1) * 2);
    int b = dy / adx;
    int x = x0;
    int y = y0;
    int err = -adx;

    v[x] *= db[y];
    ady -= iabs(b) * adx;

    while (++x < x1) {
        y += b;
        err += ady;
        if (err >= 0) {
            err -= adx;
            y += sy;
        }
        v[x] *= db[y];
    }
}

/* Floor1.cs:236-262 (the part of Apply after UnwrapPosts).  NOTE the reference does not
 * initialise stepFlags beyond what UnwrapPosts writes ([SkipLocalsInit] + stackalloc, :221,229);
 * indices >= 2 are always written by the loop, so this is well defined for post_count <= 64. */
void orc_floor1_render(const orc_floor1 *f, const int *final_y, const uint8_t *step_flags,
                       int post_count, int n, float *residue)
{
    const float *db = orc_floor1_inverse_db_table();
    int lx = 0;
    int ly = final_y[0] * f->multiplier;
    int i;
    for (i = 1; i < post_count; ++i) {
        int idx = f->sortidx[i];
        if (step_flags[idx]) {
            int hx = f->xlist[idx];
            int hy = final_y[idx] * f->multiplier;
            if (lx < n) {
                int x1 = hx < n ? hx : n; /* Math.Min(hx, n) enters the slope: quirk q2 */
                render_line_multi(lx, ly, x1, hy, db, residue);
            }
            lx = hx;
            ly = hy;
        }
        if (lx >= n) break;
    }
    if (lx < n) render_line_multi(lx, ly, n, ly, db, residue);
}

/* The integers of the render alone: the table index `y` RenderLineMulti (Floor1.cs:372-397) uses for every bin,
 * through the same walk as Apply (:236-262).  out_y[0..n), unclamped (the reference indexes its table with it). */
static void render_line_indices(int x0, int y0, int x1, int y1, int *out)
{
    int dy = y1 - y0;
    int adx = x1 - x0;
    int ady = iabs(dy);
    int sy = 1 - (((dy >> 31) & 1) * 2);
    int b = dy / adx;
    int x = x0;
    int y = y0;
    int err = -adx;

    out[x] = y;
    ady -= iabs(b) * adx;

    while (++x < x1) {
        y += b;
        err += ady;
        if (err >= 0) {
            err -= adx;
            y += sy;
        }
        out[x] = y;
    }
}

void orc_floor1_render_indices(const orc_floor1 *f, const int *final_y, const uint8_t *step_flags,
                               int post_count, int n, int *out_y)
{
    int lx = 0;
    int ly = final_y[0] * f->multiplier;
    int i;
    for (i = 1; i < post_count; ++i) {
        int idx = f->sortidx[i];
        if (step_flags[idx]) {
            int hx = f->xlist[idx];
            int hy = final_y[idx] * f->multiplier;
            if (lx < n) {
                int x1 = hx < n ? hx : n; /* Math.Min(hx, n) enters the slope: quirk q2 (:248) */
                render_line_indices(lx, ly, x1, hy, out_y);
            }
            lx = hx;
            ly = hy;
        }
        if (lx >= n) break;
    }
    if (lx < n) render_line_indices(lx, ly, n, ly, out_y);
}

/* Floor1.cs:222-268 */
void orc_floor1_apply(const orc_floor1 *f, int *posts, int post_count, int block_size, float *residue)
{
    int n = block_size / 2;
    if (post_count > 0) {
        uint8_t step_flags[64];
        memset(step_flags, 0, sizeof step_flags);
        orc_floor1_unwrap_posts(f, posts, post_count, step_flags);
        orc_floor1_render(f, posts, step_flags, post_count, n, residue);
    }
}

/* ------------------------------------------------------------------ Floor0.cs */
static float to_bark(double lsp) /* :97-100 */
{
    return (float)(13.1 * atan(0.00074 * lsp) + 2.24 * atan(0.0000000185 * lsp * lsp) + .0001 * lsp);
}

void orc_floor0_bark_map(const orc_floor0 *f, int n, int *map) /* :82-95 */
{
    float scale = (float)f->bark_map_size / to_bark(f->rate / 2.0);
    int i;
    for (i = 0; i < n + 1; ++i) map[i] = 0;
    for (i = 0; i < n + 1 - 2; ++i) {
        int v = (int)floor((double)(to_bark((f->rate / 2.0) / n * i) * scale));
        map[i] = v < f->bark_map_size - 1 ? v : f->bark_map_size - 1;
    }
    map[n] = -1;
}

int orc_floor0_apply(const orc_floor0 *f, float *coeff, float amp, int block_size, float *residue) /* :164-225 */
{
    int n = block_size / 2;
    int *bark = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    float wdel = (float)(3.14159265358979323846 / f->bark_map_size); /* (float)(Math.PI / _bark_map_size) */
    float amp_ofs = (float)f->amp_ofs;
    int i = 0, j, rc = 0;
    if (amp <= 0.0f) {
        memset(residue, 0, sizeof(float) * (size_t)n);
        free(bark);
        return 0;
    }
    orc_floor0_bark_map(f, n, bark);
    for (j = 0; j < f->order; ++j) coeff[j] = 2.0f * cosf(coeff[j]);
    while (i < n) {
        int k = bark[i];
        float p = .5f, q = .5f, w;
        if (k < 0 || k >= n) { rc = -1; break; } /* wMap has n entries (:102) */
        w = 2.0f * cosf(wdel * (float)k);     /* wMap[k], :106 */
        for (j = 1; j < f->order; j += 2) {
            q *= w - coeff[j - 1];
            p *= w - coeff[j];
        }
        if (j == f->order) {
            q *= w - coeff[j - 1];
            p *= p * (4.0f - w * w);
            q *= q;
        } else {
            p *= p * (2.0f - w);
            q *= q * (2.0f + w);
        }
        q = amp / sqrtf(p + q) - amp_ofs;
        q = expf(q * 0.11512925f);
        residue[i] *= q;
        while (bark[++i] == k) residue[i] *= q;
    }
    free(bark);
    return rc;
}

/* ------------------------------------------------------------------ Utils.cs:9-10,44-58 */
float orc_clip_value(float value, int *clipped)
{
    const float LowerClip = -0.99999994f, UpperClip = 0.99999994f;
    if (value > UpperClip) { *clipped = 1; return UpperClip; }
    if (value < LowerClip) { *clipped = 1; return LowerClip; }
    return value;
}

/* ------------------------------------------------------------------ Mapping.cs:166-195 */
void orc_mapping_synth(int channels, int block_size, float *buffer, int stride,
                       const orc_floor1 *floors, const int *floor_of_channel,
                       int *posts, const int *post_count,
                       const uint8_t *coupling_mag, const uint8_t *coupling_ang, int coupling_steps,
                       int coupling_vector_form)
{
    int half = block_size / 2;
    int i, ch;
    orc_mdct *m = orc_mdct_create(block_size);
    float *buf2 = (float *)malloc(sizeof(float) * (size_t)block_size);
    for (i = coupling_steps - 1; i >= 0; --i) /* :166-172 */
        orc_apply_coupling(buffer + (size_t)coupling_mag[i] * (size_t)stride,
                           buffer + (size_t)coupling_ang[i] * (size_t)stride, half,
                           coupling_vector_form);
    for (ch = 0; ch < channels; ++ch) { /* :180-195 */
        float *span = buffer + (size_t)ch * (size_t)stride;
        if (post_count[ch] > 0) {
            orc_floor1_apply(&floors[floor_of_channel[ch]], posts + ch * 64, post_count[ch],
                             block_size, span);
            orc_mdct_reverse(m, span, buf2);
        } else {
            memset(span, 0, sizeof(float) * (size_t)half);
        }
    }
    free(buf2);
    orc_mdct_destroy(m);
}

/* ------------------------------------------------------------------ StreamDecoder.cs (decode half) */
struct orc_stream {
    int channels, size0, size1;
    float *slope0, *slope1;        /* :226-229 */
    float *buf[2];                 /* buffer pool :371-396 */
    float *prev_buf, *next_buf;    /* _prevPacketBuf / _nextPacketBuf */
    int prev_start, prev_end, prev_stop; /* :45-49 */
    int64_t current_position;
    int has_position;
    int eos_found;
    int has_clipped;
};

orc_stream *orc_stream_create(int channels, int size0, int size1)
{
    orc_stream *s = (orc_stream *)calloc(1, sizeof *s);
    s->channels = channels;
    s->size0 = size0;
    s->size1 = size1;
    s->slope0 = (float *)malloc(sizeof(float) * (size_t)(size0 / 2));
    s->slope1 = (float *)malloc(sizeof(float) * (size_t)(size1 / 2));
    orc_window_slope(size0 / 2, s->slope0);
    orc_window_slope(size1 / 2, s->slope1);
    s->buf[0] = (float *)calloc((size_t)size1 * (size_t)channels, sizeof(float));
    s->buf[1] = (float *)calloc((size_t)size1 * (size_t)channels, sizeof(float));
    /* ProcessHeaderPackets :165-168: _currentPosition = 0; ResetDecoder(); _hasPosition = true */
    s->current_position = 0;
    s->has_position = 1;
    return s;
}

void orc_stream_destroy(orc_stream *s)
{
    if (!s) return;
    free(s->slope0); free(s->slope1); free(s->buf[0]); free(s->buf[1]); free(s);
}

void orc_stream_reset(orc_stream *s) /* :357-369 */
{
    s->prev_buf = NULL;
    s->next_buf = NULL;
    s->prev_start = s->prev_end = s->prev_stop = 0;
    s->eos_found = 0;
    s->has_clipped = 0;
    s->has_position = 0;
}

void orc_stream_mark_resync(orc_stream *s) /* DecodeNextPacket :718-722: packet.IsResync */
{
    s->has_position = 0;
}

/* StreamDecoder.SeekTo(long samplePosition, SeekOrigin.Begin) :817-880, from the point where the packet provider has
 * been positioned: provider_pos is what `_packetProvider.SeekTo(samplePosition, 1, this)` returned (the stream
 * position at the start of the packet AFTER the pre-roll packet), read_next_packet performs one ReadNextPacket (the
 * caller decodes the provider's next packet into orc_stream_next_buffer and calls orc_stream_read_next_packet; its
 * result > 0 is ReadNextPacket's `true`).  Returns 0, -1 for SeekOutOfRangeException, -2 for PreRollPacketException. */
int orc_stream_seek_to(orc_stream *s, int64_t sample_position, int64_t provider_pos, int64_t max_granule_count,
                       orc_read_next_packet_fn read_next_packet, void *user)
{
    int roll_forward = (int)(sample_position - provider_pos); /* :848 */

    orc_stream_reset(s);   /* :851 */
    s->has_position = 1;   /* :852 */

    if (read_next_packet(user) <= 0) { /* the pre-roll packet, :855-867 */
        s->eos_found |= 8; /* EndOfStreamFlags.InvalidPreroll = 1 << 3 */
        if (sample_position > max_granule_count) return -1;
        s->prev_start = s->prev_stop;
        s->current_position = sample_position;
        return 0;
    }
    if (read_next_packet(user) <= 0) { /* the actual packet, :870-876 */
        orc_stream_reset(s);
        s->eos_found |= 1; /* EndOfStreamFlags.InvalidPacket = 1 << 0 */
        return -2;
    }
    s->prev_start += roll_forward;          /* :879 */
    s->current_position = sample_position;  /* :880 */
    return 0;
}

float *orc_stream_next_buffer(orc_stream *s) /* _nextPacketBuf ??= GetBuffer() :738 */
{
    if (!s->next_buf)
        s->next_buf = (s->prev_buf == s->buf[0]) ? s->buf[1] : s->buf[0];
    return s->next_buf;
}

/* :764-791 */
static int overlap_buffers(orc_stream *s, const orc_packet_info *info, float *next_buffer, int packet_len)
{
    const float *slope = info->LeftUseSize1 ? s->slope1 : s->slope0;
    int slope_len = info->LeftUseSize1 ? s->size1 / 2 : s->size0 / 2;
    int size1 = s->size1;
    int ch, i;
    if (packet_len > slope_len) return -1; /* windowSlope.AsSpan(0, packetLen) throws :778 */
    for (ch = 0; ch < s->channels; ++ch) {
        const float *prev = s->prev_buf + (size_t)size1 * (size_t)ch + s->prev_end;
        float *chan = next_buffer + info->LeftStart + (size_t)size1 * (size_t)ch;
        for (i = 0; i < packet_len; ++i) {
            float v = chan[i];
            float v_lhs = slope[i];
            float v_prev = prev[i];
            float v_rhs = slope[packet_len - (i + 1)];
            chan[i] = (v * v_lhs) + (v_prev * v_rhs);
        }
    }
    return 0;
}

/* :640-694 */
int orc_stream_read_next_packet(orc_stream *s, int decoded, const orc_packet_info *info,
                                int64_t granule, int eos_flag)
{
    float *cur = decoded ? orc_stream_next_buffer(s) : NULL;
    int64_t sample_position = decoded ? granule : -1; /* :758-761 */
    int packet_len, right_start;

    s->eos_found |= eos_flag ? 4 : 0; /* EndOfStreamFlags.PacketFlag = 1 << 2 */
    if (!cur) return 0;

    packet_len = s->prev_stop - s->prev_end;
    right_start = info->RightStart;

    if (sample_position != -1 && eos_flag) { /* :658-666 */
        int64_t actual_end = s->current_position + packet_len;
        int diff = (int)(actual_end - sample_position);
        if (diff > 0) {
            right_start = right_start - diff;
            if (right_start < 0) right_start = 0;
        }
    }

    if (s->prev_buf) { /* :670-675 */
        if (overlap_buffers(s, info, cur, packet_len) != 0) return -1;
        s->prev_start = info->LeftStart;
    } else {
        s->prev_start = right_start; /* :679 */
    }
    s->prev_end = right_start;
    s->prev_stop = info->RightEnd;

    s->next_buf = s->prev_buf; /* :689 */
    s->prev_buf = cur;         /* :692 */

    /* Read(): pick up a position (:459-463); idx is 0 whenever a packet is fetched */
    if (sample_position != -1 && !s->has_position) {
        s->has_position = 1;
        s->current_position = sample_position - (s->prev_end - s->prev_start);
    }
    return 1;
}

int orc_stream_available(const orc_stream *s) { return s->prev_end - s->prev_start; }

void orc_stream_drain_eos(orc_stream *s) { s->prev_end = s->prev_stop; } /* :451-455 */

/* :515-592 (generic scalar tail :573-591, value-identical to the SSE paths) and :594-638 */
void orc_stream_store(orc_stream *s, float *dst, long offset, int count, long channel_stride,
                      int interleave, int clip)
{
    int ch, i;
    for (ch = 0; ch < s->channels; ++ch) {
        const float *prev = s->prev_buf + s->prev_start + (size_t)s->size1 * (size_t)ch;
        int clipped = 0;
        for (i = 0; i < count; ++i) {
            float p = prev[i];
            if (clip) p = orc_clip_value(p, &clipped);
            if (interleave) dst[((size_t)offset + (size_t)i) * (size_t)s->channels + (size_t)ch] = p;
            else            dst[(size_t)ch * (size_t)channel_stride + (size_t)offset + (size_t)i] = p;
        }
        s->has_clipped |= clipped;
    }
    s->prev_start += count;
    s->current_position += count;
}

int orc_stream_has_clipped(const orc_stream *s) { return s->has_clipped; }
int64_t orc_stream_position(const orc_stream *s) { return s->current_position; }

/* ------------------------------------------------------------------ batch driver */
long orc_synth_stream_planar(int channels, int size0, int size1, long frames, const uint8_t *flags,
                             const float *spectra, float *pcm, long pcm_stride, int clip)
{
    orc_stream *s = orc_stream_create(channels, size0, size1);
    orc_mdct *m0 = orc_mdct_create(size0), *m1 = orc_mdct_create(size1);
    float *buf2 = (float *)malloc(sizeof(float) * (size_t)size1);
    long f, total = 0;
    int half1 = size1 / 2, ch;
    for (f = 0; f < frames; ++f) {
        int bf = flags[f] & 1, pf = (flags[f] >> 1) & 1, nf = (flags[f] >> 2) & 1;
        int bs = bf ? size1 : size0;
        orc_packet_info info;
        float *cur = orc_stream_next_buffer(s);
        int avail;
        orc_get_packet_info(size0, size1, bf, pf, nf, &info);
        for (ch = 0; ch < channels; ++ch) {
            float *span = cur + (size_t)ch * (size_t)size1;
            memset(span, 0, sizeof(float) * (size_t)size1); /* Mapping.cs:117 */
            memcpy(span, spectra + ((size_t)f * (size_t)channels + (size_t)ch) * (size_t)half1,
                   sizeof(float) * (size_t)(bs / 2));
            orc_mdct_reverse(bf ? m1 : m0, span, buf2);
        }
        if (orc_stream_read_next_packet(s, 1, &info, -1, 0) < 0) { total = -1; break; }
        avail = orc_stream_available(s);
        if (avail > 0) {
            orc_stream_store(s, pcm, total, avail, pcm_stride, 0, clip);
            total += avail;
        }
    }
    free(buf2);
    orc_mdct_destroy(m0);
    orc_mdct_destroy(m1);
    orc_stream_destroy(s);
    return total;
}

/* ------------------------------------------------------------------ a floored stream, packet by packet, in C
 * bench.py's CPU baseline of the FUSED workloads (configs[2]/[3]/[4] of BASELINE.json): the packets of ONE stream as the
 * CPU stage leaves them at Mapping.cs:163 go through Residue2's de-interleave (Residue2.cs:42-51), the tail of
 * Mapping.DecodePacket (Mapping.cs:166-195: orc_apply_coupling in reverse step order, orc_floor1_apply + orc_mdct_reverse
 * per channel, or a cleared half block for a silent one), ReadNextPacket / OverlapBuffers (StreamDecoder.cs:640-694, 764-791)
 * and StoreContiguous, exactly as tests/helpers.py drives the same functions from Python -- with the Mdct tables warm
 * (the reference caches them per block size, Mdct.cs:13,23-27; orc_mapping_synth builds them per call).
 * flags[p]: bits 0..2 window flags, 0x08 EOS, 0x10 not decoded, 0x20 residue is the interleaved Residue2 vector,
 * 0x40 already floored (no coupling, no floor).  Mapping m: floor of channel c = floor_of_channel[m * channels + c],
 * coupling steps [coupling_off[m], coupling_off[m + 1]).  Returns samples per channel written (planar, pcm_stride). */
long orc_synth_stream_floored(int channels, int size0, int size1, long n_packets, const uint8_t *flags,
                              const uint8_t *mapping, const int64_t *granule, const int64_t *residue_offset,
                              const float *residue, const orc_floor1 *floors, const int *floor_of_channel,
                              const uint8_t *coupling_mag, const uint8_t *coupling_ang, const int *coupling_off,
                              const int16_t *posts, const uint8_t *post_count, float *pcm, long pcm_stride, int clip)
{
    orc_stream *s = orc_stream_create(channels, size0, size1);
    orc_mdct *m0 = orc_mdct_create(size0), *m1 = orc_mdct_create(size1);
    float *buf2 = (float *)malloc(sizeof(float) * (size_t)size1);
    int *ip = (int *)malloc(sizeof(int) * 64);
    long p, total = 0;
    int ch, i;
    for (p = 0; p < n_packets; ++p) {
        const int f = flags[p];
        const int bf = f & 1, pf = (f >> 1) & 1, nf = (f >> 2) & 1, eos = (f >> 3) & 1;
        const int bs = bf ? size1 : size0, half = bs / 2;
        orc_packet_info info;
        int avail;
        if (f & 0x10) { /* DecodeNextPacket returned null (StreamDecoder.cs:758-761) */
            orc_stream_read_next_packet(s, 0, NULL, -1, eos);
            continue;
        }
        {
            float *cur = orc_stream_next_buffer(s);
            const float *src = residue + residue_offset[p];
            const int m = mapping[p];
            orc_get_packet_info(size0, size1, bf, pf, nf, &info);
            for (ch = 0; ch < channels; ++ch) memset(cur + (size_t)ch * (size_t)size1, 0, sizeof(float) * (size_t)size1); /* Mapping.cs:117 */
            if (f & 0x20) {
                orc_residue2_deinterleave(src, half, channels, cur, size1);
            } else {
                for (ch = 0; ch < channels; ++ch)
                    memcpy(cur + (size_t)ch * (size_t)size1, src + (size_t)ch * (size_t)half, sizeof(float) * (size_t)half);
            }
            if (f & 0x40) {
                for (ch = 0; ch < channels; ++ch) orc_mdct_reverse(bf ? m1 : m0, cur + (size_t)ch * (size_t)size1, buf2);
            } else {
                for (i = coupling_off[m + 1] - 1; i >= coupling_off[m]; --i) /* Mapping.cs:166-172 */
                    orc_apply_coupling(cur + (size_t)coupling_mag[i] * (size_t)size1, cur + (size_t)coupling_ang[i] * (size_t)size1, half, 1);
                for (ch = 0; ch < channels; ++ch) { /* :180-195 */
                    float *span = cur + (size_t)ch * (size_t)size1;
                    const long rec = p * channels + ch;
                    if (post_count[rec] > 0) {
                        int k;
                        for (k = 0; k < 64; ++k) ip[k] = posts[rec * 64 + k];
                        orc_floor1_apply(&floors[floor_of_channel[m * channels + ch]], ip, post_count[rec], bs, span);
                        orc_mdct_reverse(bf ? m1 : m0, span, buf2);
                    } else {
                        memset(span, 0, sizeof(float) * (size_t)half);
                    }
                }
            }
        }
        if (orc_stream_read_next_packet(s, 1, &info, granule ? granule[p] : -1, eos) < 0) continue; /* :777-778 throws: that Read is lost */
        avail = orc_stream_available(s);
        if (avail > 0) {
            orc_stream_store(s, pcm, total, avail, pcm_stride, 0, clip);
            total += avail;
        }
    }
    free(ip);
    free(buf2);
    orc_mdct_destroy(m0);
    orc_mdct_destroy(m1);
    orc_stream_destroy(s);
    return total;
}
