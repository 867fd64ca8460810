// Floor0's two transcendental functions (Floor0.cs:103-111, 188-219: Math.Cos of the LSP coefficients and of the bark angles, Math.Exp
// of the curve), as floor0.hip evaluates them.  A header of its own so that tests/test_floor0_math_cpu.py can compile the very same
// source for the host (g++, -ffp-contract=off) and hold it to the C library's double-precision functions rounded to float.
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VPZ_F0_FN __device__ __forceinline__
#else
#define VPZ_F0_FN static inline
#endif

namespace vpz {

// (float)exp((double)x) and (float)cos((double)x) without the library's generality: the curve kernels are bound by instruction issue,
// and the library's double-precision exp and cos are 50 and 180 instructions of it.  Both are series whose truncation error is below
// a double's last place on the reduced argument (|r| <= ln2 / 2: 13 terms; |r| <= pi / 4: 9 terms each), evaluated in double and
// rounded ONCE to float like the expressions they replace; the argument reductions are the classic two-constant ones, exact for the
// quotients that occur.  Every caller in this file uses these, so the fused and the separate Floor0 routes agree bit for bit.
VPZ_F0_FN float exp_rounded_once(float xf)
{
    double x = (double)xf;
    x = xf < -800.0f ? -800.0 : (xf > 800.0f ? 800.0 : x);  // (the exponential of +-800 is 0 / infinity as a double already; NaN passes)
    const double n = __builtin_rint(x * 1.44269504088896340736);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);  // ln 2, leading 32 bits
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);         // ... the rest
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return (float)ldexp(p, (int)n);
}

// four at a time: the series' constants live in registers once, the four chains are independent (same operations per value, so the
// same results as four calls)
VPZ_F0_FN void exp_rounded_once_x4(const float (&xf)[4], float (&out)[4])
{
    double r[4], n[4], p[4];
    for (int m = 0; m < 4; ++m) {
        double x = (double)xf[m];
        x = xf[m] < -800.0f ? -800.0 : (xf[m] > 800.0f ? 800.0 : x);
        n[m] = __builtin_rint(x * 1.44269504088896340736);
        r[m] = __builtin_fma(-n[m], 6.93147180369123816490e-01, x);
        r[m] = __builtin_fma(-n[m], 1.90821492927058770002e-10, r[m]);
        p[m] = 1.0 / 6227020800.0;
    }
    const double c[13] = {1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0,
                          1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0};
    for (int i = 0; i < 13; ++i)
        for (int m = 0; m < 4; ++m) p[m] = __builtin_fma(p[m], r[m], c[i]);
    for (int m = 0; m < 4; ++m) out[m] = (float)ldexp(p[m], (int)n[m]);
}

VPZ_F0_FN float cos_rounded_once(float xf)
{
    if (!(fabsf(xf) <= 64.0f)) return (float)cos((double)xf);  // (LSP coefficients and bark angles lie in [0, pi]; anything far goes the long way)
    const double x = (double)xf;
    const double k = __builtin_rint(x * 0.63661977236758134308);
    double r = __builtin_fma(-k, 1.57079632673412561417e+00, x);  // pi / 2, leading 33 bits: k times it is exact
    r = __builtin_fma(-k, 6.07710050650619224932e-11, r);         // ... the rest
    const double z = r * r;
    double c = 1.0 / 20922789888000.0;  // 1 / 16!
    c = __builtin_fma(c, z, -1.0 / 87178291200.0);
    c = __builtin_fma(c, z, 1.0 / 479001600.0);
    c = __builtin_fma(c, z, -1.0 / 3628800.0);
    c = __builtin_fma(c, z, 1.0 / 40320.0);
    c = __builtin_fma(c, z, -1.0 / 720.0);
    c = __builtin_fma(c, z, 1.0 / 24.0);
    c = __builtin_fma(c, z, -0.5);
    c = __builtin_fma(c, z, 1.0);
    double t = 1.0 / 355687428096000.0;  // 1 / 17!
    t = __builtin_fma(t, z, -1.0 / 1307674368000.0);
    t = __builtin_fma(t, z, 1.0 / 6227020800.0);
    t = __builtin_fma(t, z, -1.0 / 39916800.0);
    t = __builtin_fma(t, z, 1.0 / 362880.0);
    t = __builtin_fma(t, z, -1.0 / 5040.0);
    t = __builtin_fma(t, z, 1.0 / 120.0);
    t = __builtin_fma(t, z, -1.0 / 6.0);
    t = __builtin_fma(t * z, r, r);  // sin r = r + r z (...)
    const int q = (int)k;  // cos(r + q pi / 2): cos r, -sin r, -cos r, sin r
    const double v = (q & 1) ? t : c;
    return (float)(((q + 1) & 2) ? -v : v);
}

}  // namespace vpz
