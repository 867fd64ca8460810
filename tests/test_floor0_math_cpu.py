"""Floor0's transcendental functions as the kernels evaluate them (vorbispizza_amd/csrc/floor0_math.hpp: series in double,
rounded once) against the C library's double-precision exp / cos rounded to float -- what the oracle computes and what the
reference's Math.Exp / Math.Cos give (Floor0.cs:103-111, 188-219).  The header is compiled for the host as it stands."""
import os
import subprocess
import textwrap

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "vorbispizza_amd", "csrc")

PROGRAM = textwrap.dedent(r"""
    #include "floor0_math.hpp"
    #include <cstdint>
    #include <cstdio>
    #include <cstring>
    static bool same(float a, float b) { return (a != a && b != b) || !memcmp(&a, &b, 4); }
    int main()
    {
        uint64_t st = 88172645463325252ull;
        auto rnd = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
        long bad_e = 0, bad_c = 0;
        const long n = 3000000;
        for (long i = 0; i < n; ++i) {
            const float x = (float)(rnd() * 220.0 - 115.0);     // exp: denormal results up to overflow
            if (!same(vpz::exp_rounded_once(x), (float)exp((double)x))) ++bad_e;
            const float y = (float)(rnd() * 80.0 - 40.0);       // cos: the series' range (|x| <= 64) and beyond the LSP range
            if (!same(vpz::cos_rounded_once(y), (float)cos((double)y))) ++bad_c;
            const float z = (float)(rnd() * 3.2);               // ... and where the coefficients live
            if (!same(vpz::cos_rounded_once(z), (float)cos((double)z))) ++bad_c;
        }
        const float special[] = {0.0f, -0.0f, 1e30f, -1e30f, 88.7f, 89.0f, -87.0f, -103.0f, -104.0f, -150.0f, INFINITY, -INFINITY, NAN,
                                 63.9f, 64.0f, 65.0f, 1e10f, 3.14159274f, 1.57079637f, 1e-30f, -1e-30f, 800.0f, -800.0f, 801.0f};
        long bad_s = 0;
        for (float x : special) {
            if (!same(vpz::exp_rounded_once(x), (float)exp((double)x))) { printf("exp %a\n", x); ++bad_s; }
            if (!same(vpz::cos_rounded_once(x), (float)cos((double)x))) { printf("cos %a\n", x); ++bad_s; }
        }
        printf("%ld %ld %ld\n", bad_e, bad_c, bad_s);
        return 0;
    }
""")


def test_series_equal_the_c_library_rounded_to_float(tmp_path):
    src = tmp_path / "f0math.cpp"
    src.write_text(PROGRAM)
    exe = tmp_path / "f0math"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", CSRC, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    bad_e, bad_c, bad_s = (int(v) for v in out[-1].split())
    # the series are good to a double's last places: a float result can only differ where the exact value lies within ~1e-16 of a
    # rounding boundary -- one argument in 10^8; none among these
    assert (bad_e, bad_c, bad_s) == (0, 0, 0), out
