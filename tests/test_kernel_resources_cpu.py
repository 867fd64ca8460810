"""Build check (no GPU): no kernel of the library may spill to scratch memory.

Round 1 shipped synth_kernel instantiations with 32-124 B/lane of scratch (interleaved output for channel counts
other than 2, every 512 / 1024 block-size variant) that nobody had looked at because only three instantiations were
benchmarked.  This test compiles every translation unit with the compiler's `-Rpass-analysis=kernel-resource-usage`
remarks (tools/kernel_resources.py) and fails on ScratchSize > 0 -- for ALL kernels, not only the hot ones -- and on
an occupancy below what the LDS budget of synth_kernel is sized for."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kernels():
    sys.path.insert(0, ROOT)
    from vorbispizza_amd import _build
    if not (shutil.which(_build._hipcc()) or os.path.exists(_build._hipcc())):
        pytest.skip("no hipcc")
    import kernel_resources as kr
    out = {}
    for src, extra in kr.all_units():
        for k in kr.analyse(src, extra):
            out[k["name"]] = dict(k, unit=os.path.basename(src))
    return out


def test_no_kernel_uses_scratch_memory(kernels):
    assert len(kernels) >= 30
    spilling = {k["name"]: k["scratch"] for k in kernels.values() if k.get("scratch", 0) > 0}
    assert not spilling, "kernels with scratch memory (bytes per lane): %r" % spilling


def test_every_synth_kernel_instantiation_is_built_and_keeps_two_workgroups_per_cu(kernels):
    synth = [k for k in kernels.values() if "synth_kernel" in k["name"]]
    # <floor?, planar / interleaved / stereo pair, general sizes?, group mode?, float32 / int16 samples>
    assert len(synth) == 2 * 3 * 2 * 2 * 2
    for k in synth:
        assert k["vgprs"] <= 128, k["name"]            # 4 waves per SIMD: 512 / 4
        assert k["lds"] <= 80 * 1024, (k["name"], k["lds"])  # two 8-wave workgroups per CU (160 KiB)
        assert k.get("occupancy", 4) >= 4, k["name"]


def test_stereo_fast_path_instantiations_and_their_budget(kernels):
    dual = [k for k in kernels.values() if "synth_dual_kernel" in k["name"]]
    # <floor?, Residue2-interleaved / planar input, planar / interleaved output, float32 / int16 samples> + the floored ones once
    # more for 16-bit residue read in place (kI16)
    assert len(dual) == 2 * 2 * 2 * 2 + 2 * 2 * 2
    for k in dual:
        assert k["vgprs"] <= 256, k["name"]                   # 2 waves per SIMD: 512 / 2
        assert k["lds"] <= 80 * 1024, (k["name"], k["lds"])   # two 4-wave workgroups per CU (160 KiB)
        assert k.get("occupancy", 2) >= 2, k["name"]


def test_the_pair_route_is_the_stereo_kernel_once_more_with_its_budget(kernels):
    pairs = [k for k in kernels.values() if "synth_pairs_kernel" in k["name"]]
    # <floor?, Residue2-interleaved / planar input, planar / interleaved output, float32 / int16 samples> (float32 residue only)
    assert len(pairs) == 2 * 2 * 2 * 2
    for k in pairs:
        assert k["unit"] == "synth_pairs.hip"
        assert k["vgprs"] <= 256, k["name"]
        assert k["lds"] <= 80 * 1024, (k["name"], k["lds"])
        assert k.get("occupancy", 2) >= 2, k["name"]


def test_the_kernel_for_4096_and_8192_blocks_keeps_its_budget(kernels):
    big = [k for k in kernels.values() if "synth_big_kernel" in k["name"]]
    assert len(big) == 2 * 2 * 2  # <floor?, float32 / int16 samples, 4096 (tails in LDS) / 8192 (tails in global memory, eight waves)>
    for k in big:
        assert k["vgprs"] <= 256, k["name"]  # two waves per SIMD where the LDS allows them (4096: two 4-wave workgroups per CU)
        # the STATIC part (the two sizes' tables; the waves' areas are dynamic): 8192 keeps its twiddle tables here too, beside seven waves
        tail_g = "Lb1EEE" in k["name"] or k["name"].rstrip(">").endswith("true")
        assert k["lds"] <= (38 if tail_g else 28) * 1024, (k["name"], k["lds"])
        assert k["lds"] + (7 * 17440 if tail_g else 2 * 53376) <= 160 * 1024 + (0 if tail_g else k["lds"]), (k["name"], k["lds"])


@pytest.mark.parametrize("flags", [["-DVPZ_STAMPS"], ["-DVPZ_WAVE_TIMES"], ["-DVPZ_TUNING"], ["-DVPZ_STAMPS", "-DVPZ_WAVE_TIMES", "-DVPZ_TUNING"],
                                   ["-DVPZ_DUAL_STEADY_BIT", "-DVPZ_GROUP_STEADY_BIT", "-DVPZ_PAIR_STORES_NT"],
                                   ["-DVPZ_DUAL_TAIL_REGS", "-DVPZ_DUAL_MIRROR_VALU"], ["-DVPZ_DUAL_NO_SLOPE_REGS", "-DVPZ_DUAL_NO_TW_REGS"]])
def test_the_diagnostic_builds_still_parse(flags):
    """The build switches of the fused kernels (phase stamps, wave clocks, tuning switches, the alternatives kept for A/B runs) are
    compiled only when somebody needs them -- and rot unseen: a syntax-only device pass of the three translation units per set."""
    import subprocess
    sys.path.insert(0, ROOT)
    from vorbispizza_amd import _build
    hipcc = _build._hipcc()
    if not (shutil.which(hipcc) or os.path.exists(hipcc)):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "vorbispizza_amd", "csrc")
    for unit in ("synth_dual.hip", "synth_pairs.hip", "synth_kernels.hip", "floor0.hip", "synth_big.hip"):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "--cuda-device-only", "-Wno-unused-function",
                            "-I", csrc, "-I", os.path.join(ROOT, "include")] + flags + [os.path.join(csrc, unit)],
                           capture_output=True, text=True)
        assert r.returncode == 0, (unit, flags, r.stderr[-2000:])
