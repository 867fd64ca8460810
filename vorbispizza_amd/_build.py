"""In-tree build of libvorbispizza_synth.so (hipcc, gfx950 only).

The built library lives in vorbispizza_amd/lib/ (git-ignored, but shipped to the GPU box by
gpurun).  No JIT cache, no torch extension machinery: plain `hipcc -shared -fPIC`.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libvorbispizza_synth.so")
OBJ_DIR = os.path.join(LIB_DIR, "obj")

# translation unit -> extra flags.  imdct_exact.hip and the host-side table / state-machine code
# must round every f32 multiply/add separately, like the reference's JIT output.
SOURCES = {
    "imdct_fast.hip": [],
    "imdct_exact.hip": ["-ffp-contract=off"],
    "synth_kernels.hip": [],
    "synth_dual.hip": [],
    "synth_pairs.hip": [],   # synth_dual.hip once more, for channel pairs of streams with 4, 6, 8, ... channels
    "synth_big.hip": [],
    "floor0.hip": ["-ffp-contract=off"],
    "vpz_context.hip": ["-ffp-contract=off"],
    "vpz_decoder.hip": ["-ffp-contract=off"],
}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-constant-logical-operand", "-fno-strict-aliasing"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


def _deps(src):
    deps = [src]
    if os.path.basename(src) == "synth_pairs.hip":
        deps.append(os.path.join(CSRC, "synth_dual.hip"))
    for name in os.listdir(CSRC):
        if name.endswith(".hpp"):
            deps.append(os.path.join(CSRC, name))
    deps.append(os.path.join(_HERE, "..", "include", "vorbispizza_synth.h"))
    return deps


HOST_DIR = os.path.join(_HERE, "host")
HOST_LIB_PATH = os.path.join(LIB_DIR, "libvorbispizza_host.so")


def build_host(force=False, verbose=False):
    """Compile the C++ host side with g++: the CPU front end (Ogg + Vorbis setup + entropy decode) and
    the VorbisReader / StreamDecoder.Read mirror, which calls the C ABI of libvorbispizza_synth.so."""
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = [os.path.join(HOST_DIR, "vorbis_front.cpp"), os.path.join(HOST_DIR, "vorbis_reader.cpp"),
            os.path.join(HOST_DIR, "vorbis_multi.cpp")]
    inc = os.path.join(_HERE, "..", "include")
    deps = srcs + [os.path.join(inc, "vorbispizza_front.h"), os.path.join(inc, "vorbispizza_reader.h"),
                   os.path.join(inc, "vorbispizza_multi.h"), os.path.join(inc, "vorbispizza_synth.h"), LIB_PATH]
    stale = force or not os.path.exists(HOST_LIB_PATH) or any(
        os.path.exists(d) and os.path.getmtime(d) > os.path.getmtime(HOST_LIB_PATH) for d in deps)
    if stale:
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-pthread",
               "-o", HOST_LIB_PATH] + srcs + ["-L" + LIB_DIR, "-lvorbispizza_synth", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return HOST_LIB_PATH


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link the C-ABI shared library, then the
    C++ host library that sits above it."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    # (diagnostic builds: VPZ_EXTRA_HIPCC_FLAGS=-DVPZ_STAMPS; a change of flags rebuilds everything)
    extra_env = os.environ.get("VPZ_EXTRA_HIPCC_FLAGS", "")
    flags_file = os.path.join(OBJ_DIR, "flags.txt")
    if (open(flags_file).read() if os.path.exists(flags_file) else "") != extra_env:
        force = True
        with open(flags_file, "w") as f:
            f.write(extra_env)
    objs, relink, jobs = [], force or not os.path.exists(LIB_PATH), []
    for name, extra in SOURCES.items():
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            raise RuntimeError("missing source " + src)
        obj = os.path.join(OBJ_DIR, name.replace(".hip", ".o"))
        stale = force or not os.path.exists(obj) or any(
            os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps(src))
        if stale:
            jobs.append([hipcc] + COMMON + extra + extra_env.split() + ["-c", src, "-o", obj])
            relink = True
        objs.append(obj)
    if jobs:  # the translation units are independent: compile them side by side (a handful of hipcc processes)
        from concurrent.futures import ThreadPoolExecutor

        def compile_one(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)

        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as pool:
            list(pool.map(compile_one, jobs))
    if relink:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    build_host(force, verbose)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
