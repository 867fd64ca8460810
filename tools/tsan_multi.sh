#!/bin/bash
# The dispatcher tests with the HOST library built under ThreadSanitizer (vorbispizza_amd/lib_ab/tsan/, see the g++ line in HISTORY.md);
# races inside non-instrumented modules (the HIP runtime, Python) are ignored.   usage (through gpurun): bash tools/tsan_multi.sh [out]
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r4/tsan_multi.txt}
mkdir -p $(dirname $OUT)
export VPZ_LIB_DIR=$PWD/vorbispizza_amd/lib_ab/tsan
export TSAN_OPTIONS="ignore_noninstrumented_modules=1 halt_on_error=0 report_signal_unsafe=0 exitcode=0 log_path=$PWD/gpurun_out/r4/tsan_log"
rm -f gpurun_out/r4/tsan_log.*
timeout -k 10 600 setarch $(uname -m) -R env LD_PRELOAD=$(gcc -print-file-name=libtsan.so) python -m pytest tests/test_multi_gpu.py -x -q > $OUT 2>&1   # (-R: no address randomisation, this libtsan needs its mappings where it expects them)
tail -3 $OUT
ls gpurun_out/r4/tsan_log.* 2>/dev/null | head; cat gpurun_out/r4/tsan_log.* 2>/dev/null | grep -E "WARNING|#0|#1|#2|Location|Previous|vorbis_" | head -60
