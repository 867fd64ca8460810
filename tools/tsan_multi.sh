#!/bin/bash
# The dispatcher tests with the HOST library built under ThreadSanitizer (vorbispizza_amd/lib_ab/tsan/, see the g++ line in HISTORY.md);
# races inside non-instrumented modules (the HIP runtime, Python) are ignored.   usage (through gpurun): bash tools/tsan_multi.sh [out]
cd "$GRAFT_REPO_ROOT"
# the instrumented host library (built here: vorbispizza_amd/lib_ab/ is scratch and not tracked), next to a copy of the product's synth library
T=vorbispizza_amd/lib_ab/tsan
mkdir -p $T && cp vorbispizza_amd/lib/libvorbispizza_synth.so $T/ &&
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=thread -ffp-contract=off -fno-fast-math -Wall -pthread -o $T/libvorbispizza_host.so \
    vorbispizza_amd/host/vorbis_front.cpp vorbispizza_amd/host/vorbis_reader.cpp vorbispizza_amd/host/vorbis_multi.cpp \
    -Iinclude -L$T -lvorbispizza_synth '-Wl,-rpath,$ORIGIN' || exit 1
OUT=${1:-gpurun_out/r5/tsan_multi.txt}
mkdir -p $(dirname $OUT)
export VPZ_LIB_DIR=$PWD/vorbispizza_amd/lib_ab/tsan
export TSAN_OPTIONS="ignore_noninstrumented_modules=1 halt_on_error=0 report_signal_unsafe=0 exitcode=0 log_path=$PWD/gpurun_out/r5/tsan_log"
rm -f gpurun_out/r5/tsan_log.*
# (torch's own GPU start-up does not survive the preloaded libtsan: the one test that needs it stays out)
# (-R: no address randomisation, this libtsan needs its mappings where it expects them)
timeout -k 10 600 setarch $(uname -m) -R env LD_PRELOAD=$(gcc -print-file-name=libtsan.so) python -m pytest tests/test_multi_gpu.py tests/test_residue_i16_gpu.py -x -q -k "not device_memory" > $OUT 2>&1
tail -3 $OUT
ls gpurun_out/r5/tsan_log.* 2>/dev/null | head; cat gpurun_out/r5/tsan_log.* 2>/dev/null | grep -E "WARNING|#0|#1|#2|Location|Previous|vorbis_" | head -60
