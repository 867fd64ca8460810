#!/bin/bash
# Where do the fused kernels' LDS bank conflicts come from?  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS of one
# kbench_synth workload with the phases of the frame loop switched off one at a time (VPZ_SYNTH_ABLATE: wrong results, right
# counters) -- needs the TUNING build of the libraries (VPZ_EXTRA_HIPCC_FLAGS=-DVPZ_TUNING, copied to vorbispizza_amd/lib_ab/tuning).
#   usage: bash tools/pmc_lds_conflicts.sh <out.txt> <which: real|floor> <kernel substring> <ablate values...>
OUT=$1; W=$2; K=$3; shift; shift; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export VPZ_LIB_DIR="$PWD/vorbispizza_amd/lib_ab/tuning"
mkdir -p "$(dirname "$OUT")"
for ab in "$@"; do
  D=gpurun_out/r4/ldsc_$ab
  rm -rf $D
  VPZ_SYNTH_ABLATE=$ab rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d $D --output-format csv -- python3 tools/kbench_synth.py --which $W --steps 3 > $D.log 2>&1
  echo "== $W, VPZ_SYNTH_ABLATE=$ab" >> "$OUT"
  python3 tools/pmc_summary.py $D "$K" >> "$OUT"
done
cat "$OUT"
