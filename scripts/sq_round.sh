#!/bin/bash
# SQ counters of the isolated conv kernels (two --pmc passes each, counters only, as MI355X_MICROARCH.md prescribes):
#   bash scripts/sq_round.sh r03_a   -> gpurun_out/<tag>_sq_conv_kernels.txt, <tag>_sq_wgrad.txt
TAG=${1:-r03_a}
OUT=gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
for what in "conv_kernels fwd dgrad dgradacc" "wgrad wgrad"; do
  set -- $what
  name=$1; shift
  : > $OUT/${TAG}_sq_${name}.txt
  for pass in 1 2; do
    if [ $pass = 1 ]; then P=$P1; else P=$P2; fi
    rm -rf /tmp/sq_$name
    rocprofv3 --pmc $P --output-format csv -d /tmp/sq_$name -o s -- python3 scripts/bench_conv.py "$@" --iters 3 > /tmp/sq_${name}.log 2>&1
    echo "# pass $pass: --pmc $P  (scripts/bench_conv.py $* --iters 3)" >> $OUT/${TAG}_sq_${name}.txt
    python scripts/pmc_sq.py /tmp/sq_$name >> $OUT/${TAG}_sq_${name}.txt
  done
done
rm -rf /tmp/sq_conv_kernels /tmp/sq_wgrad
wc -l $OUT/${TAG}_sq_*.txt
