#!/bin/bash
# Round profile of the default bench command (run ON the GPU box through gpurun):
#   bash scripts/profile_round.sh r02_a
# 1. plain bench line, 2. rocprofv3 --kernel-trace --stats of the same command, 3./4. two separate --pmc passes (FETCH_SIZE,
# WRITE_SIZE) as MI355X_MICROARCH.md prescribes (never combined with trace domains).  Summaries land in gpurun_out/<tag>_*;
# copy what should be judged into profiles/.
set -o pipefail
TAG=${1:-r03_a}
OUT=gpurun_out
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 --through-trainer > $OUT/${TAG}_bench_cfg2.json 2> $OUT/${TAG}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-h2d > $OUT/${TAG}_kt.log 2>&1 || exit 1
cp $(find $OUT/${TAG}_kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_cfg2_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-h2d > $OUT/${TAG}_pmc_f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o w -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-h2d > $OUT/${TAG}_pmc_w.log 2>&1 || exit 1
python scripts/pmc_traffic.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write > $OUT/${TAG}_pmc_hbm_traffic.json
python scripts/step_timeline.py $OUT/${TAG}_kt > $OUT/${TAG}_step_timeline.txt
python scripts/kstats.py $OUT/${TAG}_bench_cfg2_kernel_stats.csv 25 30   # 5 warm-up + 20 timed steps under the tracer
# keep the merge-back small: the raw traces are large
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
