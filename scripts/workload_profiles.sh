#!/bin/bash
# One bench line + one rocprofv3 kernel-stats CSV per NON-headline workload, on ONE box:
#   bash scripts/workload_profiles.sh r03          -> gpurun_out/<tag>_wl_<workload>.json / _kernel_stats.csv, <tag>_workload_table.txt
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT
: > $OUT/${TAG}_workload_table.txt
for spec in "cfg2 bf16" "cfg1 bf16" "cfg3 bf16" "cfg5 fp16" "cfg2se bf16" "inkyaml bf16" "dumbyaml bf16"; do
  set -- $spec
  wl=$1; dt=$2
  python bench.py --workload $wl --dtype $dt --steps 12 --warmup 5 --no-cpu-baseline --no-pmc --no-h2d > $OUT/${TAG}_wl_${wl}.json 2> $OUT/${TAG}_wl_${wl}.err || { echo "$wl FAILED" >> $OUT/${TAG}_workload_table.txt; continue; }
  if [ "$wl" != "cfg2" ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}_kt_$wl -o kt -- python3 bench.py --workload $wl --dtype $dt --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-h2d --no-pmc > $OUT/${TAG}_wl_${wl}_kt.log 2>&1 \
      && cp $(find /tmp/${TAG}_kt_$wl -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_wl_${wl}_kernel_stats.csv
    rm -rf /tmp/${TAG}_kt_$wl $OUT/${TAG}_wl_${wl}_kt.log
  fi
  python - "$wl" "$dt" "$OUT/${TAG}_wl_${wl}.json" >> $OUT/${TAG}_workload_table.txt <<'PY'
import json, sys
wl, dt, path = sys.argv[1:4]
d = json.loads(open(path).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"| {wl} | {dt} | {d['config']['global_batch']} | {d['value']:.1f} | {d['ms_per_step']:.2f} | `{r['kernel'].split(':')[-1]}` {r['frac']:.3f} |")
PY
done
cat $OUT/${TAG}_workload_table.txt
