#!/bin/bash
# same-box A/B of environment knobs on the default bench step:  bash scripts/ab_bench.sh NAME "ENV=.. ENV=.." [NAME "ENV.."]...
# prints ms/step and the hbm-block families named in $AB_FAMILIES (comma separated)
mkdir -p gpurun_out
while [ $# -gt 0 ]; do
  name=$1; envs=$2; shift 2
  env $envs python bench.py --steps ${AB_STEPS:-20} --warmup 6 --no-cpu-baseline --no-pmc --no-h2d ${AB_FLAGS:---no-kernel-timing} > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - "$name" <<'PY'
import json, os, sys
name = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/ab_{name}.json").read().strip().splitlines()[-1])
except Exception as e:
    print(name, "FAILED", e, open(f"gpurun_out/ab_{name}.err").read()[-600:]); sys.exit(0)
fam = [f for f in os.environ.get("AB_FAMILIES", "").split(",") if f]
extra = ""
if d.get("hbm"):
    extra = " ".join(f"{f}={d['hbm'][f]['ms_per_step']:.3f}" for f in fam if f in d["hbm"])
print(f"{name:24s} {d['ms_per_step']:.3f} ms  {d['value']:.2f} patches/s  {extra}", flush=True)
PY
done
