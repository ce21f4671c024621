# one line per bench workload on one box (GPU side):  bash scripts/workload_table.sh > gpurun_out/workload_table.txt
for w in cfg2 cfg1 cfg3 cfg5 cfg2se inkyaml dumbyaml; do
  dt=bf16; [ $w = cfg5 ] && dt=fp16
  python bench.py --workload $w --dtype $dt --steps 12 --warmup 4 --no-cpu-baseline --no-h2d --no-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(f\"$w $dt  {d['value']:.1f} patches/s  {d['ms_per_step']:.2f} ms/step  batch {d['config']['global_batch']}  dominant {r['kernel']} frac {r['frac']:.3f}\")"
done
