# marginal cost of whole kernel families inside the overlapped step: the step with that family's launches SKIPPED
# (timing ablation only: results are garbage).  bash scripts/ab_skip.sh
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3))"; }
run full RX_X=0
run no_wgrad RX_ABLATION=1 RX_SKIP=wgrad
run no_inbwd_reduce RX_ABLATION=1 RX_SKIP=inbwd_reduce
run no_dgrad RX_ABLATION=1 RX_SKIP=dgrad
run no_wgrad_no_dgrad RX_ABLATION=1 RX_SKIP=wgrad,dgrad
run no_overlap RX_OVERLAP_WGRAD=0
run full2 RX_X=0
