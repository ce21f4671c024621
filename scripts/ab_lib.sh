# same-box A/B of two builds of the library: csrc/librxunet_base.so (copy of the previous build) vs csrc/librxunet.so
#   bash scripts/ab_lib.sh [extra bench args]
B=multi-task-3d-resencoder-unet_amd/csrc
EXTRA="$@"
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-h2d ${EXTRA} 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d.get('hbm') or {}
print('$name', round(d['ms_per_step'],3), d['final_loss'], {k: round(v['ms_per_step'],3) for k,v in h.items() if k in ('in_act_bwd(colreduce+apply)','in_act_bwd_apply','in_act_fwd','in_fwd(stats+apply)')})"; }
run base RX_LIBRARY=$PWD/$B/librxunet_base.so
run new RX_X=0
run base RX_LIBRARY=$PWD/$B/librxunet_base.so
run new RX_X=0
