run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing --no-h2d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), d['final_loss'])"; }
run default RX_X=0
run lean_all RX_WGH_WS=0 RX_CH64WS=0 RX_NO_CH32P=1 RX_PLANAR_CAT=0
run lean_wgrad RX_WGH_WS=0
run lean_conv RX_CH64WS=0 RX_NO_CH32P=1 RX_PLANAR_CAT=0
run noplanar RX_PLANAR_CAT=0
run default2 RX_X=0
