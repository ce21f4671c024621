#!/bin/bash
# ablation of conv_halo64ws (needs a -DRX_ABLATION=1 build of rx_conv_halo.hip as abl_librxunet.so at the repo root):
# RX_DBG bits: 1 no halo DMA, 2 no weight DMA, 4 no MFMA loops, 8 no stores
for m in ${ABL_MASKS:-0 1 2 3 4 8 10 11 12 15}; do
  echo "RX_DBG=$m: $(RX_LIBRARY=$PWD/abl_librxunet.so RX_DBG=$m python scripts/bench_conv.py "64->64@64" "128->128@32" fwd dgrad --iters 20 2>/dev/null | awk '{printf "%s %s %s us | ", $1, $2, $3}')"
done
