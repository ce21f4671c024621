#!/bin/bash
# Build librxunet from a git revision into csrc/librxunet_<name>.so for same-box A/B runs:
#   scripts/build_ref_lib.sh HEAD base   ->  RX_LIBRARY=$PWD/multi-task-3d-resencoder-unet_amd/csrc/librxunet_base.so python bench.py ...
set -e
rev=${1:-HEAD}; name=${2:-base}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" multi-task-3d-resencoder-unet_amd/csrc include | tar -x -C "$tmp"
cd "$tmp/multi-task-3d-resencoder-unet_amd/csrc"
objs=""
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-result -I"$tmp/include" -c "$f" -o "${f%.hip}.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/multi-task-3d-resencoder-unet_amd/csrc/librxunet_${name}.so" *.o
rm -rf "$tmp"
echo "$root/multi-task-3d-resencoder-unet_amd/csrc/librxunet_${name}.so"
