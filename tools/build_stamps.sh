#!/bin/bash
# Diagnostic build of the library with the per-phase cycle stamps (-DMAVA_STAMPS): tools/libmavahip_stamps.so, objects in /tmp.
# Use with MAVA_LIB_PATH=tools/libmavahip_stamps.so (tools/train_stamps.py, tools/*_stamps.py).
set -e
cd "$(dirname "$0")/.."
O=/tmp/mava_stamps_obj; mkdir -p $O
pids=()
for f in mava_amd/csrc/*.hip mava_amd/csrc/*.cpp; do
  b=$(basename $f); b=${b%.*}
  if [[ $f == *.hip ]]; then A="--offload-arch=gfx950"; else A=""; fi
  if [ ! -f $O/$b.o ] || [ $f -nt $O/$b.o ] || [ -n "$(find mava_amd/csrc -name '*.h' -newer $O/$b.o)" ]; then
    /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -DMAVA_STAMPS -I mava_amd/csrc -I include $A -c $f -o $O/$b.o &
    pids+=($!)
    if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/libmavahip_stamps.so $O/*.o -ldl
ls -la tools/libmavahip_stamps.so
