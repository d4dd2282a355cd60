#!/bin/bash
# Same-box A/B: tools/libmavahip_prev.so = the library as built now, except that the sources named on the command line are
# taken from git HEAD (e.g. `bash tools/build_prev.sh ppo_train_w8.hip`).  Use with MAVA_LIB_PATH=tools/libmavahip_prev.so.
set -e
cd "$(dirname "$0")/.."
O=/tmp/mava_prev_obj; rm -rf $O; mkdir -p $O/src
cp mava_amd/csrc/*.o $O/
for f in "$@"; do
  git show HEAD:mava_amd/csrc/$f > $O/src/$f
  b=${f%.*}
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -I mava_amd/csrc -I include --offload-arch=gfx950 -ffp-contract=fast -c $O/src/$f -o $O/$b.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/libmavahip_prev.so $O/*.o -ldl
ls -la tools/libmavahip_prev.so
