#!/bin/bash
# Builds casualhdrsplat_amd/libhdrsplat_<name>.so = the product sources with extra -D flags on some translation units (A/B
# and ablation experiments; the variants travel to the GPU box with the snapshot).
# usage: bash scripts/build_variant.sh <name> <tu[,tu...]: binning|render|preprocess|api|spline> "<flags>"
set -e
cd "$(dirname "$0")/../casualhdrsplat_amd/csrc"
name=$1; tus=${2//,/ }; flags=$3
make -s all >/dev/null
objs=""
for o in api preprocess binning render spline; do
  if [[ " $tus " == *" $o "* ]]; then
    contract=off; [ "$o" = render ] && contract=fast
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -std=c++17 -Wall -Wno-unused-function -ffp-contract=$contract $flags -c $o.hip -o /tmp/${o}_$name.o
    objs="$objs /tmp/${o}_$name.o"
  else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhdrsplat_$name.so $objs
echo built ../libhdrsplat_$name.so
