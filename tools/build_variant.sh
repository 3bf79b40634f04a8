#!/bin/bash
# usage: tools/build_variant.sh SOURCE.hip NAME "-DFLAG=1 ..."  -- relink the in-tree library with ONE translation unit rebuilt with extra
# flags -> nmfgpu_amd/lib/variants/NAME.so (select with NMFAMD_LIBRARY).  The in-tree library must be built already.
# VARIANT_OF=diag: a variant of the measurement build (objects of lib/obj_diag, -DNMFAMD_DIAG_BUILD): stamped kernels, tuning_env switches
src=$1; name=$2; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/nmfgpu_amd/lib/obj
if [ "$VARIANT_OF" = diag ]; then obj=$root/nmfgpu_amd/lib/obj_diag; flags="$flags -DNMFAMD_DIAG_BUILD"; fi
mkdir -p $root/nmfgpu_amd/lib/variants /tmp/variant_$name
extra="-fno-slp-vectorize"      # as build.py DEVICE_FLAGS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DNMFGPU_EXPORTING -Wno-unknown-pragmas -Wno-unused-function -Wno-unused-result \
  $extra $flags -x hip -c $root/nmfgpu_amd/csrc/$src -o /tmp/variant_$name/${src%.*}.o || exit 1
objs=""
for o in $obj/*.o; do
  if [ "$(basename $o)" = "${src%.*}.o" ]; then objs="$objs /tmp/variant_$name/${src%.*}.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o $root/nmfgpu_amd/lib/variants/$name.so $objs && echo built $name
