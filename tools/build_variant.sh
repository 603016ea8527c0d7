#!/bin/bash
# build_variant.sh <name> <file.hip[,file2.hip,...]> <extra flags...>: libnvt_hip_<name>.so with the
# listed objects rebuilt with the extra flags (the rest are the default build's objects).  Variant
# objects are built with -DNVT_AB_SWITCHES: the run-time A / B switches between kernel variants
# (nvt_common.hpp ab_env) exist only there, the default library has none
set -e
cd $(dirname $0)/../nvtabular_amd/csrc
name=$1; srcs=${2//,/ }; shift 2
skip=""; new=""
for src in $srcs; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DNVT_AB_SWITCHES "$@" -c $src -o /tmp/variant_${name}_${src%.hip}.o
  skip="$skip ${src%.hip}.o"; new="$new /tmp/variant_${name}_${src%.hip}.o"
done
objs=""
for o in *.o; do case " $skip " in *" $o "*) ;; *) objs="$objs $o";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $new -ldl -o ../libnvt_hip_$name.so
echo built ../libnvt_hip_$name.so
