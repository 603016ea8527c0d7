#!/bin/bash
# build_variant.sh <name> <file.hip> <extra flags...>: libnvt_hip_<name>.so with one object rebuilt
set -e
cd $(dirname $0)/../nvtabular_amd/csrc
name=$1; src=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $src -o /tmp/variant_$name.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/variant_$name.o -ldl -o ../libnvt_hip_$name.so
echo built ../libnvt_hip_$name.so
