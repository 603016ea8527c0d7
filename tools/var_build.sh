#!/bin/bash
# A / B of a compile-time switch of nvt_dense_count.hip on the GPU box: tools/var_build.sh <outdir> "<flags A>" "<flags B>"
cd /root/repo; out=gpurun_out/$1; mkdir -p $out
run() { name=$1; timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extra > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    r=json.load(open("$out/$name.json")); print("$name", {k:round(r[k],2) for k in ("ms_per_step","gpu_busy_ms_per_step")}, {k:v for k,v in r["roofline"]["per_kernel_ms_per_step"].items() if "dense_count_h" in k})
except Exception as e: print("$name", "FAILED", e)
PY
}
build() { touch nvtabular_amd/csrc/nvt_dense_count.hip; make -C nvtabular_amd/csrc CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $1" > $out/build.log 2>&1 || tail -5 $out/build.log; }
build "$2"; run a1
build "$3"; run b1
build "$2"; run a2
build "$3"; run b2
