# var_kern.sh <variant>: per-kernel ms per step (alone pass) of the default library and of a variant
run() { env $2 NVT_READBACK_TIMEOUT=60 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['ms_per_step'],3),d['gpu_busy_ms_per_step'],d['roofline']['per_kernel_ms_per_step'])"; }
V=$PWD/nvtabular_amd/libnvt_hip_$1.so
run base ""; run $1 NVT_HIP_LIB=$V; run base ""; run $1 NVT_HIP_LIB=$V
