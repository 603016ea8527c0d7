# var_lib.sh <variant>: A/B/A/B of libnvt_hip_<variant>.so against the default library (one box)
run() { NVT_READBACK_TIMEOUT=60 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['ms_per_step'],3),d['gpu_busy_ms_per_step'],{k:v['ms_per_step'] for k,v in d['roofline']['per_family'].items()})"; }
V=$PWD/nvtabular_amd/libnvt_hip_$1.so
run base; NVT_HIP_LIB=$V run $1; run base; NVT_HIP_LIB=$V run $1
