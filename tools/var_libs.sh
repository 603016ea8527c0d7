# var_libs.sh <variant>...: the default library and every libnvt_hip_<variant>.so once, per-family ms (one box)
run() { NVT_READBACK_TIMEOUT=60 timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['ms_per_step'],3),d['gpu_busy_ms_per_step'],{k:v['ms_per_step'] for k,v in d['roofline']['per_family'].items()})"; }
run base
for v in "$@"; do NVT_HIP_LIB=$PWD/nvtabular_amd/libnvt_hip_$v.so run $v; done
run base
