# var_env.sh "<ENV=val ...>": A/B/A/B of the headline step with / without the given environment
run() { env $2 NVT_READBACK_TIMEOUT=60 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['ms_per_step'],3),d['gpu_busy_ms_per_step'],{k:v['ms_per_step'] for k,v in d['roofline']['per_family'].items()})"; }
run base ""; run "$1" "$1"; run base ""; run "$1" "$1"
