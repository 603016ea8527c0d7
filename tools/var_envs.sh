# var_envs.sh "ENV=V ENV2=W" ...: the bench (headline only) once per environment setting, base first and last
run() { env $2 NVT_READBACK_TIMEOUT=60 timeout 300 python bench.py --steps ${STEPS:-12} --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['ms_per_step'],3),d['gpu_busy_ms_per_step'],{k:v['ms_per_step'] for k,v in d['roofline']['per_family'].items()}, d.get('parity',{}).get('parity_ok'))"; }
run base ""
for v in "$@"; do run "$v" "$v"; done
run base ""
