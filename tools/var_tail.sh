run() { NVT_READBACK_TIMEOUT=60 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['ms_per_step'],3),d['gpu_busy_ms_per_step'])"; }
run batch; NVT_NO_TAIL_BATCH=1 run nobatch; run batch; NVT_NO_TAIL_BATCH=1 run nobatch
