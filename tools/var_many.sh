# var_many.sh [-t "<pytest files>"] <variant> ...: one box, one bench run per argument in the order
# given ("base" = the default library, anything else = libnvt_hip_<variant>.so; "name@ENV=V" adds an
# environment variable).  Prints ms / step, GPU busy, the families and the counting regions alone.
cd $GRAFT_REPO_ROOT
export NVT_READBACK_TIMEOUT=60
if [ "$1" = "-t" ]; then timeout 900 python -m pytest $2 -x -q -m gpu 2>&1 | tail -3; shift 2; fi
run() { env $2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>gpurun_out/var_err_$1.log | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);pk=d['roofline']['per_kernel_ms_per_step']
print('%-10s'%'$1', round(d['ms_per_step'],3),d['gpu_busy_ms_per_step'],{k:v['ms_per_step'] for k,v in d['roofline']['per_family'].items()},{k.replace('dense_count_',''):pk[k] for k in pk if 'dense_count' in k})"; }
for v in "$@"; do
  name=${v%%@*}; envs="NVT_X=1"; [ "$v" != "$name" ] && envs=${v#*@}
  lib=${name%%+*}
  if [ "$lib" = "base" ]; then run $name "$envs"; else run $name "NVT_HIP_LIB=$PWD/nvtabular_amd/libnvt_hip_$lib.so $envs"; fi
done
