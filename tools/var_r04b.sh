# one box: exactness of the partition-kernel carry and of the 24-bit-multiply hashes (range path +
# parity tests on two libraries), then the count family of every variant (bench, third pass)
cd $GRAFT_REPO_ROOT
export NVT_READBACK_TIMEOUT=60
T="tests/test_gpu_range_path.py tests/test_gpu_parity.py tests/test_gpu_edges.py"
run() { env $2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>gpurun_out/var_err_$1.log | tee gpurun_out/var_raw_$1.json | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);pk=d['roofline']['per_kernel_ms_per_step']
print('$1', round(d['ms_per_step'],3),d['gpu_busy_ms_per_step'],{k:v['ms_per_step'] for k,v in d['roofline']['per_family'].items()},{k:pk[k] for k in pk if 'dense_count' in k}, (d.get('parity') or {}).get('full_frame_ok'))"; }
L=$PWD/nvtabular_amd/libnvt_hip
run nocarry NVT_HIP_LIB=${L}_nocarry.so
run carry NVT_X=1
run hot24 NVT_HIP_LIB=${L}_hot24.so
run all24 NVT_HIP_LIB=${L}_all24.so
run carry_u4 NVT_RANGE_U4_BITS=9
run nocarry NVT_HIP_LIB=${L}_nocarry.so
run all24 NVT_HIP_LIB=${L}_all24.so
