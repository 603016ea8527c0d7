# merge kernel variants (workgroup size x values per thread) on bench-sized lists
for v in ${@:-default mg_1024_5 mg_1024_3 mg_1024_7 mg_512_5 mg_512_3 mg_768_5 default}; do
  if [ $v = default ]; then python tools/merge_probe.py 6000000 13; else NVT_HIP_LIB=$PWD/nvtabular_amd/libnvt_hip_$v.so python tools/merge_probe.py 6000000 13; fi
done 2>&1 | grep "^lib"
