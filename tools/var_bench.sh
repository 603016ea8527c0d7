cp nvtabular_amd/libnvt_hip.so /tmp/orig.so
for v in orig hot2 hot3; do
  if [ $v != orig ]; then cp nvtabular_amd/libnvt_v_$v.so nvtabular_amd/libnvt_hip.so; fi
  echo "== $v"; python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['per_kernel_ms_per_step']['encode_i32'])"
done
cp /tmp/orig.so nvtabular_amd/libnvt_hip.so
