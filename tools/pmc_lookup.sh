#!/bin/bash
# L2 hit / miss requests and fabric fetch bytes of the cfg4 lookup kernel, key directory vs flat table
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_lookup; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp NVT_READBACK_TIMEOUT=60
P="python $GRAFT_REPO_ROOT/tools/cfg4_probe.py"
for mode in 1 0; do
  for pmc in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    NVT_KEYED_IMAGES=$mode timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out -o k${mode}_$tag -- $P > $out/k${mode}_$tag.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
out=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc_lookup"
for f in sorted(glob.glob(out+"/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "lookup" not in k and "image_build" not in k and "keydir" not in k: continue
        a=acc[(k[:60], r["Counter_Name"])]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    print(os.path.basename(f))
    for (k,c),(n,v) in sorted(acc.items()): print("   %-62s %-24s launches %3d  per launch %.4g" % (k,c,n,v/n))
PY
find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
