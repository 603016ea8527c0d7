#!/bin/bash
out=/root/repo/gpurun_out/$1; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for col in C1 C11; do for hot in 0 1; do
  rm -rf /tmp/hp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp -o s -- python /root/repo/tools/hot_probe.py $col $hot > /dev/null 2>&1
  f=$(find /tmp/hp -name "s_kernel_stats.csv" | head -1)
  echo "== $col hot=$hot" >> $out/summary.txt
  python - "$f" >> $out/summary.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    nm=r["Name"].split("(")[0].replace("void ","").replace("nvt::","")[:46]
    if any(x in nm for x in ("part_","hot_","scan_","lds_","range_")):
        print(f"{nm:46s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done; done
cat $out/summary.txt
