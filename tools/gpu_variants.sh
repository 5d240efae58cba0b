#!/bin/bash
# usage: gpu_variants.sh PATTERN VARIANT...   -- rocprofv3 kernel stats of tools/run_k2_shapes.py per library variant ("base" = shipped)
O=gpurun_out/variants; mkdir -p $O; export TMPDIR=/tmp
PAT=$1; shift
for v in "$@"; do
  R=/tmp/kmu_var_$v; rm -rf $R
  if [ "$v" = base ]; then unset KMU_LIB_VARIANT; else export KMU_LIB_VARIANT=$v; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R -o v -- python3 ${RUNNER:-tools/run_k2_shapes.py} 10 > $O/$v.log 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT $v"; exit 1; fi
  ST=$(find $R -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep "done" $O/$v.log
  python3 - "$ST" "$PAT" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("  %-70s %4d x %8.1f us" % (r["Name"][:70], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
done
