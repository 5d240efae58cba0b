#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate, no trace options) on tools/run_dominant_kernel.py -> HBM bytes per launch of the glue kernels
O=gpurun_out/pmc_glue; mkdir -p $O; export TMPDIR=/tmp
R=/tmp/kmu_pmcg; rm -rf $R
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/f -o f -- python3 tools/run_dominant_kernel.py 3 > $O/f.log 2>&1
run 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/w -o w -- python3 tools/run_dominant_kernel.py 3 > $O/w.log 2>&1
F=$(find $R/f -name "*counter_collection.csv" | head -1); W=$(find $R/w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $O/pmc_traffic.json f=$F w=$W > $O/pmc_traffic.txt 2>&1
grep -E "tn_|lca_|mean_rows|pw_gemm|bn_apply|dwconv3x3_kernel" $O/pmc_traffic.txt | cut -c1-160
