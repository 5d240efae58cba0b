#!/bin/bash
# SQ counters + HBM traffic of the fused FFN kernels (tools/bench_ffn.py), separate --pmc passes
O=gpurun_out/pmc_ffn; mkdir -p $O; export TMPDIR=/tmp
R=/tmp/kmu_pmcffn; rm -rf $R
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/sq -o sq -- python3 tools/bench_ffn.py 3 > $O/sq.log 2>&1
run 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d $R/sq2 -o sq2 -- python3 tools/bench_ffn.py 3 > $O/sq2.log 2>&1
run 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/f -o f -- python3 tools/bench_ffn.py 3 > $O/f.log 2>&1
run 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/w -o w -- python3 tools/bench_ffn.py 3 > $O/w.log 2>&1
S=$(find $R/sq -name "*counter_collection.csv" | head -1); S2=$(find $R/sq2 -name "*counter_collection.csv" | head -1)
F=$(find $R/f -name "*counter_collection.csv" | head -1); W=$(find $R/w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $O/pmc_sq.json sq=$S > $O/pmc_sq.txt 2>&1
python3 tools/pmc_summary.py $O/pmc_sq2.json sq=$S2 > $O/pmc_sq2.txt 2>&1
python3 tools/pmc_summary.py $O/pmc_traffic.json f=$F w=$W --fetch-x2 > $O/pmc_traffic.txt 2>&1
cut -c1-330 $O/pmc_sq.txt; python3 - $O/pmc_sq2.json <<'PY'
import json, sys
for k, d in json.load(open(sys.argv[1]))["kernels"].items():
    print(k, {c: int(v) for c, v in d.items()})
PY
cut -c1-200 $O/pmc_traffic.txt; tail -3 $O/sq2.log
