#!/bin/bash
O=gpurun_out/timeline; mkdir -p $O; export TMPDIR=/tmp
R=/tmp/kmu_tl; rm -rf $R
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R -o tl -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
T=$(find $R -name "*kernel_trace.csv" | head -1); echo "trace: $T $(wc -l < $T) rows"
head -1 $T > $O/trace_header.txt
python3 tools/timeline_summary.py $T 5 60 > $O/timeline.txt 2>&1; head -90 $O/timeline.txt | cut -c1-180
