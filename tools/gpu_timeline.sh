#!/bin/bash
# per-dispatch timeline of the replayed step: gpurun -- 'bash tools/gpu_timeline.sh TAG'
TAG=${1:-r03}; O=gpurun_out/tl_$TAG; mkdir -p $O
R=/tmp/kmu_tl; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R -o tl -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
tail -c 600 $O/bench.json
T=$(find $R -name "*kernel_trace.csv" | head -1); echo "trace: $T"; head -2 $T | cut -c1-600
python3 tools/timeline.py $T $O/timeline.json $O/sequence.txt > $O/timeline.txt 2>&1; cat $O/timeline.txt
