#!/bin/bash
# GPU session 1 (round 2): bench line format, graph-replay bisect, gradient divergence.
O=gpurun_out/s1; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -c 2200 $O/bench.json
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
for cfg in "--sync explicit" "--sync item" "--sync none" "--droppath 0" "--loss mse" "--loss aten" "--opt 0" "--loss aten --droppath 0"; do
  run 300 python tools/graph_replay_bisect.py losses $cfg >> $O/graph_losses.log 2>&1
done
cat $O/graph_losses.log
run 400 python tools/graph_replay_bisect.py trace --out $O/trace_pkt1.json > $O/trace_pkt1.log 2>&1
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run 400 python tools/graph_replay_bisect.py trace --out $O/trace_pkt0.json > $O/trace_pkt0.log 2>&1
run 100 python tools/graph_replay_bisect.py diff $O/trace_pkt1.json $O/trace_pkt0.json > $O/trace_diff.log 2>&1
cat $O/trace_diff.log
unset DEBUG_CLR_GRAPH_PACKET_CAPTURE
run 400 python tools/grad_divergence.py --seed 3 --dump $O > $O/graddiv_seed3.log 2>&1
run 400 python tools/grad_divergence.py --seed 0 --dump $O > $O/graddiv_seed0.log 2>&1
run 400 python tools/grad_divergence.py --seed 3 --train --dump $O > $O/graddiv_seed3_train.log 2>&1
tail -5 $O/graddiv_seed3.log $O/graddiv_seed0.log $O/graddiv_seed3_train.log
