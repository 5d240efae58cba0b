#!/bin/bash
# GPU session 2 (round 2): finer graph-replay bisect + torch-only reproducer, determinism check, smoke, full gpu test suite.
O=gpurun_out/s2; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
for cfg in "--loss aten" "--loss aten_conv" "--loss aten_nominmax" "--loss aten_nossim"; do
  run 300 python tools/graph_replay_bisect.py losses $cfg >> $O/graph_losses.log 2>&1
done
for cfg in "--blocks 24 --droppath 1" "--blocks 24 --droppath 0" "--blocks 80 --droppath 1"; do
  run 300 python tools/repro_graph_packet_capture_torch_only.py $cfg >> $O/torch_only.log 2>&1
done
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run 300 python tools/repro_graph_packet_capture_torch_only.py --blocks 24 --droppath 1 >> $O/torch_only.log 2>&1
grep -v amdgpu.ids $O/graph_losses.log $O/torch_only.log
unset DEBUG_CLR_GRAPH_PACKET_CAPTURE
run 400 python tools/graph_replay_bisect.py trace --opt 0 --droppath 0 --out $O/trace_fixed.json > $O/trace_fixed.log 2>&1
run 100 python tools/graph_replay_bisect.py selfdiff $O/trace_fixed.json > $O/selfdiff.log 2>&1
cat $O/selfdiff.log
run 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1
grep -v amdgpu.ids $O/smoke.log | tail -8
run 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log
