#!/bin/bash
O=gpurun_out/s11; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
for i in 1 2 3; do
KMU_FORCE_DIST=1 run 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 1 --steps 3 --warmup 1 --batch 2 --size 64 --no-cpu-baseline > $O/rccl$i.json 2> $O/rccl$i.err
echo "rccl run $i: $(grep -c collective $O/rccl$i.json) line(s)"; grep -m3 -E "terminate|what\(\)|Error" $O/rccl$i.err | cut -c1-300
done
run 1000 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.log | tail -8 | cut -c1-200
