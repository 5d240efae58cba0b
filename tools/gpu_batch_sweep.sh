#!/bin/bash
# Batch sweep of the headline step on one box: is the step latency-bound at B=8?
#   gpurun -- 'bash tools/gpu_batch_sweep.sh [tag]'   ->  gpurun_out/sweep_<tag>/sweep.json (copy to profiles/)
TAG=${1:-r03}
O=gpurun_out/sweep_$TAG; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
for B in 2 8 16 32; do
  run 400 python3 bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --kernels-out $O/kernels_b$B.json > $O/b$B.json 2> $O/b$B.err
  tail -c 300 $O/b$B.err | tail -2
done
python3 - "$O" <<'EOF'
import json, sys, os
o = sys.argv[1]
rows = []
for b in (2, 8, 16, 32):
    try:
        d = json.loads(open(os.path.join(o, "b%d.json" % b)).read().strip().splitlines()[-1])
        rows.append({"batch": b, "ms_per_step": d["ms_per_step"], "frames_per_s": d["value"],
                     "us_per_frame": 1e3 * d["ms_per_step"] / (b * 10), "hip_kernels_ms_per_step": d.get("hip_kernels_ms_per_step")})
    except Exception as e:
        rows.append({"batch": b, "error": str(e)})
json.dump({"workload": "bench.py --batch B --steps 20 --warmup 5 --no-cpu-baseline (KM_UNetV3_SH train step, T=10, 128x128, hipGraph replay), one box",
           "rows": rows}, open(os.path.join(o, "sweep.json"), "w"), indent=1)
print(json.dumps(rows))
EOF
