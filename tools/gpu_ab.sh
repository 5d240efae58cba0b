#!/bin/bash
# A/B of runtime switches at the whole-step level on ONE box: gpu_ab.sh "NAME=ENV1=V,ENV2=V" ...   ("base=" = defaults)
O=gpurun_out/ab; mkdir -p $O
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  ( IFS=,; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
    timeout -k 10 300 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --kernels-out $O/k_$name.json > $O/$name.json 2> $O/$name.err )
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT $name"; exit 1; fi
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $O/$name.json)"
done
