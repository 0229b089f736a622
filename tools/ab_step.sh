#!/bin/bash
# Same-box A/B of the train step: bash tools/ab_step.sh "<env A>" "<env B>" [global_batch]   (ms/step of 20 replayed iterations each, twice)
B=${3:-256}
for rep in 1 2; do
  for e in "$1" "$2"; do
    echo -n "[$e] "; env $e timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no_cpu --no_micro --no_roofline --no_side --no_pmc --global_batch $B 2>/dev/null | python -c "import sys,json; p=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(p['ms_per_step'], p.get('per_gpu_share_ms'))" || exit 1
  done
done
