#!/bin/bash
# A/B of the round-2 (GAN_AMD_HALO4=0) and round-3 4-wave resident-halo kernels on the train step's shapes
for shp in "fwd 256 128 32 32 128 3" "fwd 256 256 16 16 256 3" "fwd_up2 256 128 32 32 64 3" "fwd_up2 256 256 16 16 128 3" "fwd_pool 256 64 64 64 128 3" "fwd_pool 256 128 32 32 256 3" "fwd_up2 256 512 8 8 256 3" "fwd 64 128 32 32 128 3"; do
  for v in 0 1; do
    echo -n "HALO4=$v  "; GAN_AMD_HALO4=$v timeout -k 10 60 python tools/bench_op.py $shp 30 || exit 1
  done
done
