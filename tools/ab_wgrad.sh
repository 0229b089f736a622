#!/bin/bash
# A/B of two builds of the library on the tiled weight gradient's shapes: bash tools/ab_wgrad.sh <old.so>
for shp in "256 128 32 32 128 3" "256 256 16 16 256 3" "256 512 8 8 512 3" "256 512 4 4 512 3" "256 64 64 64 128 3" "256 128 32 32 256 3" "256 128 16 16 256 1" "32 128 32 32 128 3" "32 512 8 8 512 3"; do
  for lib in $1 gan_amd/libgan_amd.so; do
    echo -n "$(basename $lib)  "; GAN_AMD_LIB=$PWD/$lib timeout -k 10 60 python tools/bench_op.py wgrad $shp 30 2>/dev/null || exit 1
  done
done
