#!/bin/bash
# FIR microbenchmark, round-2 kernel (GAN_AMD_FIR_OLD=1) against the compile-time-geometry kernel: bash tools/ab_fir.sh
for shp in "32 16 1024 1024" "32 32 512 512" "32 64 256 256" "32 128 128 128" "32 256 64 64" "32 512 32 32" "32 512 16 16"; do
  for ud in "1 1" "2 1"; do
    [ "$ud" = "2 1" ] && [ "$shp" = "32 16 1024 1024" ] && continue
    [ "$ud" = "2 1" ] && [ "$shp" = "32 32 512 512" ] && continue
    for v in 1 ""; do
      echo -n "old=$v  "; GAN_AMD_FIR_OLD=$v timeout -k 10 60 python tools/bench_fir.py $shp $ud 20 2>/dev/null || exit 1
    done
  done
done
