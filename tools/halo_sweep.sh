set -e
cd /root/repo
for shape in "256 128 32 32 128 3" "256 256 32 32 128 3" "256 512 8 8 512 3" "256 512 16 16 256 3" "256 256 16 16 256 3" "256 128 64 64 64 3" "256 64 64 64 64 3"; do
  for m in 0 1; do
    echo -n "HALO=$m "; GAN_AMD_HALO=$m timeout -k 10 60 python tools/bench_op.py fwd $shape 50
  done
done
