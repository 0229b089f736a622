# halo kernel on / off over the wide layers of the 256x256 networks (profiling aid)
cd /root/repo
while read -r op shape; do
  for m in 0 1; do
    echo -n "HALO=$m "; GAN_AMD_HALO=$m timeout -k 10 60 python tools/bench_op.py $op $shape 50 2>&1 | grep -v amdgpu.ids
  done
done <<'LIST'
fwd_up2 256 128 32 32 64 3
fwd_up2 256 256 16 16 128 3
fwd_up2 256 512 8 8 256 3
fwd_up2 256 512 4 4 512 3
fwd_pool 256 64 64 64 128 3
fwd_pool 256 128 32 32 256 3
fwd_pool 256 256 16 16 512 3
fwd 256 128 64 64 64 3
fwd 256 64 64 64 64 3
LIST
