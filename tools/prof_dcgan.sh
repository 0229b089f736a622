# rocprofv3 kernel statistics of BASELINE config 5 (DCGAN WGAN-GP, 64x64, batch 64) -> gpurun_out/prof_dcgan/
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_dcgan
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_dcgan -o p --output-format csv -- python3 /root/repo/bench.py --no_cpu --no_micro --no_side --no_pmc --no_roofline --recipe dcgan --img_size 64 --base_dim 16 --global_batch 64 --steps 10 > /root/repo/gpurun_out/prof_dcgan.log 2>&1
rm -f /root/repo/gpurun_out/prof_dcgan/*kernel_trace.csv
tail -1 /root/repo/gpurun_out/prof_dcgan.log | cut -c1-300
