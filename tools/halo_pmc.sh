# PMC pass over one conv layer with the halo kernel on and off (profiling aid)
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  export GAN_AMD_HALO=$m
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY --kernel-trace -d /root/repo/gpurun_out/pmc_halo$m -o p --output-format csv -- python3 /root/repo/tools/bench_op.py fwd 256 128 32 32 128 3 5 > /root/repo/gpurun_out/pmc_halo$m.log 2>&1 || exit 1
done
