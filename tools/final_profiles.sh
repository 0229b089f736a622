# the round's judged artifacts from the tree as it is: full bench line, rocprof kernel stats (batch 256 and 32), the per-GPU batch table
set -e
cd /root/repo
mkdir -p gpurun_out/final
timeout -k 10 900 python bench.py > gpurun_out/final/bench_full.json 2> gpurun_out/final/bench_full.err
for b in 128 64 32; do
  timeout -k 10 300 python bench.py --no_cpu --no_micro --no_roofline --no_side --global_batch $b --steps 10 2>/dev/null | tail -1 > gpurun_out/final/bench_b$b.json
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/final/prof256 -o r03 --output-format csv -- python3 /root/repo/bench.py --no_cpu --no_micro --no_side --no_pmc --no_roofline > /root/repo/gpurun_out/final/prof256.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/final/prof32 -o r03b32 --output-format csv -- python3 /root/repo/bench.py --no_cpu --no_micro --no_side --no_pmc --no_roofline --global_batch 32 --steps 10 > /root/repo/gpurun_out/final/prof32.log 2>&1
rm -f /root/repo/gpurun_out/final/prof256/*kernel_trace.csv /root/repo/gpurun_out/final/prof32/*kernel_trace.csv
echo done
