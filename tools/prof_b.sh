# rocprofv3 kernel statistics of bench.py at one global batch:  bash tools/prof_b.sh 32  ->  gpurun_out/prof_b32/
B=${1:-32}
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_b$B
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_b$B -o p --output-format csv -- python3 /root/repo/bench.py --no_cpu --no_micro --no_side --no_pmc --no_roofline --global_batch $B --steps 10 > /root/repo/gpurun_out/prof_b$B.log 2>&1
rm -f /root/repo/gpurun_out/prof_b$B/*kernel_trace.csv
tail -1 /root/repo/gpurun_out/prof_b$B.log | cut -c1-200
