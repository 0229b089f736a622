# bench lines at the per-GPU batches of the strong-scaling sweep (no side legs): python bench.py ... for 256 / 64 / 32
cd /root/repo
mkdir -p gpurun_out/quick
for b in 256 64 32; do
  timeout -k 10 300 python bench.py --no_cpu --no_micro --no_roofline --no_side --global_batch $b --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($b, d['ms_per_step'], d['value'])" || exit 1
done
