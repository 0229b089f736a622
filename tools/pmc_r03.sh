#!/bin/bash
# Round-3 PMC evidence for the three families that own the step: one rocprofv3 --pmc pass per
# counter group (SQ <= 8, FETCH_SIZE alone, WRITE_SIZE alone), each its own short run of tools/bench_op.py.
# usage: bash tools/pmc_r03.sh <outdir>     (run from the repo root on the GPU box)
set -u
OUT=${1:-gpurun_out/r03pmc}
ROOT=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {  # name, counters, bench_op args...
  local name=$1 ctrs=$2; shift 2
  ( cd /tmp && timeout -s KILL 150 rocprofv3 --pmc $ctrs --kernel-trace -d "$ROOT/$OUT/$name" -o pmc --output-format csv -- python3 "$ROOT/tools/bench_op.py" "$@" 6 ) > "$ROOT/$OUT/$name.log" 2>&1 || { echo "FAILED $name"; tail -3 "$ROOT/$OUT/$name.log"; return 1; }
  echo "done $name"
}
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES"
# PMC_SPECS: space-separated subset of the tags below (default: all)
ALL=("wgrad:wgrad 256 128 32 32 128 3" "halo:fwd 256 128 32 32 128 3" "thin64:fwd 256 64 64 64 64 3" "thin16:fwd 256 16 256 256 16 3" "thin16dp:fwd_pool 256 16 256 256 32 3" "thin32dp:fwd_pool 256 32 128 128 64 3")
TAGS=""
for spec in "${ALL[@]}"; do
  tag=${spec%%:*}
  if [ -n "${PMC_SPECS:-}" ] && [[ " $PMC_SPECS " != *" $tag "* ]]; then continue; fi
  TAGS="$TAGS $tag"
done
for spec in "${ALL[@]}"; do
  if [[ " $TAGS " != *" ${spec%%:*} "* ]]; then continue; fi
  tag=${spec%%:*}; args=${spec#*:}
  run ${tag}_sq1 "$SQ1" $args && run ${tag}_sq2 "$SQ2" $args && run ${tag}_fetch "FETCH_SIZE" $args && run ${tag}_write "WRITE_SIZE" $args && run ${tag}_grbm "GRBM_GUI_ACTIVE" $args || exit 1
done
for tag in $TAGS; do
  echo "=== $tag"
  for p in sq1 sq2 fetch write grbm; do python3 tools/pmc_summary.py "$OUT/${tag}_$p" conv_; done
done > "$OUT/summary.txt"
# the raw csv files are large (one row per counter and dispatch): keep the summary only
rm -rf "$OUT"/*_sq1 "$OUT"/*_sq2 "$OUT"/*_fetch "$OUT"/*_write "$OUT"/*_grbm
echo "summary written"
