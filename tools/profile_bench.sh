#!/bin/bash
# Profiles bench.py on the GPU box: rocprofv3 kernel trace + stats of the default command, then
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*), each with --kernel-trace only, plus a
# FETCH_SIZE calibration on a kernel with a known byte count and the same access pattern
# (tools/ubench/random_record_read). Usage: tools/profile_bench.sh <out-dir> [bench args...]
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-entry --no-cpu --no-recall --no-latency --no-e2e $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH --steps 20 --warmup 3 > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$tag" -o pmc -- $BENCH --no-accounting --steps 4 --warmup 1 > "$OUT/pmc_$tag.json" 2> "$OUT/pmc_$tag.err"
done
# calibration: 8M random 512-byte line-aligned records = 4 294 967 296 bytes per launch, 6 launches per run
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_fetch" -o pmc -- "$REPO/tools/ubench/random_record_read" 8192 512 8 > "$OUT/calib.txt" 2>&1
python "$REPO/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"
# the raw traces are large: keep only the CSV summaries
find "$OUT" -name "*.db" -delete; find "$OUT" -type f -size +400k -delete   # (gpurun copies at most 64 MiB back: summaries only)
