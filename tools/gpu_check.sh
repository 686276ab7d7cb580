#!/bin/bash
# One GPU-box visit: parity tests, the bench line, the rocprofv3 kernel summary and the PMC
# traffic passes of the same bench command.  Every GPU step has its own time limit and the
# steps are chained: the first failure ends the script.
#   gpurun --timeout 1100 -- '[SKIP_TESTS=1] bash tools/gpu_check.sh r1 c2 [extra bench flags]'
set -eo pipefail
TAG=${1:-r1}
CFG=${2:-c2}
EXTRA=${3:-}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread \
    > "$OUT/pytest_gpu.log" 2>&1 || { tail -30 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
fi
timeout -k 10 600 python -u bench.py --config "$CFG" $EXTRA > "$OUT/bench_$CFG.json" 2> "$OUT/bench_$CFG.err" \
    || { tail -30 "$OUT/bench_$CFG.err"; exit 1; }
cat "$OUT/bench_$CFG.json"
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/prof_$CFG" -o "$CFG" -- \
    python3 "$ROOT/bench.py" --config "$CFG" $EXTRA --no-cpu --steps 2 --warmup 1 \
    > "$ROOT/$OUT/prof_$CFG.log" 2>&1 || { tail -30 "$ROOT/$OUT/prof_$CFG.log"; exit 1; }
# the rocpd databases are tens of MB each: keep the text summaries, drop the databases
python3 "$ROOT/tools/rocpd_summary.py" stats "$ROOT/$OUT/prof_$CFG/${CFG}_results.db" \
    > "$ROOT/$OUT/${CFG}_kernel_stats.csv" && rm -rf "$ROOT/$OUT/prof_$CFG"
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  NAME=$(echo "$PASS" | cut -d' ' -f1)
  timeout -s KILL 600 rocprofv3 --pmc $PASS -d "$ROOT/$OUT/pmc_${CFG}_$NAME" -o "$CFG" -- \
      python3 "$ROOT/bench.py" --config "$CFG" $EXTRA --no-cpu --steps 1 --warmup 0 \
      > "$ROOT/$OUT/pmc_${CFG}_$NAME.log" 2>&1 || { tail -30 "$ROOT/$OUT/pmc_${CFG}_$NAME.log"; exit 1; }
  python3 "$ROOT/tools/rocpd_summary.py" pmc "$ROOT/$OUT/pmc_${CFG}_$NAME/${CFG}_results.db" \
      > "$ROOT/$OUT/${CFG}_pmc_$NAME.csv" && rm -rf "$ROOT/$OUT/pmc_${CFG}_$NAME"
done
cd "$ROOT"
ls "$OUT"
