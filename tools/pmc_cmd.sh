#!/bin/bash
# Kernel summary + the three PMC passes of one python command (a tool script rather than bench.py):
#   bash tools/pmc_cmd.sh r03 chol40k tools/chol_bench.py 40000
# writes gpurun_out/<tag>/<name>_kernel_stats.csv and <name>_pmc_{FETCH_SIZE,WRITE_SIZE,SQ_WAVE_CYCLES}.csv.
# Each pass is a run of its own with its own time limit; the first failure ends the script.
set -eo pipefail
TAG=$1; NAME=$2; shift 2
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SCRIPT=$ROOT/$1; shift
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$NAME" -o "$NAME" -- \
    python3 "$SCRIPT" "$@" > "$OUT/prof_$NAME.log" 2>&1 || { tail -20 "$OUT/prof_$NAME.log"; exit 1; }
python3 "$ROOT/tools/rocpd_summary.py" stats "$OUT/prof_$NAME/${NAME}_results.db" \
    > "$OUT/${NAME}_kernel_stats.csv" && rm -rf "$OUT/prof_$NAME"
echo "$NAME: kernel stats done"
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  P=$(echo "$PASS" | cut -d' ' -f1)
  timeout -s KILL 300 rocprofv3 --pmc $PASS -d "$OUT/pmc_${NAME}_$P" -o "$NAME" -- \
      python3 "$SCRIPT" "$@" > "$OUT/pmc_${NAME}_$P.log" 2>&1 || { tail -20 "$OUT/pmc_${NAME}_$P.log"; exit 1; }
  python3 "$ROOT/tools/rocpd_summary.py" pmc "$OUT/pmc_${NAME}_$P/${NAME}_results.db" \
      > "$OUT/${NAME}_pmc_$P.csv" && rm -rf "$OUT/pmc_${NAME}_$P"
  echo "$NAME: $P done"
done
