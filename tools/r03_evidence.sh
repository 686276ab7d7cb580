#!/bin/bash
# Round-3 GPU visit in stages (each GPU step under its own time limit, chained: the first failure
# ends the script).   gpurun --timeout 1200 -- 'bash tools/r03_evidence.sh tests|gated|bench|pmc'
set -eo pipefail
STAGE=${1:-tests}
OUT=gpurun_out/r03
mkdir -p "$OUT"
PYT="python -u -m pytest -x -q --timeout 300 --timeout-method thread"
case "$STAGE" in
tests)       # the whole GPU suite on the default (shipped) paths
  timeout -k 10 900 $PYT tests -m gpu > "$OUT/pytest_gpu_full.log" 2>&1 || { tail -30 "$OUT/pytest_gpu_full.log"; exit 1; }
  tail -2 "$OUT/pytest_gpu_full.log" ;;
gated)       # the routes that are off by default, on the tests that reach them
  HYDRA_CHOL_FUSED_STEP=1 timeout -k 10 600 $PYT tests/test_gpu_parity.py tests/test_dist.py tests/test_gpu_fullsize_c345.py \
      -m gpu -k "chol or dense or cache or rowshard or row_sharded or solve or c5" > "$OUT/pytest_fused_step.log" 2>&1 \
      || { tail -30 "$OUT/pytest_fused_step.log"; exit 1; }
  tail -2 "$OUT/pytest_fused_step.log"
  HYDRA_SH_OWN_GEMM=1 timeout -k 10 600 $PYT tests/test_gpu_fullsize_c345.py tests/test_gpu_flags.py tests/test_gpu_heads2.py \
      -m gpu -k "streamed or sh_ or lhs" > "$OUT/pytest_sh_own_gemm.log" 2>&1 \
      || { tail -30 "$OUT/pytest_sh_own_gemm.log"; exit 1; }
  tail -2 "$OUT/pytest_sh_own_gemm.log"
  for n in 2000 4096; do
    timeout -k 10 120 python -u tools/chol_bench.py $n 2>&1 | tail -2
    HYDRA_CHOL_FUSED_STEP=1 timeout -k 10 120 python -u tools/chol_bench.py $n 2>&1 | tail -2
  done
  HYDRA_CHOL_FUSED_STEP=1 timeout -k 10 120 python -u tools/chol_bench.py 40000 2>&1 | tail -2 ;;
bench)
  timeout -k 10 300 python -u bench.py > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err" || { tail -20 "$OUT/bench_c2.err"; exit 1; }
  cat "$OUT/bench_c2.json"
  timeout -k 10 500 python -u bench.py --config c5 --steps 2 --warmup 1 --no-cpu > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err" \
      || { tail -20 "$OUT/bench_c5.err"; exit 1; }
  cat "$OUT/bench_c5.json" ;;
pmc)
  bash tools/pmc_cmd.sh r03 chol40k tools/chol_bench.py 40000
  bash tools/pmc_cmd.sh r03 antspace350 tools/antspace_big_bench.py
  SKIP_TESTS=1 bash tools/gpu_check.sh r03 c4
  SKIP_TESTS=1 bash tools/gpu_check.sh r03 c3t40 ;;
esac
