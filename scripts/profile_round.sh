#!/bin/bash
# Profiles of one round, run on the GPU box from the repo root:  bash scripts/profile_round.sh r01
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ and the summaries under gpurun_out/profiles/ (only
# gpurun_out/ travels back from the GPU box): copy those into profiles/ and commit them.
# Counter passes are separate runs (kernel-trace/stats never combined with --pmc; TCC FETCH_SIZE and WRITE_SIZE
# do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -o pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
cd /tmp && export TMPDIR=/tmp

# 1. kernel trace + stats of the HEADLINE workload alone (the default command's other workloads -- host-fed frames,
#    schedule runs with other Gaussian counts -- would mix their launches into the per-kernel averages): the JSON line of
#    this very run is kept next to the stats, so that roofline.avg_launch_us can be set against the CSV's AverageNs
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o run --output-format csv -- python3 "$ROOT/bench.py" \
    --no-host-frames --no-schedule --no-stable-targets --no-cpu-baseline > "$OUT/trace.log" 2>&1 || exit 1
grep '^{"metric"' "$OUT/trace.log" > "$SUM/${TAG}_bench_c3_under_rocprof.json"
cp "$OUT/trace/run_kernel_stats.csv" "$SUM/${TAG}_bench_c3_kernel_stats.csv"
python3 "$ROOT/scripts/category_summary.py" "$OUT/trace" > "$SUM/${TAG}_bench_c3_category_summary.txt" || exit 1
python3 "$ROOT/scripts/median_timeline.py" "$OUT/trace" > "$SUM/${TAG}_timeline_median_step.txt" || exit 1
echo "[profile] trace done"

# 2. HBM traffic: two TCC passes
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d "$OUT/pmc_$C" -o run --output-format csv -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-host-frames --no-schedule \
      --no-cpu-baseline --no-stable-targets > "$OUT/pmc_$C.log" 2>&1 || exit 1
done
python3 "$ROOT/scripts/pmc_summary.py" "$OUT/pmc_FETCH_SIZE/run_counter_collection.csv" \
    "$OUT/pmc_WRITE_SIZE/run_counter_collection.csv" "$SUM/${TAG}_pmc_hbm_traffic_c3" || exit 1
echo "[profile] hbm passes done"

# 3. what the resident waves do: SQ pass (8 slots)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -d "$OUT/pmc_sq" -o run --output-format csv -- python3 "$ROOT/bench.py" \
    --steps 5 --warmup 2 --no-cpu-baseline --no-stable-targets --no-host-frames --no-schedule > "$OUT/pmc_sq.log" 2>&1 || exit 1
python3 "$ROOT/scripts/pmc_sq_summary.py" "$OUT/pmc_sq/run_counter_collection.csv" \
    "$SUM/${TAG}_pmc_sq_c3.csv" || exit 1
echo "[profile] sq pass done"

# 4. LDS / scalar / MFMA instruction mix of the blend kernels (own pass; a counter this build does not know only loses
#    this summary)
if rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM \
    SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES -d "$OUT/pmc_lds" -o run --output-format csv -- python3 "$ROOT/bench.py" \
    --steps 5 --warmup 2 --no-cpu-baseline --no-stable-targets --no-host-frames --no-schedule > "$OUT/pmc_lds.log" 2>&1; then
  python3 "$ROOT/scripts/pmc_sq_summary.py" "$OUT/pmc_lds/run_counter_collection.csv" "$SUM/${TAG}_pmc_lds_c3.csv"
  echo "[profile] lds pass done"
else
  tail -5 "$OUT/pmc_lds.log"; echo "[profile] lds pass skipped"
fi
