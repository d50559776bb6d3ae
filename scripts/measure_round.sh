#!/bin/bash
# Round measurement recipe (run on the GPU box from the repo root: `gpurun -- 'bash scripts/measure_round.sh r05'`).
# Writes everything under gpurun_out/<tag>_*; `python scripts/install_profiles.py <tag>` copies the summaries into profiles/.
#   1. driver-style bench line (default flags: split9 headline + `alt` fp32-MFMA line + other configs + CPU baseline), a 50-step line,
#      the configs[4] line
#   2. rocprofv3 --kernel-trace --marker-trace of bench.py -> per-kernel stats + per-stream timeline (scripts/rocpd_stats.py, rocpd_timeline.py);
#      VD_ROCTX=1: the library's ROCTx ranges (step phases) are in the same database (scripts/rocpd_ranges.py)
#   3. three separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy + clocks) over scripts/pmc_target.py, once per
#      arithmetic of the option recurrence (fp32 MFMA, split9)
#   4. non-GEMM kernels: alone GB/s + in-step averages (scripts/hbm_kernels.py)
TAG=${1:-r06}
OUT=$PWD/gpurun_out
ROOT=$PWD
mkdir -p "$OUT"
python bench.py > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err"
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > "$OUT/${TAG}_bench_50steps.json" 2>> "$OUT/${TAG}_bench_default.err"
python bench.py --config 4 --steps 30 --warmup 8 --no-cpu-baseline > "$OUT/${TAG}_bench_config4.json" 2>> "$OUT/${TAG}_bench_default.err"
cd /tmp; export TMPDIR=/tmp
VD_ROCTX=1 rocprofv3 --kernel-trace --marker-trace -d "$OUT/${TAG}_trace" -o run -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-alt > "$OUT/${TAG}_trace_bench.json" 2> "$OUT/${TAG}_trace.err"
DB=$(ls "$OUT/${TAG}_trace"/*.db | head -1)
python "$ROOT/scripts/rocpd_stats.py" "$DB" --steps-only > "$OUT/${TAG}_kernel_stats_bench.txt"
python "$ROOT/scripts/rocpd_timeline.py" "$DB" > "$OUT/${TAG}_stream_timeline_bench.txt"
python "$ROOT/scripts/rocpd_ranges.py" "$DB" > "$OUT/${TAG}_roctx_ranges.txt"
python -c "import shutil, sys; shutil.rmtree(sys.argv[1], ignore_errors=True)" "$OUT/${TAG}_trace"
: > "$OUT/${TAG}_pmc_option_lstm_kernels.txt"
for MODE in fp32 split9; do
for CTR in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  rocprofv3 --pmc $CTR -d "$OUT/${TAG}_pmc" -o p -- python "$ROOT/scripts/pmc_target.py" $MODE > /dev/null 2>> "$OUT/${TAG}_trace.err"
  DBP=$(ls "$OUT/${TAG}_pmc"/*.db | head -1)
  echo "## $MODE $CTR" >> "$OUT/${TAG}_pmc_option_lstm_kernels.txt"
  python "$ROOT/scripts/rocpd_pmc.py" "$DBP" | grep -v "^# pmc_events" | grep -A5 -E "^gemm_f32_glds|^gemm_split" | grep -v "^--" >> "$OUT/${TAG}_pmc_option_lstm_kernels.txt"
  python -c "import shutil, sys; shutil.rmtree(sys.argv[1], ignore_errors=True)" "$OUT/${TAG}_pmc"
done
done
cd "$ROOT"
python scripts/hbm_kernels.py "$OUT/${TAG}_kernel_stats_bench.txt" > "$OUT/${TAG}_hbm_kernels.txt" 2>> "$OUT/${TAG}_trace.err"
tail -c 1500 "$OUT/${TAG}_bench_default.json"
