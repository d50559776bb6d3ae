#!/bin/bash
# Round measurement recipe (run on the GPU box from the repo root: `gpurun -- 'bash scripts/measure_round.sh r03'`).
# Writes everything under gpurun_out/<tag>_*; the builder copies the summaries into profiles/.
#   1. driver-style bench line (default flags) + a 50-step line            -> <tag>_bench_default.json, <tag>_bench_50steps.json
#   2. rocprofv3 --kernel-trace of bench.py -> per-kernel stats + per-stream timeline (scripts/rocpd_stats.py, rocpd_timeline.py)
#   3. three separate rocprofv3 --pmc passes over scripts/pmc_target.py (FETCH_SIZE | WRITE_SIZE | MFMA busy + clocks)
#   4. non-GEMM kernels: alone GB/s + in-step averages (scripts/hbm_kernels.py)
#   5. BASELINE.json configs[4] bench line
TAG=${1:-r03}
OUT=$PWD/gpurun_out
ROOT=$PWD
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_bench_50steps.json 2>> $OUT/${TAG}_bench_default.err
python bench.py --config 4 --steps 30 --warmup 8 --no-cpu-baseline > $OUT/${TAG}_bench_config4.json 2>> $OUT/${TAG}_bench_default.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/${TAG}_trace -o run -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
DB=$(ls $OUT/${TAG}_trace/*.db | head -1)
python $ROOT/scripts/rocpd_stats.py $DB --steps-only > $OUT/${TAG}_kernel_stats_bench.txt
python $ROOT/scripts/rocpd_timeline.py $DB > $OUT/${TAG}_stream_timeline_bench.txt
rm -rf $OUT/${TAG}_trace
: > $OUT/${TAG}_pmc_option_lstm_kernels.txt
for CTR in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  rocprofv3 --pmc $CTR -d $OUT/${TAG}_pmc -o p -- python $ROOT/scripts/pmc_target.py > /dev/null 2>> $OUT/${TAG}_trace.err
  DBP=$(ls $OUT/${TAG}_pmc/*.db | head -1)
  echo "## $CTR" >> $OUT/${TAG}_pmc_option_lstm_kernels.txt
  python $ROOT/scripts/rocpd_pmc.py $DBP | grep -v "^# pmc_events" >> $OUT/${TAG}_pmc_option_lstm_kernels.txt
  rm -rf $OUT/${TAG}_pmc
done
cd $ROOT
python scripts/hbm_kernels.py $OUT/${TAG}_kernel_stats_bench.txt > $OUT/${TAG}_hbm_kernels.txt 2>> $OUT/${TAG}_trace.err
tail -c 1500 $OUT/${TAG}_bench_default.json
