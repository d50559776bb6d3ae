# per-stream timeline + kernel stats of one BASELINE configuration through the native host:  bash scripts/trace_config.sh <config> <tag>
CFG=${1:-4}; TAG=${2:-r05_config4}
ROOT=$PWD; OUT=$PWD/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/${TAG}_trace -o run -- python $ROOT/scripts/run_config.py $CFG 6 3 > $OUT/${TAG}_trace.log 2>&1
DB=$(ls $OUT/${TAG}_trace/*.db | head -1)
python $ROOT/scripts/rocpd_timeline.py $DB > $OUT/${TAG}_timeline.txt
python $ROOT/scripts/rocpd_stats.py $DB > $OUT/${TAG}_kernel_stats.txt
python -c "import shutil, sys; shutil.rmtree(sys.argv[1], ignore_errors=True)" $OUT/${TAG}_trace
tail -2 $OUT/${TAG}_trace.log
