# stand-alone duration of every kernel of one configuration's step: a counter run serialises the dispatches, so the kernel trace of
# `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` holds each kernel's time with the chip to itself.   bash scripts/alone_config.sh <config> <tag>
CFG=${1:-4}; TAG=${2:-r05_config4}
ROOT=$PWD; OUT=$PWD/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT/${TAG}_alone -o run -- python $ROOT/scripts/run_config.py $CFG 3 2 > $OUT/${TAG}_alone.log 2>&1
DB=$(ls $OUT/${TAG}_alone/*.db | head -1)
python $ROOT/scripts/rocpd_stats.py $DB > $OUT/${TAG}_alone_kernel_stats.txt
python -c "import shutil, sys; shutil.rmtree(sys.argv[1], ignore_errors=True)" $OUT/${TAG}_alone
tail -1 $OUT/${TAG}_alone.log
