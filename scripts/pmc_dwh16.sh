# PMC passes over scripts/mb_dwh16.py (the bf16 pass's weight-gradient contraction alone): HBM-side bytes, L2 hits, MFMA busy, clock
ROOT=$PWD; OUT=$PWD/gpurun_out; cd /tmp; export TMPDIR=/tmp
: > $OUT/r05_pmc_dwh16.txt
for CTR in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rocprofv3 --pmc $CTR -d $OUT/pmc16 -o p -- python $ROOT/scripts/mb_dwh16.py > /dev/null 2>&1
  DBP=$(ls $OUT/pmc16/*.db | head -1)
  echo "## $CTR" >> $OUT/r05_pmc_dwh16.txt
  python $ROOT/scripts/rocpd_pmc.py $DBP | grep -A4 "tn_tr" | grep -v "^at::\|^__amd" >> $OUT/r05_pmc_dwh16.txt
  python -c "import shutil, sys; shutil.rmtree(sys.argv[1], ignore_errors=True)" $OUT/pmc16
done
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT/pmc16 -o p -- python $ROOT/scripts/mb_dwh16.py > /dev/null 2>&1
echo "## clock (kernel-trace + GRBM_GUI_ACTIVE)" >> $OUT/r05_pmc_dwh16.txt
DBP=$(ls $OUT/pmc16/*.db | head -1)
sqlite3 $DBP "select 1" > /dev/null 2>&1
python - "$DBP" >> $OUT/r05_pmc_dwh16.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
d = [r[0] for r in db.execute("select duration from kernels where name like '%tn_tr%' order by duration")]
c = [r[0] for r in db.execute("select counter_value from pmc_events where name like '%tn_tr%' and counter_name='GRBM_GUI_ACTIVE' order by counter_value")]
print("durations us:", [round(x * 1e-3, 1) for x in d])
print("GRBM cycles :", [int(x) for x in c])
if d and c:
    print("clock of the largest dispatch: %.3f GHz" % (c[-1] / d[-1]))
PY
python -c "import shutil, sys; shutil.rmtree(sys.argv[1], ignore_errors=True)" $OUT/pmc16
cat $OUT/r05_pmc_dwh16.txt
