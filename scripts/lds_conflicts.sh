ROOT=$PWD; OUT=$PWD/gpurun_out; cd /tmp; export TMPDIR=/tmp
for WHAT in "bench" "cfg1"; do
  if [ $WHAT = bench ]; then CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-alt"; else CMD="python $ROOT/scripts/run_config.py 1 3 1"; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $OUT/ldsc -o p -- $CMD > /dev/null 2>&1
  DBP=$(ls $OUT/ldsc/*.db | head -1)
  python - "$DBP" > $OUT/r05_lds_conflicts_$WHAT.txt <<'PY'
import sqlite3, sys
from collections import defaultdict
sys.path.insert(0, '/root/repo/scripts')
db = sqlite3.connect(sys.argv[1])
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for name, cn, v in db.execute("select name, counter_name, counter_value from pmc_events"):
    agg[name][cn] += v
    if cn == 'SQ_LDS_IDX_ACTIVE': cnt[name] += 1
rows = []
for k, d in agg.items():
    act, conf = d.get('SQ_LDS_IDX_ACTIVE', 0), d.get('SQ_LDS_BANK_CONFLICT', 0)
    if act > 0: rows.append((conf, act, d.get('SQ_BUSY_CYCLES', 0), k))
rows.sort(reverse=True)
print("# LDS bank-conflict cycles / LDS active cycles per kernel (summed over dispatches), share of SQ busy cycles")
for conf, act, busy, k in rows[:25]:
    print("%5.1f %% of LDS cycles are conflicts | LDS active = %5.1f %% of SQ busy | %s" % (100 * conf / act, 100 * act / max(busy, 1), k[:150]))
PY
  python -c "import shutil, sys; shutil.rmtree(sys.argv[1], ignore_errors=True)" $OUT/ldsc
done
cat $OUT/r05_lds_conflicts_bench.txt; echo; cat $OUT/r05_lds_conflicts_cfg1.txt
