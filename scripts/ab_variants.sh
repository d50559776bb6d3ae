#!/bin/bash
OUT=gpurun_out/r06_ab_cg.txt; : > $OUT
B="--steps 30 --warmup 6 --no-alt --no-other-configs --no-cpu-baseline"
one() { python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); f=d['roofline']['families']
print('%-10s step %.3f ms  fwd %.3f bwd %.3f dWh %.3f' % (sys.argv[1], d['ms_per_step'], f['opt_lstm_fwd']['ms_total_per_step'], f['opt_lstm_bwd']['ms_total_per_step'], f['opt_lstm_dWh']['ms_total_per_step']))" $1; }
for rep in 1 2; do
for v in cg8 cg4 default cg1; do
  if [ $v = default ]; then unset VD_LIB_PATH; else export VD_LIB_PATH=$PWD/visdial_amd/libvisdial_hip_$v.so; fi
  echo "== $v alone" >> $OUT; python scripts/mb_recurrence.py 20 20000 512 split9 2>/dev/null >> $OUT
  one $v >> $OUT
done; done
cat $OUT
