#!/bin/bash
# same-box A/B of a variant library against the product on one bench configuration: bash scripts/ab_lib.sh <variant> <config 3|4> [tag]
V=$1; CFG=${2:-3}; TAG=${3:-$1}
OUT=gpurun_out/r06_ablib_$TAG.txt; : > $OUT
FL="--config $CFG --steps 30 --warmup 6 --no-alt --no-other-configs --no-cpu-baseline"
one() { python bench.py $FL 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); f=d['roofline']['families']
print('%-12s step %.3f ms  fwd %.3f (%.1f us)  bwd %.3f (%.1f us)  dWh %.3f' % (sys.argv[1], d['ms_per_step'], f['opt_lstm_fwd']['ms_total_per_step'], 1e3*f['opt_lstm_fwd']['avg_launch_ms'], f['opt_lstm_bwd']['ms_total_per_step'], 1e3*f['opt_lstm_bwd']['avg_launch_ms'], f['opt_lstm_dWh']['ms_total_per_step']))" "$1"; }
for rep in 1 2 3; do
for v in product $V; do
  if [ $v = product ]; then unset VD_LIB_PATH; else export VD_LIB_PATH=$PWD/visdial_amd/libvisdial_hip_$v.so; fi
  one $v >> $OUT
done; done
cat $OUT
