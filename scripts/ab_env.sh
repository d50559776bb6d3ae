#!/bin/bash
# same-box A/B of an environment switch of the library: bash scripts/ab_env.sh VAR a b [tag]   (e.g. VD_SPLIT_PLANES 0 1)
VAR=$1; A=$2; B_=$3; TAG=${4:-$1}
OUT=gpurun_out/r06_ab_$TAG.txt; : > $OUT
FL="--steps 30 --warmup 6 --no-alt --no-other-configs --no-cpu-baseline"
one() { python bench.py $FL 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); f=d['roofline']['families']
print('%-22s step %.3f ms  fwd %.3f (%.1f us)  bwd %.3f (%.1f us)  dWh %.3f' % (sys.argv[1], d['ms_per_step'], f['opt_lstm_fwd']['ms_total_per_step'], 1e3*f['opt_lstm_fwd']['avg_launch_ms'], f['opt_lstm_bwd']['ms_total_per_step'], 1e3*f['opt_lstm_bwd']['avg_launch_ms'], f['opt_lstm_dWh']['ms_total_per_step']))" "$1"; }
for rep in 1 2 3; do
for v in $A $B_; do
  export $VAR=$v
  one "$VAR=$v" >> $OUT
done; done
cat $OUT
