#!/bin/bash
# alone timings of the option recurrence's three kernels: product library vs a variant library (VD_LIB_PATH)
#   gpurun -- 'bash scripts/ab_probe.sh <variant-name> [out-tag]'
V=$1; TAG=${2:-$1}
OUT=gpurun_out/r06_probe_$TAG.txt; : > $OUT
for rep in 1 2; do
for v in product $V; do
  if [ $v = product ]; then unset VD_LIB_PATH; else export VD_LIB_PATH=$PWD/visdial_amd/libvisdial_hip_$v.so; fi
  echo "== $v" >> $OUT
  python scripts/mb_recurrence.py 20 20000 512 split9 2>/dev/null >> $OUT
  python scripts/mb_dwh.py 2>/dev/null | grep split9 >> $OUT
done; done
cat $OUT
