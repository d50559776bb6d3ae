#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
./scripts/probes/mfma_bf16_peak > $OUT/r05_mfma_sustained.txt 2>&1
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "split or lstm or saturated or bf16" > $OUT/r05_ops_tests.txt 2>&1
python scripts/mb_recurrence.py 20 20000 512 fp32,split9 > $OUT/r05_mb_recurrence.txt 2>&1
VD_LIB_PATH=$PWD/visdial_amd/libvisdial_hip_s40.so python scripts/mb_recurrence.py 20 20000 512 split9 >> $OUT/r05_mb_recurrence.txt 2>&1
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/r05_bench_fp32_b$i.json 2>> $OUT/r05_bench_b.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --recurrence split9 > $OUT/r05_bench_split9_b$i.json 2>> $OUT/r05_bench_b.err
VD_LIB_PATH=$PWD/visdial_amd/libvisdial_hip_s40.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --recurrence split9 > $OUT/r05_bench_split9_s40_b$i.json 2>> $OUT/r05_bench_b.err
done
cat $OUT/r05_mfma_sustained.txt; tail -n 5 $OUT/r05_ops_tests.txt; cat $OUT/r05_mb_recurrence.txt | grep -v amdgpu
for f in $OUT/r05_bench_*_b?.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f'))
r=d['roofline']; print(d['ms_per_step'], d['ms_per_step_median'], {k:(v['ms_total_per_step'],v['avg_launch_ms']) for k,v in r.get('families',{}).items()})
"; done
