#!/bin/bash
# round 5, GPU call 1: split9 kernel variants + bf16 MFMA ceiling + this box's baseline step
OUT=gpurun_out; mkdir -p $OUT
./scripts/probes/mfma_bf16_peak > $OUT/r05_mfma_bf16_peak.txt 2>&1
export VD_LIB_PATH=$PWD/visdial_amd/libvisdial_hip_sv.so
timeout 600 python scripts/mb_split_variants.py 10,11,12,13,1,2,3,4 1,2,3,4,5 > $OUT/r05_split_variants.txt 2>&1
# correctness of the candidates through the parity tests (the variant library reads VD_SPLIT_FWD / VD_SPLIT_BWD)
VD_SPLIT_FWD=2 VD_SPLIT_BWD=2 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "split9" > $OUT/r05_split_tests_v2.txt 2>&1
VD_SPLIT_FWD=3 VD_SPLIT_BWD=3 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "split9" > $OUT/r05_split_tests_v3.txt 2>&1
unset VD_LIB_PATH
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/r05_bench_fp32_a.json 2> $OUT/r05_bench_a.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --recurrence split9 > $OUT/r05_bench_split9_a.json 2>> $OUT/r05_bench_a.err
cat $OUT/r05_mfma_bf16_peak.txt $OUT/r05_split_variants.txt; tail -3 $OUT/r05_split_tests_v2.txt $OUT/r05_split_tests_v3.txt; cut -c1-400 $OUT/r05_bench_fp32_a.json $OUT/r05_bench_split9_a.json
