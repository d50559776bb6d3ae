#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/r05_gpu_tests_a.txt 2>&1
tail -n 15 $OUT/r05_gpu_tests_a.txt
VD_BENCH_FORCE_QUEUE_PROBE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/r05_queue_probe.json 2> $OUT/r05_queue_probe.err
grep -E "A/B|rank" $OUT/r05_queue_probe.err | tail -5; cut -c1-300 $OUT/r05_queue_probe.json
