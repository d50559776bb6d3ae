#!/bin/bash
# round-end check on the GPU box: the whole -m gpu suite, smoke(), then the measurement recipe
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 2400 python -m pytest tests -m gpu -x -q > "$OUT/r06_gpu_tests.txt" 2>&1
tail -n 6 "$OUT/r06_gpu_tests.txt"
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
bash scripts/measure_round.sh r06 > "$OUT/r06_measure.log" 2>&1
tail -c 400 "$OUT/r06_measure.log"
