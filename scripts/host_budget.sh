#!/bin/bash
# Host budget of the step: the same bench line with the CPUs one rank of an 8-rank job would get.
#   run on the GPU box: gpurun -- 'bash scripts/host_budget.sh r06'
#   A  all CPUs of the box's quota              B  taskset -c 0,1 (16-CPU quota / 8 ranks)
#   C  taskset -c 0,1 + 14 busy-loop processes on CPUs 2-15 (the other seven ranks' hosts at 100 %)
TAG=${1:-r06}
EXTRA=${2:-}
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
F="$OUT/${TAG}_host_budget.txt"
FLAGS="--steps 40 --warmup 8 --no-alt --no-other-configs --no-cpu-baseline $EXTRA"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('%-58s ms/step %.3f  median %.3f  p10-p90 %s' % (sys.argv[1], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_p10_p90']))" "$1"; }
{
echo "# nproc=$(nproc) quota=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null) flags: $FLAGS"
python bench.py $FLAGS 2>/dev/null | line "A all CPUs"
taskset -c 0,1 python bench.py $FLAGS 2>/dev/null | line "B taskset -c 0,1"
PIDS=""
for c in 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do taskset -c $c python -c "
while True: pass" & PIDS="$PIDS $!"; done
sleep 1
taskset -c 0,1 python bench.py $FLAGS 2>/dev/null | line "C taskset -c 0,1 + 14 burners on CPUs 2-15"
taskset -c 0 python bench.py $FLAGS 2>/dev/null | line "D taskset -c 0 (ONE CPU) + 14 burners"
kill $PIDS 2>/dev/null
wait 2>/dev/null
python bench.py $FLAGS 2>/dev/null | line "A' all CPUs again (box drift)"
} > "$F" 2>&1
cat "$F"
