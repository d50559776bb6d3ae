#!/bin/bash
# Cycle budget of the configs[4] step kernels: SQ counters per kernel over scripts/mb_c16.py (the compact-bf16 option
# recurrence alone at the headline shape), and over the split9 step kernels for comparison.  Separate --pmc passes (8 SQ slots each).
#   gpurun -- 'bash scripts/pmc_budget.sh r06'
TAG=${1:-r06}
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
F="$OUT/${TAG}_pmc_budget.txt"; : > "$F"
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u > "$OUT/${TAG}_sq_counters.txt"
pass() {  # $1 = label, $2 = counters, $3... = target command
  local label=$1 ctr=$2; shift 2
  rm -rf "$OUT/${TAG}_pmcb"
  rocprofv3 --pmc $ctr -d "$OUT/${TAG}_pmcb" -o p -- "$@" > /dev/null 2>> "$OUT/${TAG}_pmc_budget.err"
  local db=$(ls "$OUT/${TAG}_pmcb"/*.db 2>/dev/null | head -1)
  echo "## $label :: $ctr" >> "$F"
  [ -n "$db" ] && python "$ROOT/scripts/rocpd_pmc.py" "$db" | grep -v "^# pmc_events" | grep -A12 -E "^gemm_f32_glds|^gemm_f32<|^gemm_split" | grep -v "^--" >> "$F"
  rm -rf "$OUT/${TAG}_pmcb"
}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
P3="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
for P in "$P1" "$P2" "$P3"; do
  pass "configs[4] compact bf16 recurrence (mb_c16.py)" "$P" python "$ROOT/scripts/mb_c16.py"
  pass "split9 step kernels (pmc_target.py split9)" "$P" python "$ROOT/scripts/pmc_target.py" split9
done
cd "$ROOT"; wc -l "$F"; tail -5 "$OUT/${TAG}_pmc_budget.err"
