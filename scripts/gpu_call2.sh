#!/bin/bash
OUT=$PWD/gpurun_out; ROOT=$PWD; mkdir -p $OUT
export VD_LIB_PATH=$PWD/visdial_amd/libvisdial_hip_sv.so
MB_ZERO=1 timeout 300 python scripts/mb_split_variants.py 10,11,12,1,2 4 > $OUT/r05_split_variants_zero.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/r05_counters_list.txt 2>&1
for PASS in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM"; do
  rm -rf $OUT/pmc_tmp
  MB_REPS=1 timeout 600 rocprofv3 --pmc $PASS -d $OUT/pmc_tmp -o p -- python $ROOT/scripts/mb_split_variants.py 10,11,12,1,2,4 1,4 > $OUT/r05_pmc_run.log 2>&1
  DBP=$(ls $OUT/pmc_tmp/*.db 2>/dev/null | head -1)
  echo "## $PASS" >> $OUT/r05_pmc_split_variants.txt
  if [ -n "$DBP" ]; then python $ROOT/scripts/rocpd_pmc.py $DBP | grep -v "^# pmc_events" | grep -A9 -E "^gemm_split|^gemm_f32_glds" >> $OUT/r05_pmc_split_variants.txt; else tail -5 $OUT/r05_pmc_run.log >> $OUT/r05_pmc_split_variants.txt; fi
done
rm -rf $OUT/pmc_tmp
cat $OUT/r05_split_variants_zero.txt; cat $OUT/r05_pmc_split_variants.txt | cut -c1-150
