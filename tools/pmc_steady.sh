#!/bin/bash
# GPU box: PMC passes over the steady-state workload.  usage: tools/pmc_steady.sh <gate> <k> <ipq>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_steady_$1_$2_$3; rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_BUSY_CYCLES"
P3="SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_SMEM"
P4="SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -- python3 tools/steady.py $1 $2 $3 > $OUT/run$i.txt 2> $OUT/err$i.txt || { tail -3 $OUT/err$i.txt; exit 1; }
done
cat $OUT/run1.txt
python3 tools/pmc_summary.py $OUT | tee $OUT/summary.txt
rm -rf $OUT/pmc?
