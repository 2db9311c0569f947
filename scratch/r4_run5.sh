#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_node_team.py -m gpu -q -x 2>&1 | tail -15
O=/tmp/pmc_gemm
mkdir -p $O gpurun_out/r4_pmc_gemm
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/scratch/gemm_one.py N T 20000 20000 624 > gpurun_out/r4_pmc_gemm/p$i.log 2>&1
  tail -2 gpurun_out/r4_pmc_gemm/p$i.log
done
python3 scratch/pmc_gemm_summarise.py $O/p1 $O/p2 $O/p3 $O/p4 > gpurun_out/r4_pmc_gemm_summary.txt 2>&1
cat gpurun_out/r4_pmc_gemm_summary.txt
