#!/bin/bash
# Run on the GPU box: PMC passes (never combined with --stats / sys traces) over tools/time_encoder.py for the fused
# encoder kernels.  Output: gpurun_out/pmc_encoder_<i>/.../*_counter_collection.csv + a counter listing.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|GRBM|TCC|TCP|TA)_[A-Z0-9_]+" | sort -u > $O/pmc_counters.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_IFETCH SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_encoder_$i -- python3 $R/tools/time_encoder.py > $O/pmc_encoder_$i.log 2>&1
done
