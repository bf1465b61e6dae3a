#!/bin/bash
# PMC passes over k_bge_soft_mf (scripts/gpu_soft_bench.py): wave cycles, waits, instruction mix.  usage (GPU box): scripts/pmc_soft_mf.sh <tag>
TAG=${1:-r4}
mkdir -p gpurun_out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  bash scripts/rocprof_pmc.sh soft_$tag "$set" -- python scripts/gpu_soft_bench.py 2>&1 | grep -i "k_bge_soft" >> gpurun_out/${TAG}_soft_mf_pmc.txt
done
cat gpurun_out/${TAG}_soft_mf_pmc.txt
