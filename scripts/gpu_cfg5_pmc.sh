#!/bin/bash
# PMC passes of config 5 (one repetition of the driver's window): issue, waits, LDS -> gpurun_out/<tag>_cfg5_pmc.txt
TAG=${1:-r4}
A="--config 5 --no-cpu-baseline --reps 1 --min-seconds 0 --steps 10 --warmup 2"
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; do
  t=$(echo $set | cut -d' ' -f1)
  LASTN=0 bash scripts/rocprof_pmc.sh ${TAG}_c5_$t "$set" -- python bench.py $A 2>&1 | grep -E "k_nn_logprobs_h|k_nn_grad|k_acyc_bfw" >> gpurun_out/${TAG}_cfg5_pmc.txt
done
cat gpurun_out/${TAG}_cfg5_pmc.txt
