#!/bin/bash
# PMC counters of the split fused layer kernel (run on the GPU box from the repo root): tools/prof_sl.sh [variant]
cd /tmp && export TMPDIR=/tmp
V=${1:-default}
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /root/repo/gpurun_out/pmc_sl_$tag
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d /root/repo/gpurun_out/pmc_sl_$tag -o p -- python3 /root/repo/tools/sl_tune.py one $V > /root/repo/gpurun_out/pmc_sl_$tag.log 2>&1 || echo "pass $tag failed"
  python3 /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/pmc_sl_$tag split_layer_kernel
done
