#!/bin/bash
# SQ counter passes over the fused attention kernels (tools/attn_bench.py: encoder self-attention, decoder self / cross).
# Three separate rocprofv3 --pmc runs (8 SQ slots each), the program directly behind `--`.
# usage (on the GPU box, through gpurun): bash tools/attn_pmc.sh r02
set -e -o pipefail
tag=${1:-r02}
out=gpurun_out/attn_pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32"
C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
i=0
for set in "$A" "$B" "$C"; do
  i=$((i+1))
  rm -rf /tmp/attn_pmc_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/attn_pmc_$i -o p -- python tools/attn_bench.py 3 > $out/pass$i.log 2> $out/pass$i.err
  python tools/pmc_kernels.py /tmp/attn_pmc_$i flash_ > $out/pass$i.txt
  echo "pass $i done"
done
cat $out/pass1.txt $out/pass2.txt $out/pass3.txt > $out/flash_counters.txt
