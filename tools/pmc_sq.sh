#!/bin/bash
# Collect SQ issue/stall counters for the bench kernels (one rocprofv3 pass per counter group).
# usage (on the GPU box, from the repo root): tools/pmc_sq.sh <tag> [bench args...]
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
g1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
g2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_LDS_BANK_CONFLICT"
g3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SENDMSG GRBM_GUI_ACTIVE"
# lane utilisation of the vector instructions: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64) = active lanes per issued VALU cycle
g4="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_INSTS_VALU"
i=0
for g in "$g1" "$g2" "$g3" "$g4"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $g -d $out/g$i --output-format csv -- python3 bench.py --no-parity --quick "$@" > $out/g$i.log 2>&1
  echo "group $i rc=$?"
done
python3 tools/pmc_summary.py $out
