#!/bin/bash
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> [log2n] -- separate rocprofv3 passes (kernel trace, SQ, LDS, HBM)
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
L=${2:-28}
mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" --output-format csv -d $O/$name -- python3 tools/prof_driver.py $L 2 > $O/$name.log 2>&1 || echo "pass $name failed rc=$?"; }
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
echo passes done
