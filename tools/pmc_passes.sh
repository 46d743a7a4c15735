#!/bin/bash
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> <config> [reps] [kernel-pattern]
# Separate rocprofv3 passes (kernel trace, SQ, LDS, HBM read, HBM write) of tools/prof_driver.py <config>;
# the python program goes directly after `--`.  Writes <outdir>/summary.txt (tools/pmc_summary.py).
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
CFG=${2:-C2}
REPS=${3:-3}
PAT=${4:-welch4096}
mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" --output-format csv -d $O/$name -- python3 tools/prof_driver.py $CFG $REPS > $O/$name.log 2>&1 || echo "pass $name failed rc=$?"; }
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py $O "$PAT" > $O/summary.txt 2>&1
cat $O/trace.log | tail -3
echo "passes done: $CFG"
