#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <tag>
# Leaves under gpurun_out/profiles_<tag>/: the rocprofv3 kernel-trace --stats summary of the bench.py command,
# that run's bench JSON, and the PMC passes (separate invocations, counters only) on the same workload.
set -u
export TMPDIR=/tmp
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $O
python3 bench.py > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err || echo "bench failed"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/trace.log || echo "trace failed"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" --output-format csv -d $O/$name -- python3 tools/prof_driver.py 28 3 > $O/$name.log 2>&1 || echo "pass $name failed rc=$?"; }
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# calibration of FETCH_SIZE on a known byte count in a wide coalesced read (the read probe: 2 GiB per launch)
python3 tools/pmc_summary.py $O welch4096 > $O/summary_welch4096.txt 2>&1
python3 tools/pmc_summary.py $O read_probe > $O/summary_read_probe.txt 2>&1
cat $O/bench_unprofiled.json
echo collected
