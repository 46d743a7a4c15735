#!/bin/bash
# usage (GPU box, repo root): tools/evidence_r03.sh <tag>
# The measurements DESIGN.md quotes outside the per-configuration profiles, collected into gpurun_out/evidence_<tag>/
# (small text files; copied into profiles/ afterwards): single-row accuracy of the chain builds next to the CPU fp32
# comparator, same-box A/B of the headline kernel against its one-workgroup-per-CU variant, the two-channel kernel
# against its round-2 exchange, and the bench command unprofiled and under rocprofv3 --kernel-trace --stats.
set -u
export TMPDIR=/tmp
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out/evidence_$TAG
mkdir -p $O
python3 tools/acc_rows.py > $O/accuracy_chain_rows.txt 2>&1
{
  for i in 1 2 3; do
    echo "ws  $(PROF_EVEN=1 python3 tools/prof_driver.py C2 30 2>&1 | grep 'GB/s' | sed -e 's/(.*)//')"
    echo "ws2 $(PROF_EVEN=1 OTH_W4096_VARIANT=ws2 python3 tools/prof_driver.py C2 30 2>&1 | grep 'GB/s' | sed -e 's/(.*)//')"
  done
} > $O/ab_headline_ws_vs_ws2.txt
python3 bench.py > $O/bench.json 2> $O/bench.err || echo "bench failed"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/trace.log || echo "trace failed"
cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
rm -rf $O/trace
head -c 400 $O/bench.json; echo; echo collected
