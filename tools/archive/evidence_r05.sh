#!/bin/bash
# usage (GPU box, repo root): tools/evidence_r05.sh <tag>
# The measurements DESIGN.md quotes outside the per-configuration profiles, collected into gpurun_out/evidence_<tag>/
# (small text files; copied into profiles/ afterwards):
#   bench.json / bench_under_rocprof.json / bench_kernel_stats.csv      `python bench.py` unprofiled, and THE SAME command under
#                                                                        rocprofv3 --kernel-trace --stats (python directly after `--`)
#   bench_c2only_*                                                       the headline workload alone (one kernel size in the trace)
#   c3_bisect.txt        tools/archive/c3_bisect.py: why round 4's C3 profile (host-output calls) read 10 % faster than the bench
#   host_visible_*.txt   tools/host_visible.py: blocking oth_welch_exec per step, polling against hipStreamSynchronize,
#                        pilot in the kernel against its own launch
#   hv_kernel_stats.csv  the same loop under rocprofv3 --kernel-trace --stats (the finalize launch that signals)
#   ab_8k.txt            the 8192-point one-exchange builds against the 2 x 4096 kernels, interleaved on one box
#   kernel_resources.txt registers / scratch / LDS of every kernel in the shipped library
set -u
export TMPDIR=/tmp
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out/evidence_$TAG
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err || echo "bench failed"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py > $O/bench_under_rocprof.json 2> $O/trace.log || echo "trace failed"
cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
rm -rf $O/trace
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-extras --no-cpu-baseline > $O/bench_c2only_under_rocprof.json 2>> $O/trace.log || echo "c2-only trace failed"
cp $O/trace/*/*_kernel_stats.csv $O/bench_c2only_kernel_stats.csv 2>/dev/null
rm -rf $O/trace
timeout -k 10 600 python3 tools/archive/c3_bisect.py > $O/c3_bisect.txt 2>&1
python3 tools/host_visible.py 28 20 > $O/host_visible_poll.txt 2>&1
OTH_HOSTWAIT=sync python3 tools/host_visible.py 28 20 > $O/host_visible_sync.txt 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/host_visible.py 28 20 > /dev/null 2>> $O/trace.log
cp $O/trace/*/*_kernel_stats.csv $O/hv_kernel_stats.csv 2>/dev/null
rm -rf $O/trace
bash tools/archive/ab_8k.sh 2>&1 | grep -v amdgpu.ids > $O/ab_8k.txt
python3 tools/kernel_resources.py > $O/kernel_resources.txt 2>&1
head -c 300 $O/bench.json; echo; tail -4 $O/host_visible_poll.txt; echo collected
