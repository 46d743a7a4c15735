#!/bin/bash
# usage (GPU box, repo root): tools/evidence_r04.sh <tag>
# The measurements DESIGN.md quotes outside the per-configuration profiles, collected into gpurun_out/evidence_<tag>/
# (small text files; copied into profiles/ afterwards): the bench command unprofiled and THE SAME command under
# rocprofv3 --kernel-trace --stats (the python program directly after `--`), and same-box interleaved A/Bs of the
# round-4 kernels against the ones they replace (tuning variant 16k4 = round 3's 4 x 4096 build).
set -u
export TMPDIR=/tmp
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out/evidence_$TAG
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err || echo "bench failed"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py > $O/bench_under_rocprof.json 2> $O/trace.log || echo "trace failed"
cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
rm -rf $O/trace
# the headline workload alone: welch4096ws_kernel runs at one size only, so the trace average compares with the live figure
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-extras --no-cpu-baseline > $O/bench_c2only_under_rocprof.json 2>> $O/trace.log || echo "c2-only trace failed"
cp $O/trace/*/*_kernel_stats.csv $O/bench_c2only_kernel_stats.csv 2>/dev/null
rm -rf $O/trace
{
  echo "# BASELINE config 5 (64 x 2^22 samples, 16384-pt rect mean), HIP-event average of 40 launches, interleaved on one box"
  for i in 1 2 3; do
    echo "welch16k1x_pipe (default)   $(python3 tools/prof_driver.py C5 40 2>&1 | grep 'GB/s' | sed -e 's/(.*)//')"
    echo "welch16k1x plain            $(OTH_W4096_VARIANT=16kplain python3 tools/prof_driver.py C5 40 2>&1 | grep 'GB/s' | sed -e 's/(.*)//')"
    echo "welch16k<0,4> (round 3)     $(OTH_W4096_VARIANT=16k4 python3 tools/prof_driver.py C5 40 2>&1 | grep 'GB/s' | sed -e 's/(.*)//')"
  done
} > $O/ab_c5_kernels.txt
{
  echo "# Welch 16384-pt Hann, 50 % overlap, detrend constant, 2^27 samples, interleaved on one box"
  for i in 1 2 3; do
    echo "welch16k1x_half<2> (default) $(python3 tools/prof_driver.py w16384 30 2>&1 | grep 'GB/s' | sed -e 's/(.*)//')"
    echo "welch16k<2,4,HALF> (round 3) $(OTH_W4096_VARIANT=16k4 python3 tools/prof_driver.py w16384 30 2>&1 | grep 'GB/s' | sed -e 's/(.*)//')"
  done
} > $O/ab_w16384_kernels.txt
head -c 300 $O/bench.json; echo; cat $O/ab_c5_kernels.txt $O/ab_w16384_kernels.txt; echo collected
