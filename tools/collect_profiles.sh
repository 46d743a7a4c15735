#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <tag>
# Leaves under gpurun_out/profiles_<tag>/: the bench JSON of a plain `python3 bench.py`, the rocprofv3
# kernel-trace --stats summary of the same command (the python program directly after `--`) and that run's JSON.
# The PMC passes per configuration are tools/collect_all_profiles.sh.
set -u
export TMPDIR=/tmp
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $O
python3 bench.py > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err || echo "bench failed"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/trace.log || echo "trace failed"
cat $O/bench_unprofiled.json | head -c 600
echo
echo collected
