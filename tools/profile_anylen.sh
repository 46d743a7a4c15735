#!/bin/bash
# usage (GPU box, repo root): tools/profile_anylen.sh <tag> <config ...>      e.g.  tools/profile_anylen.sh r06 w32768 w65536 w1000
# The any-length routes (csrc/fft_any.hip, fft_tl.hip) are SEVERAL launches per call (block sums / means, K1, K2[, K3]): this
# runs tools/prof_driver.py <config> under rocprofv3 in separate passes (--kernel-trace --stats | --pmc FETCH_SIZE |
# --pmc WRITE_SIZE; the python program directly after `--`) and leaves gpurun_out/profiles_<tag>/<tag>_<config>.txt
# (tools/anylen_summary.py: per-kernel averages, the per-call sum, the fraction of the byte roofline, HBM bytes per call).
set -u
export TMPDIR=/tmp
TAG=$1; shift
REPS=5
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
for CFG in "$@"; do
    O=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$CFG
    mkdir -p $O
    run() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" --output-format csv -d $O/$name -- python3 tools/prof_driver.py $CFG $REPS > $O/$name.log 2>&1 || echo "pass $name failed rc=$?"; }
    run trace --kernel-trace --stats
    run fetch --pmc FETCH_SIZE
    run write --pmc WRITE_SIZE
    python3 tools/anylen_summary.py $O $CFG $REPS > $GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG/${TAG}_$CFG.txt 2>&1
    tail -4 $GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG/${TAG}_$CFG.txt
    rm -rf $O/fetch $O/write      # per-dispatch CSVs: too large to carry back
done
echo collected
