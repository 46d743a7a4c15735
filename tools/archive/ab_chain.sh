#!/bin/bash
# usage (GPU box, repo root): tools/archive/ab_chain.sh <out-file-under-gpurun_out> <lib-tag> [<lib-tag> ...]
# Accuracy (tools/acc_rows.py) and whole-push time (tools/prof_driver.py chainN) of the periodogram chain for A/B
# library builds lib/libofdmtools_hip_<tag>.so ("default" = the shipped library).
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
: > $OUT
for tag in "$@"; do
    if [ "$tag" = default ]; then unset OFDM_TOOLS_HIP_LIB; else export OFDM_TOOLS_HIP_LIB=$GRAFT_REPO_ROOT/gr-ofdm_tools_amd/lib/libofdmtools_hip_$tag.so; fi
    echo "==== $tag" >> $OUT
    python3 tools/acc_rows.py 2>&1 | grep -E "HIP|per-row" | grep -v radix >> $OUT
    for n in 256 512 1024 2048 4096; do python3 tools/prof_driver.py chain$n 40 2>&1 | grep GB/s >> $OUT; done
done
echo done
