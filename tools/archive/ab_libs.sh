#!/bin/bash
# usage (GPU box, repo root): tools/archive/ab_libs.sh <out-file-under-gpurun_out> <config> <reps> <rounds> <lib-tag> [<lib-tag> ...]
# Interleaved same-box A/B of tools/prof_driver.py <config> over library builds lib/libofdmtools_hip_<tag>.so
# ("default" = the shipped library): <rounds> passes over the tags, one line per run.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; CFG=$2; REPS=$3; ROUNDS=$4; shift 4
: > $OUT
for r in $(seq 1 $ROUNDS); do
    for tag in "$@"; do
        if [ "$tag" = default ]; then unset OFDM_TOOLS_HIP_LIB; else export OFDM_TOOLS_HIP_LIB=$GRAFT_REPO_ROOT/gr-ofdm_tools_amd/lib/libofdmtools_hip_$tag.so; fi
        echo "$tag $(python3 tools/prof_driver.py $CFG $REPS 2>&1 | grep GB/s | sed -e 's/(.*)//')" >> $OUT
    done
done
sort $OUT | awk '{print $1, $5}' | awk '{s[$1]+=$2; n[$1]++} END {for (k in s) printf "%s mean %.4f ms over %d runs\n", k, s[k]/n[k], n[k]}'
