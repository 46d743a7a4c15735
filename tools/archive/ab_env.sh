#!/bin/bash
# usage (GPU box, repo root): tools/archive/ab_env.sh <out-file-under-gpurun_out> <config> <reps> <rounds> <VAR> <value> [<value> ...]
# Interleaved same-box A/B of tools/prof_driver.py <config> over values of one environment variable (e.g. PROF_DETREND
# default exact): <rounds> passes over the values, one line per run, then the mean per value.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; CFG=$2; REPS=$3; ROUNDS=$4; VAR=$5; shift 5
: > $OUT
for r in $(seq 1 $ROUNDS); do
    for val in "$@"; do
        echo "$val $(env $VAR=$val python3 tools/prof_driver.py $CFG $REPS 2>&1 | grep GB/s | sed -e 's/(.*)//')" >> $OUT
    done
done
awk '{print $1, $5}' $OUT | awk '{s[$1]+=$2; n[$1]++} END {for (k in s) printf "%s mean %.4f ms over %d runs\n", k, s[k]/n[k], n[k]}'
