#!/bin/bash
# usage (on the GPU box, repo root): tools/collect_all_profiles.sh <tag> [config ...]      (default: all)
# PUBLISH_ROUND=r03 in the environment: summarise on the box (tools/publish_profiles.py <tag> r03 gpurun_out/profiles_<tag>)
# and drop the per-dispatch CSVs, which together exceed what gpurun copies back; then `cp gpurun_out/profiles_<tag>/* profiles/`.
# rocprofv3 kernel-trace + PMC passes (tools/pmc_passes.sh) for every configuration DESIGN.md quotes; leaves
# gpurun_out/prof_<tag>_<config>/summary.txt.  tools/publish_profiles.py <tag> <round> copies them into profiles/.
set -u
TAG=$1; shift
ONLY=" $* "
for spec in "C2:welch4096ws" "C2fast:welch4096ws" "C3:csd4096ws" "C4:welch4096ws" "C4ref:welch4096_kernel" "C5:welch16k1x_pipe" "C5old:welch16k_kernel" \
            "scan8192:welch16k1x_pipe" "w256:seg_kernel" "w512:seg_kernel" "w1024:segws_kernel" "w2048:segws_kernel" "w8192:welch8kws" \
            "w16384:welch16k1x_half" "p1024:seg_kernel" "p2048:seg_kernel" "p8192:welch16k" "p16384:welch16k" \
            "chain256:seg_kernel" "chain512:seg_kernel" "chain1024:seg_kernel" "chain2048:seg_kernel" "chain4096:seg_kernel" \
            "chain8192:chain16k" "chain16384:chain16k"; do
    cfg=${spec%%:*}; pat=${spec##*:}
    if [ "$ONLY" != "  " ] && [[ "$ONLY" != *" $cfg "* ]]; then continue; fi
    drv=$cfg; var=
    if [ "$cfg" = "C5old" ]; then drv=C5; var=16k4; fi
    unset PROF_DETREND
    if [ "$cfg" = "C2fast" ]; then drv=C2; export PROF_DETREND=fast; fi      # OTH_DETREND_CONSTANT_FAST: the build without the pilot
    OTH_W4096_VARIANT=$var tools/pmc_passes.sh prof_${TAG}_${cfg} $drv 5 $pat > /dev/null 2>&1
    echo "== $cfg"; grep -E "GB/s" $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_${cfg}/trace.log | tail -1
done
if [ -n "${PUBLISH_ROUND:-}" ]; then
    python3 tools/publish_profiles.py $TAG $PUBLISH_ROUND $GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG > /dev/null
    rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_*
fi
echo collected
