# usage (GPU box): bash tools/archive/ab_half.sh <tag> [rounds]   - interleaved w16384 / w8192, shipped library against lib/libofdmtools_hip_<tag>.so
TAG=$1; R=${2:-3}
for i in $(seq 1 $R); do
  for cfg in w16384 w8192; do
    for det in default fast; do
      [ $det = fast ] && export PROF_DETREND=fast || unset PROF_DETREND
      echo "shipped $det $(python3 tools/prof_driver.py $cfg 20 2>&1 | grep GB/s | sed 's/(.*)//')"
      echo "$TAG $det $(OFDM_TOOLS_HIP_LIB=$GRAFT_REPO_ROOT/gr-ofdm_tools_amd/lib/libofdmtools_hip_$TAG.so python3 tools/prof_driver.py $cfg 20 2>&1 | grep GB/s | sed 's/(.*)//')"
    done
  done
done
