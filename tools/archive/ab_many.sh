# usage (GPU box): bash tools/archive/ab_many.sh <lib-tag> <rounds> <config> [<config> ...]   - interleaved prof_driver runs, shipped library against lib/libofdmtools_hip_<tag>.so
TAG=$1; R=$2; shift 2
for i in $(seq 1 $R); do
  for cfg in "$@"; do
    echo "shipped $(python3 tools/prof_driver.py $cfg 20 2>&1 | grep GB/s | sed 's/(.*)//')"
    echo "$TAG $(OFDM_TOOLS_HIP_LIB=$GRAFT_REPO_ROOT/gr-ofdm_tools_amd/lib/libofdmtools_hip_$TAG.so python3 tools/prof_driver.py $cfg 20 2>&1 | grep GB/s | sed 's/(.*)//')"
  done
done | sort | awk '{k=$1" "$2; s[k]+=$5; n[k]++} END {for (k in s) printf "%s mean %.4f ms over %d runs\n", k, s[k]/n[k], n[k]}' | sort -k2
