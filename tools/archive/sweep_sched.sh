for cfg in w8192 w16384; do
  echo "== $cfg default: $(python3 tools/prof_driver.py $cfg 20 2>&1 | grep -E 'recipe|GB/s algo' | sed 's/(.*)//' | tr '\n' ' ')"
  for sched in 0 1 2; do for chunk in 4 8 16 32; do
    echo "$cfg sched=$sched chunk=$chunk: $(OTH_W4096_SCHED=$sched OTH_W4096_CHUNK=$chunk python3 tools/prof_driver.py $cfg 20 2>&1 | grep -E 'GB/s algo' | sed 's/(.*)//')"
  done; done
done
