for i in 1 2; do
for cfg in scan8192 w8192 w16384; do
  PROF_VARIANT=16k4 python tools/prof_driver.py $cfg 20 | grep -E "GB/s" | sed "s/^/old /;s/(.*)//"
  python tools/prof_driver.py $cfg 20 | grep -E "GB/s|recipe" | sed "s/^/new /;s/(.*)//" | cut -c1-150
done
OTH_CHAIN16K=old python tools/prof_driver.py chain8192 20 | grep GB/s | sed "s/^/old /;s/(.*)//"
python tools/prof_driver.py chain8192 20 | grep GB/s | sed "s/^/new /;s/(.*)//"
OTH_CHAIN16K=old python tools/prof_driver.py chainr8192 20 | grep GB/s | sed "s/^/old /;s/(.*)//"
python tools/prof_driver.py chainr8192 20 | grep GB/s | sed "s/^/new /;s/(.*)//"
done
