#!/bin/bash
# Samples rocm-smi (socket power, sclk, mclk) while one configuration's kernel runs back to back: shows which
# kernels run into the board's power limit and what clock the part sustains under them.
# usage: tools/power_probe.sh <config> [seconds]      (configs as tools/prof_driver.py)
cfg=${1:-C2}; secs=${2:-7}
PROF_LOOP_SECS=$secs python tools/prof_driver.py "$cfg" 20 &
pid=$!
sleep 3.5
for i in 1 2 3 4; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk|mclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.6
done
wait $pid
