#!/bin/bash
# power / shader clock / temperature the SMI reports while ONE convolution shape loops (un-profiled): usage smi_power_during_conv.sh "<conv_bench args>" [env...]
ARGS="$1"; shift
env "$@" python scripts/conv_bench.py $ARGS &
BP=$!
sleep 6
for i in 1 2 3 4; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i -E "sclk|power|junction|Temperature \(Sensor (edge|hot)" | tr '\n' ' '; echo
  sleep 0.7
done
wait $BP
