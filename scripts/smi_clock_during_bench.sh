#!/bin/bash
# sample the shader clock the SMI reports while bench.py's inference loop runs un-profiled ("show the un-profiled clock")
python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-train-leg > gpurun_out/r02_bench_long.json 2> /dev/null &
BP=$!
sleep 14
for i in $(seq 1 12); do
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
  sleep 0.5
done
wait $BP
tail -c 600 gpurun_out/r02_bench_long.json
