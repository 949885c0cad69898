#!/bin/bash
# Sample clocks / power with rocm-smi while the bench loop runs (diagnostic: is the chip power-limited under these kernels?).
python bench.py --no-cpu-baseline --steps 8000 --warmup 3 > gpurun_out/clk_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use|fclk" | tr '\n' ';'
  echo
  sleep 2
done
wait $BP
cut -c1-160 gpurun_out/clk_bench.json
