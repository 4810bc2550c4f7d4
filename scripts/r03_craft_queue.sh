#!/bin/bash
# The massless sweep with and without the work queue, on both craft populations (profiles/r03_craft_queue.md).
# usage (GPU box): bash scripts/r03_craft_queue.sh
mkdir -p gpurun_out
for pop in transfer mixed; do
  for q in auto 1 0; do
    if [ $pop = mixed ]; then args="--craft 524288 --craft-days 2 --steps 2"; else args="--craft 262144 --craft-days 0.25 --steps 3"; fi
    if [ $q = auto ]; then unset EPH_CRAFT_QUEUE; else export EPH_CRAFT_QUEUE=$q; fi
    timeout 600 python bench.py --workload craft --population $pop $args --no-cpu-baseline \
      > gpurun_out/r03_craft_${pop}_q${q}.json 2> gpurun_out/r03_craft_${pop}_q${q}.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r03_craft_${pop}_q${q}.json").read().strip().splitlines()[-1])
    print("${pop} queue=${q}: %.3e craft-steps/s, ms_per_step %.2f, launch_us %.0f, fp64 frac %.3f, divergence %.2f" % (
        d["value"], d["ms_per_step"], d["roofline"]["launch_us"], d["fp64"]["frac"], d["divergence"]["attempts_max_over_mean_per_wave"]))
except Exception as e:
    print("${pop} queue=${q}: failed", e, open("gpurun_out/r03_craft_${pop}_q${q}.err").read()[-600:])
PY
  done
done
