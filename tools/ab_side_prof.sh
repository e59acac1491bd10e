#!/bin/bash
# side-stream weight gradients with and without the profiler attached, one box
export TMPDIR=/tmp
B="bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-large-batch --no-sampling"
for r in 1 2; do
for s in 0 1; do
  export IDF_WGRAD_SIDE=$s
  python $B 2>/dev/null | tail -1 | grep -o '"ms_per_step[a-z_]*": [0-9.]*' | tr '\n' ' '; echo " SIDE=$s plain"
  rocprofv3 --kernel-trace -d /tmp/_sp$s$r -o t -- python3 $B 2>/dev/null | grep -o '"ms_per_step[a-z_]*": [0-9.]*' | tr '\n' ' '; echo " SIDE=$s rocprofv3 --kernel-trace"
done
done
