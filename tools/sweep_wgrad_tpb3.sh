#!/bin/bash
# pixel tiles per block of the shared-tile 3x3 weight gradient (fewer blocks = fewer fp32-atomic bytes): tools/sweep_wgrad_tpb3.sh
B="bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-large-batch --no-sampling"
run() { env "$@" python $B 2>/dev/null | tail -1 | grep -o '"ms_per_step[a-z_]*": [0-9.]*' | tr '\n' ' '; echo " $@"; }
for r in 1 2; do
for t in 64 128 192 256; do run IDF_WGRAD_TPB3=$t; done
done
