#!/bin/bash
# step inventory of the bench's training step under rocprofv3 (kernel trace): tools/inv_step.sh <tag> [ENV=VALUE ...]
tag=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_p1 -o train -- python3 bench.py --no-cpu-baseline --no-large-batch --no-sampling --no-roofline --no-dp-probe > gpurun_out/${tag}_p1.log 2>&1
db=$(find gpurun_out/${tag}_p1 -name '*results.db' | head -1)
python tools/step_inventory.py $db 120 > gpurun_out/${tag}_step_inventory.txt
rm -rf gpurun_out/${tag}_p1
head -${LINES_OUT:-60} gpurun_out/${tag}_step_inventory.txt
