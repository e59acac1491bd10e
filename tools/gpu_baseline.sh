#!/bin/bash
# One box: bench line + step inventory under rocprofv3 (kernel trace) into gpurun_out/<tag>_*.
# usage: tools/gpu_baseline.sh <tag>
tag=$1
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-large-batch > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 1500 $out/${tag}_bench.json
rocprofv3 --kernel-trace --stats -d $out/${tag}_p1 -o train -- python3 bench.py --no-cpu-baseline --no-large-batch --no-roofline --no-sampling > $out/${tag}_p1.log 2>&1
db=$(find $out/${tag}_p1 -name '*results.db' | head -1)
python tools/step_inventory.py $db 90 > $out/${tag}_step_inventory.txt
rm -rf $out/${tag}_p1
head -40 $out/${tag}_step_inventory.txt
