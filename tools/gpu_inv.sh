#!/bin/bash
# step inventory under an env setting: tools/gpu_inv.sh TAG [VAR=val ...]
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats -d $out/${tag}_p1 -o train -- python3 bench.py --no-cpu-baseline --no-large-batch --no-sampling --no-roofline > $out/${tag}_p1.log 2>&1
db=$(find $out/${tag}_p1 -name '*results.db' | head -1)
python tools/step_inventory.py $db 90 > $out/${tag}_step_inventory.txt
python tools/level_inventory.py $db 90 > $out/${tag}_level_inventory.txt
python tools/overlap_inventory.py $db > $out/${tag}_overlap.txt
rm -rf $out/${tag}_p1
head -50 $out/${tag}_step_inventory.txt
