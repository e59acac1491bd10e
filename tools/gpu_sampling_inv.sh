#!/bin/bash
# per-evaluation kernel inventory of DDIM-100 sampling at B = 256: tools/gpu_sampling_inv.sh TAG [VAR=val ...]
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace -d $out/${tag}_p2 -o samp -- python3 tools/run_sampling.py 256 100 2 > $out/${tag}_p2.log 2>&1
db2=$(find $out/${tag}_p2 -name '*results.db' | head -1)
python tools/eval_inventory.py $db2 45 > $out/${tag}_sampling_eval_inventory_b256.txt
rm -rf $out/${tag}_p2
head -30 $out/${tag}_sampling_eval_inventory_b256.txt
