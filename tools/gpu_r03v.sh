#!/bin/bash
# side-stream weight gradients: A/B on one box.  tools/gpu_r03v.sh
export TMPDIR=/tmp
out=gpurun_out/r03v_ab_wgrad_side.txt
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad" 2>&1 | tail -2 > $out
python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "train_step or data_parallel" 2>&1 | tail -2 >> $out
for rep in 1 2; do
for cfg in "IDF_WGRAD_SIDE=0" "IDF_WGRAD_SIDE=1" "IDF_WGRAD_SIDE=1 IDF_WGRAD_EVERY=8" "IDF_WGRAD_SIDE=1 IDF_WGRAD_EVERY=16" "IDF_WGRAD_SIDE=1 IDF_WGRAD_EVERY=32"; do
  echo "== $cfg" >> $out
  env $cfg python bench.py --no-cpu-baseline --no-large-batch --no-sampling --no-roofline --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_median'))" >> $out
done
done
cat $out
