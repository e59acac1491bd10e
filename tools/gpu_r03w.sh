#!/bin/bash
# step time against the batch size (is the B = 32 step a latency chain?) + the 1-rank RCCL test.  tools/gpu_r03w.sh
export TMPDIR=/tmp
out=gpurun_out/r03w_batch_sweep.txt
python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "data_parallel_path" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 > $out
for b in 8 16 32 64; do
  echo "== batch $b" >> $out
  python bench.py --batch $b --no-cpu-baseline --no-large-batch --no-sampling --no-roofline --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_median'))" >> $out
done
cat $out
