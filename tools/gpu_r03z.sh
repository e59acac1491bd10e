#!/bin/bash
# conditioning path as one entry point: A/B on one box.  tools/gpu_r03z.sh
export TMPDIR=/tmp
out=gpurun_out/r03z_ab_temb_fused.txt
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conditioning_path" 2>&1 | tail -1 > $out
python tools/bench_temb_film.py 2>&1 | grep "replayed" >> $out
for rep in 1 2 3; do
for cfg in "IDF_TEMB_FUSED=0" "IDF_TEMB_FUSED=1"; do
  echo "== $cfg" >> $out
  env $cfg python bench.py --no-cpu-baseline --no-large-batch --no-sampling --no-roofline --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_median'))" >> $out
done
done
cat $out
