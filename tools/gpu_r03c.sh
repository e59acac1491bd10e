#!/bin/bash
out=gpurun_out
python tools/bf16_grad_profile.py > $out/r03c_gradprof.txt 2>&1; grep -v Warning $out/r03c_gradprof.txt | tail -90
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "chain" > $out/r03c_kernels.log 2>&1; tail -5 $out/r03c_kernels.log
for v in "IDF_BWD_CHAIN=0" "IDF_BWD_LAZY=0" "IDF_BWD_LAZY=1"; do
  env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-large-batch --no-sampling 2>/dev/null | tail -1 > /tmp/_ab.json
  python - "$v" <<'PY'
import json, sys
d = json.load(open('/tmp/_ab.json'))
print('%-18s ms/step %.3f median %.3f (%.0f img/s)' % (sys.argv[1], d['ms_per_step'], d['ms_per_step_median'], d['value']))
PY
done 2>&1 | tee $out/r03c_ab.txt
for v in "IDF_BWD_CHAIN=0" "IDF_BWD_LAZY=1"; do
  env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-large-batch --no-sampling 2>/dev/null | tail -1 > /tmp/_ab.json
  python - "$v" <<'PY'
import json, sys
d = json.load(open('/tmp/_ab.json'))
print('%-18s ms/step %.3f median %.3f (%.0f img/s)' % (sys.argv[1], d['ms_per_step'], d['ms_per_step_median'], d['value']))
PY
done 2>&1 | tee -a $out/r03c_ab.txt
