#!/bin/bash
# re-sweep of the pixel tiles per block / minimum blocks of the weight-gradient classes that still use the row-split kernel
# (1x1, stride 2, up-sampling) now that the 3x3 stride-1 class has its own plan.  tools/gpu_r03_tpb.sh
for rep in 1 2; do
for cfg in "IDF_WGRAD_TPB3=64 IDF_WGRAD_MINB3=16" "IDF_WGRAD_TPB3=32 IDF_WGRAD_MINB3=16" "IDF_WGRAD_TPB3=48 IDF_WGRAD_MINB3=16" "IDF_WGRAD_TPB3=96 IDF_WGRAD_MINB3=16" "IDF_WGRAD_TPB3=64 IDF_WGRAD_MINB3=48" "IDF_WGRAD_TPB3=128 IDF_WGRAD_MINB3=8"; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-large-batch --no-sampling 2>/dev/null | tail -1 > /tmp/_ab.json
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open('/tmp/_ab.json'))
print('%-40s ms/step %.3f median %.3f' % (sys.argv[1], d['ms_per_step'], d['ms_per_step_median']))
PY
done
done
