#!/bin/bash
# A/B of library variants on the train step only: tools/ab_libs_train.sh default v1 v2 ...   (three rounds each)
for round in 1 2 3; do
for v in "$@"; do
  if [ "$v" = default ]; then unset IDF_LIB; else export IDF_LIB=$PWD/infodiffusion_amd/variants/libinfodiff_hip_$v.so; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-large-batch --no-sampling 2>/dev/null | tail -1 > /tmp/_ab.json
  python - "$v" <<'PY'
import json, sys
d = json.load(open('/tmp/_ab.json'))
print('%-8s ms/step %.3f median %.3f' % (sys.argv[1], d['ms_per_step'], d['ms_per_step_median']))
PY
done
done
