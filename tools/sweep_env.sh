#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ... -- runs bench.py once per value, prints ms_per_step (mean, median)
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > /tmp/_sweep.json
  python - "$var" "$v" <<'PY'
import json, sys
d = json.load(open('/tmp/_sweep.json'))
print(sys.argv[1], sys.argv[2], 'ms/step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'value', d['value'])
PY
done
