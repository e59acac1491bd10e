#!/bin/bash
# A/B of one environment switch on the train step AND DDIM-100 sampling, alternating, on one box:
#   tools/ab_env.sh VAR v1 v2 [rounds]
var=$1; a=$2; b=$3; rounds=${4:-2}
for r in $(seq $rounds); do
for v in $a $b; do
  env $var=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-large-batch --no-dp-probe 2>/dev/null | tail -1 > /tmp/_ab.json
  python - "$var" "$v" <<'PY'
import json, sys
d = json.load(open('/tmp/_ab.json'))
print('%s=%-6s ms/step %.3f median %.3f (%.0f img/s)  DDIM-100 B256 %.1f img/s' % (sys.argv[1], sys.argv[2], d['ms_per_step'],
      d['ms_per_step_median'], d['value'], d['sampling']['value']))
PY
done
done
