#!/bin/bash
out=gpurun_out
python -m pytest tests -m gpu -q > $out/r03e_pytest.log 2>&1; tail -8 $out/r03e_pytest.log
ab() {
  env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-large-batch --no-sampling 2>/dev/null | tail -1 > /tmp/_ab.json
  python - "$*" <<'PY'
import json, sys
d = json.load(open('/tmp/_ab.json'))
print('%-40s ms/step %.3f median %.3f (%.0f img/s)' % (sys.argv[1], d['ms_per_step'], d['ms_per_step_median'], d['value']))
PY
}
{ ab X=0; ab IDF_CHAIN_BM=128; ab IDF_CHAIN_BM=64; ab IDF_GN_APPLY_BLOCKS=2048; ab IDF_GN_APPLY_BLOCKS=512; ab X=0; ab IDF_CHAIN_BM=128; } 2>&1 | tee $out/r03e_ab.txt
