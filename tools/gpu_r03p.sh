#!/bin/bash
out=gpurun_out
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad" 2>&1 | tail -2
python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "arena or up_block or bf16_train_step_celeba_at" 2>&1 | tail -2
ab() {
  env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-large-batch --no-sampling 2>/dev/null | grep metric > /tmp/_ab.json
  python - "$*" <<'PY'
import json, sys
d = json.load(open('/tmp/_ab.json'))
print('%-40s ms/step %.3f median %.3f (%.0f img/s)' % (sys.argv[1], d['ms_per_step'], d['ms_per_step_median'], d['value']))
PY
}
{ ab IDF_WGRAD_KR3=0; ab IDF_WGRAD_KR3=1; ab IDF_WGRAD_KR3=1 IDF_WGRAD_TPB3=24; ab IDF_WGRAD_KR3=1 IDF_WGRAD_TPB3=96; ab IDF_WGRAD_KR3=1 IDF_WGRAD_MINB3=8; ab IDF_WGRAD_KR3=0; ab IDF_WGRAD_KR3=1; } 2>&1 | tee $out/r03p_ab_kr3.txt
