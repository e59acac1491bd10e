#!/bin/bash
out=gpurun_out
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "chain" 2>&1 | tail -3
python tools/bench_chain.py > $out/r03g_chain_micro.txt 2>&1; grep -v amdgpu.ids $out/r03g_chain_micro.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-large-batch --no-sampling 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('ms/step %.3f median %.3f' % (d['ms_per_step'], d['ms_per_step_median']))"
