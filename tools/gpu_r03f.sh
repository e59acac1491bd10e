#!/bin/bash
out=gpurun_out
python tools/bench_chain.py > $out/r03f_chain_micro.txt 2>&1; grep -v amdgpu.ids $out/r03f_chain_micro.txt
python tools/op_sources.py > $out/r03f_opsrc.txt 2>&1; grep -v amdgpu.ids $out/r03f_opsrc.txt | head -50
python -m pytest tests/test_gpu_model.py -m gpu -q -k "bf16_train_step" 2>&1 | tail -3
