#!/bin/bash
out=gpurun_out
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "chain or dgrad_conv_with" > $out/r03b_kernels.log 2>&1; tail -15 $out/r03b_kernels.log
python -m pytest tests/test_gpu_model.py -m gpu -q -k "bf16_train_step or up_block or arena" > $out/r03b_model.log 2>&1; tail -30 $out/r03b_model.log
python tools/bf16_grad_profile.py > $out/r03b_gradprof.txt 2>&1; tail -60 $out/r03b_gradprof.txt
