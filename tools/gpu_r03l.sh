#!/bin/bash
out=gpurun_out
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "chain or conv_fwd or prologue or dgrad_conv_with or epilogue_statistics or two_source" 2>&1 | tail -2
python tools/bench_chain.py 2>&1 | grep apply | head -10
python tools/bench_small_conv.py 2>&1 | grep -v amdgpu | head -8
bash tools/ab_libs.sh pfd1 default 2>&1 | tee $out/r03l_ab_pfd.txt
