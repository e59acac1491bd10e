#!/bin/bash
out=gpurun_out
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "prologue_conv_one_launch" 2>&1 | tail -2
python -m pytest tests/test_gpu_model.py -m gpu -q -k "sampling_b256 or samplers_real" 2>&1 | tail -2
run() { echo "== $*"; env "$@" python tools/run_sampling.py 256 100 2 2>&1 | grep "img/s" | tail -1; }
{ run IDF_DLDS_EVAL=0; run IDF_DLDS_EVAL=1; run IDF_LIB=$PWD/infodiffusion_amd/variants/libinfodiff_hip_eg4.so; run IDF_DLDS_EVAL=0; run IDF_DLDS_EVAL=1; } 2>&1 | tee $out/r03j_sampling_eval.txt
IDF_LIB=$PWD/infodiffusion_amd/variants/libinfodiff_hip_stamp.so IDF_CONV_PS=0 IDF_GN_FUSE_FORCE=1 python tools/dlds_stamps.py 256 64 64 64 1 2>&1 | grep -v amdgpu
