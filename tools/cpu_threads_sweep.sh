#!/bin/bash
# CPU oracle train-step time vs thread count on the GPU box's host (bench.py's cpu_baseline worker): tools/cpu_threads_sweep.sh [threads ...]   (default 8 16 32 64 128)
for t in ${@:-8 16 32 64 128}; do
  f=/tmp/_cpu_$t.txt; rm -f $f
  IDF_CPU_THREADS=$t OMP_NUM_THREADS=$t CUDA_VISIBLE_DEVICES= HIP_VISIBLE_DEVICES= timeout 200 python bench.py --cpu-baseline-worker $f --a_dim 32 > /dev/null 2>&1
  echo "threads $t: $(grep train $f | tr '\n' ' ')  $(grep eval $f | tr '\n' ' ')"
done
