#!/bin/bash
# round 3, call A: new parity tests, rounding study, baseline bench on this box
out=gpurun_out
python -m pytest tests -m gpu -x -q > $out/r03a_pytest.log 2>&1; echo "pytest rc $?" >> $out/r03a_pytest.log
tail -5 $out/r03a_pytest.log
python tools/bf16_rounding_study.py celeba > $out/r03a_rounding_celeba.txt 2>&1
python tools/bf16_rounding_study.py fmnist > $out/r03a_rounding_fmnist.txt 2>&1
cat $out/r03a_rounding_celeba.txt
python bench.py --no-cpu-baseline > $out/r03a_bench.json 2> $out/r03a_bench.err
tail -c 3000 $out/r03a_bench.json
