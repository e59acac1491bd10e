#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/<tag>_*; copy what should be judged into profiles/.
# usage: tools/collect_profiles.sh <tag>     (e.g. r02k)
tag=$1
export TMPDIR=/tmp
out=gpurun_out
rocprofv3 --kernel-trace --stats -d $out/${tag}_p1 -o train -- python3 bench.py --no-cpu-baseline --no-large-batch --no-dp-probe > $out/${tag}_p1.log 2>&1
db=$(find $out/${tag}_p1 -name '*results.db' | head -1)
python tools/step_inventory.py $db 80 > $out/${tag}_step_inventory.txt
python tools/rocpd_stats.py $db $out/${tag}_kernel_stats.csv
grep "\"metric\"" $out/${tag}_p1.log | tail -1 > $out/${tag}_bench_under_rocprof.json
rocprofv3 --kernel-trace -d $out/${tag}_p2 -o samp -- python3 tools/run_sampling.py 256 100 2 > $out/${tag}_p2.log 2>&1
db2=$(find $out/${tag}_p2 -name '*results.db' | head -1)
python tools/eval_inventory.py $db2 45 > $out/${tag}_sampling_eval_inventory_b256.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${tag}_$c -- python3 bench.py --graph 0 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-sampling --no-large-batch --no-dp-probe > $out/${tag}_$c.log 2>&1
done
python tools/pmc_traffic.py $out/${tag}_FETCH_SIZE $out/${tag}_WRITE_SIZE $out/${tag}_pmc_traffic.json
# MFMA-pipe counters of the same eager steps (their own pass: --pmc with --kernel-trace only)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $out/${tag}_MFMA -- python3 bench.py --graph 0 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-sampling --no-large-batch --no-dp-probe > $out/${tag}_MFMA.log 2>&1
python tools/pmc_mfma.py $out/${tag}_MFMA $out/${tag}_pmc_mfma.json > $out/${tag}_pmc_mfma.txt
# wave-state / vector-issue counters (their own pass): is a kernel vector-bound?
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $out/${tag}_VALU -- python3 bench.py --graph 0 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-sampling --no-large-batch --no-dp-probe > $out/${tag}_VALU.log 2>&1
python tools/pmc_valu.py $out/${tag}_VALU $out/${tag}_pmc_valu.json > $out/${tag}_pmc_valu.txt
rm -rf $out/${tag}_FETCH_SIZE $out/${tag}_WRITE_SIZE $out/${tag}_MFMA $out/${tag}_VALU
rm -rf $out/${tag}_p1 $out/${tag}_p2      # the databases are large; the summaries above are what is kept
ls -la $out | grep ${tag}_
