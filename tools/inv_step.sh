export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r05e_p1 -o train -- python3 bench.py --no-cpu-baseline --no-large-batch --no-sampling --no-roofline > gpurun_out/r05e_p1.log 2>&1
db=$(find gpurun_out/r05e_p1 -name '*results.db' | head -1)
python tools/step_inventory.py $db 90 > gpurun_out/r05e_step_inventory.txt
rm -rf gpurun_out/r05e_p1
head -60 gpurun_out/r05e_step_inventory.txt
