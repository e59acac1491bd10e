B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-large-batch --no-dp-probe --no-sampling"
run() { echo "$(env "$@" $B 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_median"])')   $@"; }
for r in 1 2; do
run X=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=2
run HSA_ENABLE_SDMA=0
run AMD_DIRECT_DISPATCH=0
run HSA_ENABLE_INTERRUPT=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
done
