export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-roofline --no-sampling --no-large-batch --steps 40 --warmup 10"
for i in 1 2; do
for v in 0 1 2; do
  echo "SYNC=$v: $(IDF_CONV_RS_SYNC=$v $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"], d.get("deterministic_value",{}).get("ms_per_step"))')"
done
done
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "bf16_train_step_celeba or deterministic or one_rank_rccl" 2>&1 | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" | tail -5
