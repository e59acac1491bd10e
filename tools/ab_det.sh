export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-roofline --no-sampling --no-large-batch --steps 40 --warmup 10"
for i in 1 2; do
for v in 0 1; do
  echo "IDF_DETERMINISTIC=$v: $(IDF_DETERMINISTIC=$v $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])')"
done
done
