# same-box A/B of a module-level switch of infodiffusion_amd.ops in the bench's training step: tools/ab_attr.sh <attribute> <value A> <value B>
export TMPDIR=/tmp
A="--no-cpu-baseline --no-roofline --no-sampling --no-large-batch --steps 40 --warmup 10"
run() { python - $A <<PY 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])'
import sys
sys.argv = ['bench.py'] + sys.argv[1:]
from infodiffusion_amd import ops
ops.$1 = $2
import bench
bench.main()
PY
}
for i in 1 2 3; do
  echo "$1 = $2: $(run $1 $2)"
  echo "$1 = $3: $(run $1 $3)"
done
