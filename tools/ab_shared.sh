export TMPDIR=/tmp
A="--no-cpu-baseline --no-roofline --no-sampling --no-large-batch --steps 40 --warmup 10"
run() { python - $A <<PY 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])'
import sys
sys.argv = ['bench.py'] + sys.argv[1:]
from infodiffusion_amd import ops
ops._RS_SHARED = bool($1)
import bench
bench.main()
PY
}
for i in 1 2; do
  echo "halo kernel for 128->128 @32x32 forward: $(run 0)"
  echo "shared-image form:                       $(run 1)"
done
