# same-box A/B of a variant library against the in-tree one: tools/ab_lib.sh <variant name> [train|sampling]
export TMPDIR=/tmp
V=$PWD/infodiffusion_amd/variants/libinfodiff_hip_$1.so
B="python bench.py --no-cpu-baseline --no-roofline --no-sampling --no-large-batch --no-dp-probe --steps 40 --warmup 10"
for i in 1 2 3; do
  echo "$1:      $(IDF_LIB=$V $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])')"
  echo "in-tree: $($B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])')"
done
