V=$PWD/infodiffusion_amd/variants/libinfodiff_hip_$1.so
B="python bench.py --no-cpu-baseline --no-roofline --no-large-batch --no-dp-probe --steps 3 --warmup 2"
for i in 1 2 3; do
  echo "$1:      $(IDF_LIB=$V $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["sampling"]["value"])')"
  echo "in-tree: $($B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["sampling"]["value"])')"
done
