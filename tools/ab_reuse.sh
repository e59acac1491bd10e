export TMPDIR=/tmp
V=$PWD/infodiffusion_amd/variants/libinfodiff_hip_noreuse.so
for i in 1 2; do
  echo "no reuse: $(IDF_LIB=$V python tools/run_sampling.py 256 100 3 2>/dev/null | tail -1)"
  echo "reuse:    $(python tools/run_sampling.py 256 100 3 2>/dev/null | tail -1)"
done
