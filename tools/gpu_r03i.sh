#!/bin/bash
out=gpurun_out
run() { echo "== $*"; env "$@" python tools/run_sampling.py 256 100 2 2>&1 | grep "img/s" | tail -1; }
{ run X=0; run IDF_GN_FUSE_FORCE=1; run IDF_CONV_PS_PRO=1; run IDF_CONV_PS_PRO=1 IDF_GN_FUSE_FORCE=1; run IDF_CONV_DLDS=0; run IDF_CONV_PS=0; run X=0; run IDF_GN_FUSE_FORCE=1; } 2>&1 | tee $out/r03i_sampling_knobs.txt
