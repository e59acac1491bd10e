#!/bin/bash
# the image-resident 8x8 ResBlock kernel beyond B = 64: DDIM-100 at B = 256 and the B = 128 train step, one box
B="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline"
run() { env "$@" python $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('DDIM-100 B256 %.1f img/s   train B128 %.0f img/s' % (d['sampling']['value'], d['large_batch']['value']), end='')"; echo "  $@"; }
for r in 1 2; do
run IDF_RB_SMALL_MAXB=64
run IDF_RB_SMALL_MAXB=128
run IDF_RB_SMALL_MAXB=256
done
