#!/bin/bash
# the one-launch attention block by batch threshold: DDIM-100 at B = 256 and the B = 128 / B = 32 train steps, one box
B="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
run() { env "$@" python $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('train B32 %.3f ms   DDIM-100 B256 %.1f img/s   train B128 %.0f img/s' % (d['ms_per_step'], d['sampling']['value'], d['large_batch']['value']), end='')"; echo "  $@"; }
for r in 1 2; do
run IDF_ATTN_BLOCK=0
run IDF_ATTN_BLOCK_MINB=256
run IDF_ATTN_BLOCK_MINB=128
done
