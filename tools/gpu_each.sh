#!/bin/bash
# every launch of the kernels matching PATTERN in the last replayed training step: tools/gpu_each.sh TAG PATTERN [VAR=val ...]
tag=$1; pat=$2; shift; shift
export TMPDIR=/tmp
out=gpurun_out
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats -d $out/${tag}_p1 -o train -- python3 bench.py --no-cpu-baseline --no-large-batch --no-sampling --no-roofline > $out/${tag}_p1.log 2>&1
db=$(find $out/${tag}_p1 -name '*results.db' | head -1)
python3 - "$db" "$pat" <<'PY' | tee $out/${tag}_each.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
gx = 'grid_x' if 'grid_x' in cols else ('grid_size' if 'grid_size' in cols else '0')
wx = 'workgroup_x' if 'workgroup_x' in cols else ('workgroup_size' if 'workgroup_size' in cols else '1')
rows = db.execute('select name, start, end, %s, %s from kernels order by start' % (gx, wx)).fetchall()
print('columns:', cols)
marks = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[0]]
seg = rows[marks[-2] + 1:marks[-1] + 1]
print('launches %d  span %.2f ms' % (len(seg), (seg[-1][2] - seg[0][1]) / 1e6))
for i, (n, s, e, gx, wx) in enumerate(seg):
    if sys.argv[2] in n:
        print('%4d  %7.1f us  blocks %6d  %s' % (i, (e - s) / 1e3, gx // max(wx, 1), n[:60]))
PY
python tools/step_inventory.py $db 90 > $out/${tag}_step_inventory.txt
rm -rf $out/${tag}_p1
