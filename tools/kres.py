"""Compile one HIP source for gfx950 and print a compact per-kernel resource table
(VGPRs, spills, occupancy) from -Rpass-analysis=kernel-resource-usage."""
import re
import subprocess
import sys

src = sys.argv[1]
extra = sys.argv[2:]
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17',
       '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/tmp/kres.o'] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
for line in out.splitlines():
    if ' error' in line or 'error:' in line:
        print(line)
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}
        rows.append(cur)
        continue
    for key in ('VGPRs', 'AGPRs', 'VGPRs Spill', 'SGPRs Spill', 'Occupancy [waves/SIMD]', 'ScratchSize [bytes/lane]', 'SGPRs'):
        m = re.search(r'remark:\s+' + re.escape(key) + r': (\d+)', line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(anonymous namespace\)::', '', name)[:90]
    print('%-92s v%-4d a%-3d sp%-3d scr%-4d occ%d sg%d' % (name, r.get('VGPRs', -1), r.get('AGPRs', 0), r.get('VGPRs Spill', 0),
                                                        r.get('ScratchSize [bytes/lane]', 0), r.get('Occupancy [waves/SIMD]', 0), r.get('SGPRs', 0)))
