#!/bin/bash
# usage: bash scripts/kernel_resources.sh <file.hip> [filter] [extra hipcc flags]  -- registers / scratch / occupancy per kernel (compile only)
F=$1; FILTER=${2:-.}; shift; shift
cd /root/repo/probaforms_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast "$@" -Rpass-analysis=kernel-resource-usage -c $F -o /tmp/kres.o 2>&1 | python3 -c "
import sys, re, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        if cur: rows.append(cur)
        cur = {'name': t.split(':',1)[1].strip()}
    elif cur:
        k, _, v = t.partition(':'); cur[k.strip()] = v.strip()
if cur: rows.append(cur)
for r in rows:
    n = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    n = re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '')).replace('void ', '').replace('rnvp::', '')
    print('%-64s VGPR %4s AGPR %4s SGPR %4s scratch %5s B/lane  spilled %3s  occ %2s  LDS %s' % (n[-64:], r.get('VGPRs','?'), r.get('AGPRs','?'), r.get('TotalSGPRs','?'), r.get('ScratchSize [bytes/lane]','?'), r.get('VGPRs Spill','?'), r.get('Occupancy [waves/SIMD]','?'), r.get('LDS Size [bytes/block]','?')))
" | grep -E "$FILTER"
