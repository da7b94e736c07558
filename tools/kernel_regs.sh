#!/bin/bash
# usage: tools/kernel_regs.sh file.hip [grep-pattern]   -- registers / occupancy per kernel (gfx950)
cd "$(dirname "$0")/../multimodal-dmm_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c "$1" -o /tmp/_kr.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c "
import re, sys
name = None; row = {}
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        name = m.group(1); row = {}; continue
    m = re.search(r'remark: [^ ]* +(\w[\w ]*?)(?: \[.*?\])?: (\S+)', line)
    if m and name:
        row[m.group(1).strip()] = m.group(2)
        if m.group(1).startswith('LDS Size'):
            print('%-90s v=%s a=%s occ=%s scratch=%s' % (name[:90], row.get('VGPRs'), row.get('AGPRs'), row.get('Occupancy'), row.get('ScratchSize')))
" | grep -E "${2:-.}"
