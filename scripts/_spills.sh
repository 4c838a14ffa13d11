#!/bin/bash
# vgpr / spill / scratch of every search_hist2_kernel instantiation of the working tree
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I /root/repo/include -S --cuda-device-only /root/repo/fba_pomdp_amd/csrc/fba_search.hip -o /tmp/search.s 2>/dev/null
python3 - <<'PY'
import re
s=open('/tmp/search.s').read()
for m in re.finditer(r'\.name:\s+(\S*search_hist2\S*)\n', s):
    pass
# metadata blocks
for blk in s.split('- .agpr_count:')[1:]:
    name=re.search(r'\.name:\s+(\S+)', blk).group(1)
    if 'search_hist2' not in name: continue
    g=lambda k: re.search(r'\.'+k+r':\s+(\d+)', blk).group(1)
    print(name[8:40], 'vgpr', g('vgpr_count'), 'spill', g('vgpr_spill_count'), 'scratch', g('private_segment_fixed_size'))
PY
