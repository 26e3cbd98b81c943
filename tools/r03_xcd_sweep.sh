#!/bin/bash
run() { python3 bench.py --no-cpu --no-gba --no-extra 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']; print('$1', d['ms_per_step'], d['value'], k['k_fast_cells']['ms_per_step'], k['k_orient_desc']['ms_per_step'])
"; }
run base
for v in 8 32 64 256; do CCM_ORB_XCD_FC=$v run FC=$v; done
for v in 0 8 32 64 256; do CCM_ORB_XCD_OD=$v run OD=$v; done
run base
