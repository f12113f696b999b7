#!/bin/bash
for b in 64 32 16 8 4; do
  python bench.py --steps 128 --warmup 8 --no-cpu-baseline --reorder-every $b 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); s=d['stage_ms_per_step']
print('every', $b, 'ms/step', round(d['ms_per_step'],3), {k:round(s[k],3) for k in ('cond','cond_cellfinish','coal','move(adve+sedi+bcnd)','post_copy','reorder_storage','hskpng_vterm_all')})"
done
