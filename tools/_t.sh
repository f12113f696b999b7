#!/bin/bash
for i in 1 2; do
python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); s=d['stage_ms_per_step']
print('C3 ms/step', round(d['ms_per_step'],3), {k:round(s[k],3) for k in ('cond','cond_cellfinish')})"
done
python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_reference_answers.py -q -m gpu 2>&1 | tail -2
