#!/bin/bash
set -o pipefail
O=gpurun_out/r03_run15; mkdir -p $O
echo "== gpu tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
echo "== ransac profile"; timeout -k 10 120 python tools/run_profile.py > $O/ransac_profile.txt 2>&1; head -16 $O/ransac_profile.txt | tail -12
echo "== soak settle"; timeout -k 10 600 python tools/soak_settle.py 300 7 > $O/soak_settle.txt 2>&1; tail -3 $O/soak_settle.txt
echo "== bench ransac bits"; timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_run15/bench.json').read().strip().split('\n')[-1])
print(d['ransac']['parity_path']); print({k:v for k,v in d['config4_panorama_8k'].items() if 'ransac' in k})
PY
