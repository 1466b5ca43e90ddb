#!/bin/bash
# headline at different numbers of contexts in flight (engine.EnginePool): bash tools/pipe_probe.sh
for p in 2 3 2 3 4; do
  timeout 300 python bench.py --no-cpu-baseline --no-point-heads --no-latency --no-pmc --no-reduced-precision --pipeline $p --steps 30 --warmup 5 2>/dev/null > /tmp/pp.json
  python - "$p" <<'PY'
import json, sys
d = json.loads(open('/tmp/pp.json').read().strip().split('\n')[-1])
print('pipeline', sys.argv[1], d['value'], d['ms_per_step'])
PY
done
