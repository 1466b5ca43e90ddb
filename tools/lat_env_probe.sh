B="python bench.py --no-reduced-precision --no-pmc --steps 10"
p() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('latency',{}).get('batch1_ms'), d.get('latency',{}).get('batch8_ms'), d.get('latency',{}).get('batch1_lanes'))"; }
$B 2>/dev/null | p "unset, full legs"
$B --no-cpu-baseline 2>/dev/null | p "unset, no cpu baseline"
$B --no-cpu-baseline --no-point-heads 2>/dev/null | p "unset, no cpu baseline, no point heads"
GPU_MAX_HW_QUEUES=4 $B 2>/dev/null | p "Q=4, full legs"
GPU_MAX_HW_QUEUES=4 $B --no-cpu-baseline --no-point-heads 2>/dev/null | p "Q=4, no cpu baseline, no point heads"
GPU_MAX_HW_QUEUES=8 $B --no-cpu-baseline --no-point-heads 2>/dev/null | p "Q=8, no cpu baseline, no point heads"
