# GPU_MAX_HW_QUEUES 4 (the runtime default; what the package leaves since round 6) vs 8 (what rounds 2-5 set on import) on every single-GPU scenario of bench.py
B="python bench.py --no-cpu-baseline --no-reduced-precision --no-point-heads --no-pmc --steps 20"
p() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('latency',{}).get('batch1_ms'), d.get('latency',{}).get('batch8_ms'))"; }
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q $B 2>/dev/null | p "queues $q: no process group"
  GPU_MAX_HW_QUEUES=$q ACRMI_FORCE_DIST=1 ACRMI_GATHER=torch $B --no-latency 2>/dev/null | p "queues $q: torch transport"
  GPU_MAX_HW_QUEUES=$q ACRMI_FORCE_DIST=1 ACRMI_GATHER=c $B --no-latency 2>/dev/null | p "queues $q: c transport"
  GPU_MAX_HW_QUEUES=$q ACRMI_FORCE_DIST=1 ACRMI_GATHER=c $B --no-latency --pipeline 1 2>/dev/null | p "queues $q: c transport, one context"
done
