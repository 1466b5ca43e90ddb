# world-size-1 probes of the C-ABI transport's cost next to the step loop
B="python bench.py --no-cpu-baseline --no-reduced-precision --no-latency --no-point-heads --no-pmc --steps 30"
p() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
$B 2>/dev/null | p "no process group"
ACRMI_FORCE_DIST=1 ACRMI_GATHER=torch $B 2>/dev/null | p "torch transport"
ACRMI_FORCE_DIST=1 ACRMI_GATHER=c $B 2>/dev/null | p "c transport, communicator before the pool"
ACRMI_FORCE_DIST=1 ACRMI_GATHER=c ACRMI_COMM_LATE=1 $B 2>/dev/null | p "c transport, communicator after the pool"
ACRMI_FORCE_DIST=1 ACRMI_GATHER=c GPU_MAX_HW_QUEUES=16 $B 2>/dev/null | p "c transport, 16 hardware queues"
ACRMI_FORCE_DIST=1 ACRMI_GATHER=c $B --pipeline 1 2>/dev/null | p "c transport, one context"
ACRMI_FORCE_DIST=1 ACRMI_GATHER=torch $B --pipeline 1 2>/dev/null | p "torch transport, one context"
