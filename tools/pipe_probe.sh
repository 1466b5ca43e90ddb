# contexts taking batches in turn: 1 .. 4 (runtime-default hardware queues), and lanes per context
B="python bench.py --no-reduced-precision --no-pmc --no-cpu-baseline --no-point-heads --no-latency --steps 20"
p() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for n in 1 2 3 4; do $B --pipeline $n 2>/dev/null | p "contexts $n"; done
$B --pipeline 2 --lanes 2 2>/dev/null | p "contexts 2, lanes 2"
$B --pipeline 1 --lanes 2 2>/dev/null | p "contexts 1, lanes 2"
$B --pipeline 1 --lanes 4 2>/dev/null | p "contexts 1, lanes 4"
