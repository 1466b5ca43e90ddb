# batch-1 / batch-8 latency A/B of the lane planner (experiments build: ACRMI_PLAN_RANK / ACRMI_PLAN_WAIT_US are read)
for cfg in "ACRMI_PLAN_RANK=0" "ACRMI_PLAN_RANK=1" "ACRMI_PLAN_RANK=1 ACRMI_PLAN_WAIT_US=30" "ACRMI_PLAN_RANK=1 ACRMI_PLAN_WAIT_US=8" "ACRMI_PLAN_RANK=0 ACRMI_PLAN_WAIT_US=30"; do
  echo "== $cfg"
  env $cfg ACRMI_LIB=build_tools/libacrmi_planx.so python bench.py --no-cpu-baseline --no-reduced-precision --no-point-heads --no-pmc --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['latency'])"
done
