for v in 0 1 2 4 7; do
  echo "== wino3 ablate $v"
  ACRMI_LIB=build_tools/libacrmi_w3a$v.so python tools/conv_bench.py --wino3 --filter 'b0 32->32 3x3 @128 no' --stamps 2>&1 | grep -v "^$" | grep -v amdgpu.ids | cut -c1-330
done
