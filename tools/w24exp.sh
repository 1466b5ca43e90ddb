for v in 0 16 32 64 128 240; do
  echo "== ablate $v"
  ACRMI_LIB=build_tools/libacrmi_w24a$v.so python tools/conv_bench.py --wino24 --filter 'b1 64->64 3x3 @64' --stamps 2>&1 | grep -v "^$" | grep -v amdgpu.ids
done
