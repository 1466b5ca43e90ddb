# conv_x3 cycle stamps of workgroup 0 (per chunk: start, MFMAs issued; per item: epilogue done)
F='b1 64->64 3x3 @64,b2 128,l1 64->64'
for t in 0 1 4; do
  echo "== ablate $t"
  ACRMI_LIB=build_tools/libacrmi_x3a$t.so python tools/conv_bench.py --x3 --filter "$F" --iters 20 --stamps 2>&1 | grep -v "^$" | grep -v amdgpu.ids | cut -c1-400
done
