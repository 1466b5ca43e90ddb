echo "== wino24c staggered stores"
ACRMI_LIB=build_tools/libacrmi_w24cs.so python tools/conv_bench.py --wino24 --filter 'b1 64->64 3x3 @64,b2 128,l1 64->64 3x3 @128,towers' --stamps 2>&1 | grep -v "^$" | grep -v amdgpu.ids | cut -c1-300
echo "== conv_wino24b (--cfg 841)"
ACRMI_LIB=build_tools/libacrmi_w24cs.so python tools/conv_bench.py --wino24 --cfg 841 --filter 'b1 64->64 3x3 @64,b2 128,l1 64->64 3x3 @128,towers' 2>&1 | grep -v "^$" | grep -v amdgpu.ids
