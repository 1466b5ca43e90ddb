echo "== wino24b variant h2"
ACRMI_LIB=build_tools/libacrmi_w24h2.so python tools/conv_bench.py --wino24 --filter 'b1 64->64 3x3 @64 no,b2 128' --stamps 2>&1 | grep -v "^$" | grep -v amdgpu.ids
echo "== wino3 (production lib)"
python tools/conv_bench.py --wino3 --filter 'b0 32->32' --stamps 2>&1 | grep -v "^$" | grep -v amdgpu.ids
echo "== pp2 (production lib)"
python tools/conv_bench.py --pp2 --filter 'fuse 32->64 3x3s2,fuse 32->32 3x3s2,fuse 64->128,s2 64->64' --stamps 2>&1 | grep -v "^$" | grep -v amdgpu.ids
echo "== wino2 segm (production lib)"
python tools/conv_bench.py --wino2 --filter 'segm 16->64,segm 64->33,segm 33->33' --stamps 2>&1 | grep -v "^$" | grep -v amdgpu.ids
