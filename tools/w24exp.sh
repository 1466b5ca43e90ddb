echo "== wino24c prefetch distance 2 (cfg 842)"
ACRMI_LIB=build_tools/libacrmi_w24cp.so python tools/conv_bench.py --wino24 --cfg 842 --filter 'b1 64->64 3x3 @64,b2 128' --stamps 2>&1 | grep -v "^$" | grep -v amdgpu.ids | cut -c1-300
ACRMI_LIB=build_tools/libacrmi_w24cp.so python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "all_positions" 2>&1 | tail -2
