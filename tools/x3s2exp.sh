timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "stride2_split" 2>&1 | tail -2
F='fuse 32->64,fuse 32->32,entry 32->512'
echo "== x3s2 (Cin 32: one n-tile per item)"; timeout 300 python tools/conv_bench.py --x3 --filter "$F" --iters 30 --stamps 2>&1 | grep -v "^$" | grep -v amdgpu.ids | cut -c1-250
echo "== x3s2 cfg 851 (two n-tiles)"; timeout 300 python tools/conv_bench.py --x3 --cfg 851 --filter "$F" --iters 30 2>&1 | grep -v "^$" | grep -v amdgpu.ids | cut -c1-200
