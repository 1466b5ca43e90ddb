# conv_x3: 2 x 2 wave layout (W22, default) vs two rows x both n-tiles per wave (cfg 852)
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "split" 2>&1 | tail -2
F='b1 64->64 3x3 @64,b2 128,l1 64->64,towers 8x64'
for c in -1 852 -1 852; do
  echo "== cfg $c"
  python tools/conv_bench.py --x3 --cfg $c --filter "$F" --iters 50 2>&1 | grep -v "^$" | grep -v amdgpu.ids | cut -c1-200
done
