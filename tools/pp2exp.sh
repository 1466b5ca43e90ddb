# conv_pp2: DMA-cursor entry in the chunk's first step (ACRMI_PP2_HOIST=1) vs behind the bridge barrier (0)
F='s2 64->64,s2 256->64,fuse 32->64,fuse 32->32,fuse 64->128,fuse 128->256,entry 32->512'
for t in pp2h0 pp2h1 pp2h0 pp2h1; do
  echo "== $t"
  ACRMI_LIB=build_tools/libacrmi_$t.so python tools/conv_bench.py --pp2 --filter "$F" --iters 50 2>&1 | grep -v "^$" | grep -v amdgpu.ids | cut -c1-200
done
ACRMI_LIB=build_tools/libacrmi_pp2h1.so python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "stride2 or polyphase or pp2" 2>&1 | tail -2
