#!/usr/bin/env python3
"""Generates tools/coissue.hip: hand-placed instruction streams that settle what a gfx950 wave can issue beside its own
fp32 MFMAs (VERDICT r5 "Next round" item 1).

    python tools/coissue_gen.py            # writes tools/coissue.hip
    hipcc --offload-arch=gfx950 -O3 tools/coissue.hip -o /tmp/coissue && /tmp/coissue     # prints the table
    /opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading ...                                  # tools/coissue_isa.sh

Every variant is ONE `asm volatile` block (the compiler cannot re-order, pack or pad it): s_memtime, then a counted loop of
16 MFMAs on four rotating accumulator tiles with k independent filler instructions behind each MFMA, s_memtime again.
One wave per SIMD (256 threads, __launch_bounds__(256, 1)), one workgroup per CU on all 256 CUs.  Fillers rotate over 12
independent registers (8 for the LDS reads), so no filler waits for another filler's result.  Three matrix instructions:
the two fp32 ones the convolutions use (v_mfma_f32_32x32x2_f32: 64 cycles per SIMD; v_mfma_f32_16x16x4_f32: 32) and, as the
CONTROL that shows the harness sees hiding where the hardware provides it, v_mfma_f32_32x32x16_bf16 (32 cycles).
"""
import os

MFMA = {
    # name: (instruction, accumulator registers, operand kind, nominal cycles per SIMD)
    'f32_32x32x2': ('v_mfma_f32_32x32x2_f32', 16, 's', 64),
    'f32_16x16x4': ('v_mfma_f32_16x16x4_f32', 4, 's', 32),
    'bf16_32x32x16': ('v_mfma_f32_32x32x16_bf16', 16, 'q', 32),
}

# filler kind -> (format with {r} = rotating register index, number of rotating registers)
FILL = {
    'v_add_f32': ('v_add_f32 %[x{r}], %[x{r}], %[c]', 12),
    'v_fma_f32': ('v_fma_f32 %[x{r}], %[x{r}], %[c], %[c]', 12),
    'v_pk_add_f32': ('v_pk_add_f32 %[p{r}], %[p{r}], %[pc]', 12),
    'v_pk_fma_f32': ('v_pk_fma_f32 %[p{r}], %[p{r}], %[pc], %[pc]', 12),
    'v_accvgpr_read_b32': ('v_accvgpr_read_b32 %[x{r}], %[ag]', 12),
    'ds_read_b128': ('ds_read_b128 %[q{r}], %[la] offset:{off}', 8),
    's_nop_0': ('s_nop 0', 1),
    'ds_write_b128': ('ds_write_b128 %[la], %[q{r}] offset:{off}', 8),
    'buffer_load_dwordx4': ('buffer_load_dwordx4 %[q{r}], %[vo], %[rs], 0 offen', 8),
    # LDS-DMA piece: 64 lanes x 16 bytes straight into LDS at m0 (+ the instruction's offset)
    'lds_dma_dwordx4': ('s_mov_b32 m0, {m0}\\n"\n      "s_nop 0\\n"\n      "buffer_load_dwordx4 %[vo], %[rs], 0 offen lds', 8),
    # the same KiB through spare accumulator registers: a buffer load INTO AGPRs, a ds_write_b128 FROM AGPRs
    'buffer_load_to_agpr': ('buffer_load_dwordx4 %[aq{r}], %[vo], %[rs], 0 offen', 4),
    'ds_write_from_agpr': ('ds_write_b128 %[la], %[aq{r}] offset:{off}', 4),
}

KS = {
    'f32_32x32x2': [0, 1, 2, 4, 6, 8, 12, 16],
    'f32_16x16x4': [0, 1, 2, 4, 6, 8],
    'bf16_32x32x16': [0, 1, 2, 4, 5, 6, 8],
}
KINDS = {
    'f32_32x32x2': ['v_add_f32', 'v_fma_f32', 'v_pk_add_f32', 'v_pk_fma_f32', 'v_accvgpr_read_b32', 'ds_read_b128', 's_nop_0',
                    'ds_write_b128', 'buffer_load_dwordx4', 'lds_dma_dwordx4'],
    'f32_16x16x4': ['v_add_f32', 'v_pk_add_f32', 'ds_read_b128'],
    'bf16_32x32x16': ['v_add_f32', 'v_fma_f32', 'v_pk_add_f32', 'v_pk_fma_f32', 'ds_read_b128'],
}

# mixed gaps in the proportions of conv_wino24b's chunk step (36 packed VALU + 12 ds_read_b128 per 48 MFMAs = 3 + 1 per 4
# MFMAs) and the same arithmetic spelled with scalar VALU (6 + 1 per 4 MFMAs); per group of four MFMAs
MIXES = {
    'mix_pk_3pk_1ds_per4': [['v_pk_add_f32'], ['v_pk_fma_f32'], ['v_pk_add_f32'], ['ds_read_b128']],
    'mix_sc_6va_1ds_per4': [['v_add_f32', 'v_add_f32'], ['v_fma_f32', 'v_fma_f32'], ['v_add_f32', 'v_add_f32'], ['ds_read_b128']],
    'mix_sc_6va_1ds_front': [['v_add_f32', 'v_add_f32', 'v_fma_f32', 'v_fma_f32', 'v_add_f32', 'v_add_f32', 'ds_read_b128'], [], [], []],
    # burst law: the same 4 packed VALU per 4 MFMAs as one burst / two bursts / four bursts
    'burst_4pk_in_1': [['v_pk_add_f32'] * 4, [], [], []],
    'burst_4pk_in_2': [['v_pk_add_f32'] * 2, [], ['v_pk_add_f32'] * 2, []],
    'burst_4pk_in_4': [['v_pk_add_f32']] * 4,
    # one KiB per 4 MFMAs into LDS three ways: LDS-DMA; buffer load to VGPRs + ds_write; the same through spare AGPRs
    'kib_dma_per4': [['lds_dma_dwordx4'], [], [], []],
    'kib_vgpr_per4': [['buffer_load_dwordx4'], [], ['ds_write_b128'], []],
    'kib_agpr_per4': [['buffer_load_to_agpr'], [], ['ds_write_from_agpr'], []],
    'kib_dma_2per4': [['lds_dma_dwordx4'], [], ['lds_dma_dwordx4'], []],
    'kib_agpr_2per4': [['buffer_load_to_agpr', 'buffer_load_to_agpr'], [], ['ds_write_from_agpr', 'ds_write_from_agpr'], []],
    'gl_1per4': [['buffer_load_dwordx4'], [], [], []],
    'gl_2per4_together': [['buffer_load_dwordx4', 'buffer_load_dwordx4'], [], [], []],
    'gl_2per4_apart': [['buffer_load_dwordx4'], [], ['buffer_load_dwordx4'], []],
    'gl_1_then_4pk': [['buffer_load_dwordx4'] + ['v_pk_add_f32'] * 4, [], [], []],
}

GROUPS = 4          # 4 x 4 MFMAs per loop trip


class Rot:
    def __init__(self):
        self.n = {}

    def emit(self, kind):
        fmt, nreg = FILL[kind]
        i = self.n.get(kind, 0)
        self.n[kind] = i + 1
        return fmt.format(r=i % nreg, off=(i % 8) * 1024, m0=8192 + (i % 8) * 1024)


def body(mfma, gaps):
    """gaps: list of 4 lists of filler kinds (one list per MFMA of a group)"""
    ins = MFMA[mfma][0]
    rot = Rot()
    lines = []
    for g in range(GROUPS):
        for j in range(4):
            lines.append('%s %%[acc%d], %%[a], %%[b], %%[acc%d]' % (ins, j, j))
            for kind in gaps[j]:
                lines.append(rot.emit(kind))
    return lines


def kernel(name, mfma, gaps):
    ins, accregs, opk, _ = MFMA[mfma]
    acct = 'f32x16' if accregs == 16 else 'f32x4'
    abt = 'float' if opk == 's' else 'f32x4'
    lines = body(mfma, gaps)
    uses_lds = any('ds_' in l for l in lines)
    uses_vm = any('buffer_load' in l for l in lines)
    # (a ds_write FROM registers a buffer load is filling has to wait for that load: the loads of a trip are waited for at
    #  its end, the writes of the next trip then store what the previous trip fetched)
    asm = ['s_memtime %[t0]', 's_waitcnt lgkmcnt(0)', '1:'] + lines + (['s_waitcnt lgkmcnt(0)'] if uses_lds else []) + (
        ['s_waitcnt vmcnt(0)'] if uses_vm else []) + [
        's_sub_u32 %[n], %[n], 1', 's_cmp_lg_u32 %[n], 0', 's_cbranch_scc1 1b', 's_memtime %[t1]', 's_waitcnt lgkmcnt(0)']
    text = '\n'.join('      "%s\\n"' % l for l in asm)
    outs = ['[t0] "=&s"(t0)', '[t1] "=&s"(t1)', '[n] "+s"(n)'] + ['[acc%d] "+a"(acc%d)' % (i, i) for i in range(4)] \
        + ['[x%d] "+v"(x%d)' % (i, i) for i in range(12)] + ['[p%d] "+v"(p%d)' % (i, i) for i in range(12)] \
        + ['[q%d] "+v"(q%d)' % (i, i) for i in range(8)] + ['[aq%d] "+a"(aq%d)' % (i, i) for i in range(4)]
    ins_ = ['[a] "v"(a)', '[b] "v"(b)', '[c] "v"(c)', '[pc] "v"(pc)', '[ag] "a"(ag)', '[la] "v"(la)', '[vo] "v"(vo)', '[rs] "s"(rs)']
    decl = []
    decl.append('  %s acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};' % acct)
    decl.append('  %s a = mk<%s>(seed + threadIdx.x), b = mk<%s>(2.f);' % (abt, abt, abt))
    decl.append('  float ' + ', '.join('x%d = seed + %d' % (i, i) for i in range(12)) + ';')
    decl.append('  f32x2 ' + ', '.join('p%d = {seed, seed + %d}' % (i, i) for i in range(12)) + ';')
    decl.append('  f32x4 ' + ', '.join('q%d = {}' % i for i in range(8)) + ';')
    decl.append('  f32x4 ' + ', '.join('aq%d = {}' % i for i in range(4)) + ';')
    sink = ' + '.join(['sum(acc%d)' % i for i in range(4)] + ['x%d' % i for i in range(12)] + ['p%d[0] + p%d[1]' % (i, i) for i in range(12)]
                      + ['q%d[0]' % i for i in range(8)] + ['aq%d[0]' % i for i in range(4)])
    return '''
__global__ __launch_bounds__(256, 1) void %s(unsigned long long* ticks, float* out, int iters, float seed, const float* src) {
  __shared__ f32x4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = f32x4{seed, seed, seed, seed};
  __syncthreads();
%s
  float c = seed, ag = seed * 3.f;
  f32x2 pc = {seed, seed};
  unsigned la = (threadIdx.x & 63) * 16;
  unsigned vo = threadIdx.x * 16;                         // 4 KiB per workgroup of a 64 KiB L2-resident buffer
  const unsigned long long sp = (unsigned long long)src;
  const i32x4 rs = {__builtin_amdgcn_readfirstlane((int)(unsigned)sp), __builtin_amdgcn_readfirstlane((int)((sp >> 32) & 0xffff)),
                    __builtin_amdgcn_readfirstlane(65536), __builtin_amdgcn_readfirstlane(0x00020000)};
  unsigned long long t0, t1;
  int n = __builtin_amdgcn_readfirstlane(iters);
  asm volatile(
%s
      : %s
      : %s
      : "memory", "scc");
  out[blockIdx.x * 256 + threadIdx.x] = %s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) ticks[threadIdx.x >> 6] = t1 - t0;
}
''' % (name, '\n'.join(decl), text, ', '.join(outs), ', '.join(ins_), sink), sum(len(g) for g in gaps) * GROUPS


def main():
    kernels = []
    rows = []          # (kernel name, mfma, label, fillers per MFMA)
    for mfma in MFMA:
        for kind in KINDS[mfma]:
            for k in KS[mfma]:
                if k == 0 and kind != KINDS[mfma][0]:
                    continue
                if kind in ('ds_write_b128', 'buffer_load_dwordx4', 'lds_dma_dwordx4') and k > 4:
                    continue
                name = 'k_%s__%s__%d' % (mfma, kind, k)
                src, nfill = kernel(name, mfma, [[kind] * k] * 4)
                kernels.append(src)
                rows.append((name, mfma, kind if k else 'none', k))
        for mix, gaps in MIXES.items():
            if mfma != 'f32_32x32x2' and not mix.startswith('mix_'):
                continue
            name = 'k_%s__%s' % (mfma, mix)
            src, nfill = kernel(name, mfma, gaps)
            kernels.append(src)
            rows.append((name, mfma, mix, nfill / 16.0))
    head = '''// GENERATED by tools/coissue_gen.py - do not edit.  What a gfx950 wave issues beside its own MFMAs: cycles per MFMA
// (s_memtime) of hand-placed streams, one wave per SIMD.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ T mk(float v);
template <> __device__ float mk<float>(float v) { return v; }
template <> __device__ f32x4 mk<f32x4>(float v) { return f32x4{v, v, v, v}; }
__device__ float sum(f32x16 v) { float s = 0; for (int i = 0; i < 16; ++i) s += v[i]; return s; }
__device__ float sum(f32x4 v) { return v[0] + v[1] + v[2] + v[3]; }
'''
    table = ',\n'.join('  {(void*)%s, "%s", "%s", %g, %d}' % (n, m, lab, k, MFMA[m][3]) for n, m, lab, k in rows)
    tail = '''
struct Row { void* fn; const char* mfma; const char* filler; double per_mfma; int nominal; };
static const Row rows[] = {
%s
};

int main(int argc, char** argv) {
  const int iters = 400;                  // x 16 MFMAs per trip
  unsigned long long* ticks; float* out; float* src;
  (void)hipMalloc(&ticks, 64); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&src, 65536); (void)hipMemset(src, 0, 65536);
  printf("# cycles per MFMA (s_memtime, mean of the 4 waves of workgroup 0; 256 workgroups x 256 threads = one wave per SIMD on every CU)\\n");
  printf("# delta = cycles per MFMA above the bare stream of the same MFMA; per_filler = delta / fillers per MFMA\\n");
  printf("%%-16s %%-24s %%8s %%10s %%8s %%10s\\n", "mfma", "filler", "per_mfma", "cyc/mfma", "delta", "per_filler");
  double bare = 0; const char* cur = "";
  for (const Row& r : rows) {
    float seed = 1.f; int it = iters;
    void* args[] = {&ticks, &out, &it, &seed, &src};
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      if (hipLaunchKernel(r.fn, dim3(256), dim3(256), args, 0, 0) != hipSuccess) { printf("launch failed\\n"); return 1; }
      if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\\n"); return 1; }
      unsigned long long h[4];
      (void)hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
      double m = (double)(h[0] + h[1] + h[2] + h[3]) / 4.0 / (16.0 * iters);
      if (m < best) best = m;
    }
    if (strcmp(cur, r.mfma) != 0) { cur = r.mfma; bare = best; }
    double d = best - bare;
    printf("%%-16s %%-24s %%8.2f %%10.2f %%8.2f %%10.2f\\n", r.mfma, r.filler, r.per_mfma, best, d, r.per_mfma > 0 ? d / r.per_mfma : 0.0);
  }
  return 0;
}
''' % table
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'coissue.hip')
    with open(path, 'w') as f:
        f.write(head + ''.join(kernels) + tail)
    print(path, len(rows), 'kernels')


if __name__ == '__main__':
    main()
