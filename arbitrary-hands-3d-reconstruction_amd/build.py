"""Build libacrmi.so (gfx950) in-tree with hipcc.  `python -m` usable; __graft_entry__.build() calls build()."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libacrmi.so')
SOURCES = ['acrmi.hip', 'acrmi_program.hip', 'acrmi_ops.hip', 'acrmi_comm.hip', 'conv_mfma.hip', 'conv_h16.hip', 'elementwise.hip', 'heads.hip', 'mano.hip', 'stem.hip', 'stem7.hip', 'pair1x1.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (ROCm toolchain required to build libacrmi.so)')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'acrmi.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, experiments=False):
    """experiments: -DACRMI_EXPERIMENTS - the library then reads the environment switches of timing experiments / A-B runs
    (csrc/kernels.h experiment_env: ACRMI_ABLATE_LANE_SYNC, ACRMI_EVENT_FLAGS, ACRMI_PLAN_WAIT_US, ACRMI_XCD_SWIZZLE,
    ACRMI_CONV_DMA, ACRMI_WINO24B).  The production build ignores them."""
    if not force and not experiments and not needs_build():
        return LIB
    hipcc = _hipcc()
    flags = FLAGS + (['-DACRMI_EXPERIMENTS'] if experiments else [])
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        cmd = [hipcc] + flags + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, out.decode(errors='replace')))
        if verbose and out.strip():
            print(out.decode(errors='replace'))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, experiments='--experiments' in sys.argv)
    print(LIB)
