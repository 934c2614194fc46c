"""Build libomp355.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m advancedliteratemachinery_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is written next to this file (git-ignored, but it
travels with the tree to the GPU box).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libomp355.so')
SOURCES = ['api.hip', 'gemm.hip', 'mlp.hip', 'norm.hip', 'swin_attn.hip', 'swin_block.hip', 'fpn.hip', 'decoder.hip', 'vit.hip', 'preprocess.hip']
HEADERS = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'omp355_debug.h'), os.path.join(CSRC, 'gemm256.inc'), os.path.join(CSRC, 'gemm4w.inc'), os.path.join(CSRC, 'gemm4wr.inc'), os.path.join(CSRC, 'gemm4wp.inc'), os.path.join(os.path.dirname(HERE), 'include', 'omp355.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            # gemm.hip keeps its device assembly next to the object: the audit below reads it
            jobs.append([hipcc] + FLAGS + (['-save-temps=obj'] if src == 'gemm.hip' else []) + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stderr))
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    _audit_gemm4w()
    return LIB


def _audit_gemm4w():
    """gemm_4w names all 256 accumulator registers in asm: refuse a build in which the compiler spilled or used the accumulator file
    itself (tools/audit_gemm4w.py; silent corruption otherwise)."""
    asm = os.path.join(OBJ, 'gemm-hip-amdgcn-amd-amdhsa-gfx950.s')
    if not os.path.exists(asm):
        return   # objects from an older build tree: the next rebuild of gemm.hip writes it
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tools'))
    try:
        import audit_gemm4w
        n, bad = audit_gemm4w.audit(asm)
    finally:
        sys.path.pop(0)
    if bad or n == 0:
        raise RuntimeError('gemm_4w register audit failed (%d kernels):\n%s' % (n, '\n'.join(bad) or 'no gemm_4w kernel found in ' + asm))


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
