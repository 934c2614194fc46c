"""Build libomp355.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m advancedliteratemachinery_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is written next to this file (git-ignored, but it
travels with the tree to the GPU box).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libomp355.so')
SOURCES = ['api.hip', 'gemm.hip', 'mlp.hip', 'norm.hip', 'swin_attn.hip', 'swin_block.hip', 'fpn.hip', 'decoder.hip', 'dec_rows.hip', 'dec_rows_x3.hip', 'kv_rows.hip', 'vit.hip', 'preprocess.hip']
HEADERS = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'omp355_debug.h'), os.path.join(CSRC, 'gemm256.inc'), os.path.join(CSRC, 'gemm4w.inc'), os.path.join(CSRC, 'gemm4wr.inc'), os.path.join(CSRC, 'gemm4wp.inc'), os.path.join(CSRC, 'rows_common.inc'), os.path.join(os.path.dirname(HERE), 'include', 'omp355.h')]
AUDITED = ('gemm.hip', 'dec_rows.hip', 'dec_rows_x3.hip', 'kv_rows.hip')   # their device assembly stays next to the object: the audits read it
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def _digest(paths, extra=''):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _fresh(target, digest):
    """An object is fresh when the stamp beside it holds the digest of (compiler flags, source, every header): CONTENT, not mtimes --
    a snapshot of the tree (gpurun, a checkout) does not keep mtimes, and `build()` then proves that the library on disk was compiled
    from the sources on disk."""
    try:
        with open(target + '.stamp') as f:
            return os.path.exists(target) and f.read().strip() == digest
    except OSError:
        return False


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    stamps = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        objs.append(o)
        flags = FLAGS + (['-save-temps=obj'] if src in AUDITED else [])
        dg = _digest([s] + HEADERS, ' '.join(flags))
        stamps.append(dg)
        if force or not _fresh(o, dg):
            jobs.append(([hipcc] + flags + ['-c', s, '-o', o], o, dg))

    def run(job):
        cmd, obj, dg = job if isinstance(job, tuple) else (job, None, None)
        if obj is not None and os.path.exists(obj + '.stamp'):
            os.remove(obj + '.stamp')
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stderr))
        if obj is not None:
            with open(obj + '.stamp', 'w') as f:
                f.write(dg)
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    # the register audits read the device assembly of THIS build and run before the link: a library whose asm-addressed kernels the
    # compiler broke (spills, accumulator-file traffic, copies of registers with loads in flight) must not be left on disk
    try:
        _audit()
    except Exception:
        if os.path.exists(LIB):
            os.remove(LIB)
        raise
    ldg = hashlib.sha256(' '.join(stamps).encode()).hexdigest()
    if force or jobs or not _fresh(LIB, ldg):
        run(([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs, LIB, ldg))
    return LIB


def _audit():
    """gemm_4w names all 256 accumulator registers in asm; gemm_4w_r / _p and the decoders' row-owner kernel (csrc/dec_rows.hip) load
    operand fragments with asm statements and count the waits by hand: refuse a build in which the compiler spilled, used the
    accumulator file itself or copied a register with a load in flight (advancedliteratemachinery_amd/audit.py; silent corruption otherwise)."""
    from . import audit
    asm = os.path.join(OBJ, 'gemm-hip-amdgcn-amd-amdhsa-gfx950.s')
    if os.path.exists(asm):   # objects from an older build tree have none: the next rebuild of gemm.hip writes it
        n, bad = audit.audit(asm)
        if bad or n == 0:
            raise RuntimeError('gemm_4w register audit failed (%d kernels):\n%s' % (n, '\n'.join(bad) or 'no gemm_4w kernel found in ' + asm))
    for stem in ('dec_rows', 'dec_rows_x3', 'kv_rows'):
        asm = os.path.join(OBJ, '%s-hip-amdgcn-amd-amdhsa-gfx950.s' % stem)
        if os.path.exists(asm):
            n, bad = audit.audit_dec_rows(asm)
            if bad or n == 0:
                raise RuntimeError('%s register audit failed (%d kernels):\n%s' % (stem, n, '\n'.join(bad) or 'no row-owner kernel found in ' + asm))


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
