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


def _asm_of(src):
    return os.path.join(OBJ, '%s-hip-amdgcn-amd-amdhsa-gfx950.s' % src.replace('.hip', ''))


def _fresh(target, digest, stamp=None):
    """An object is fresh when the stamp beside it holds the digest of (compiler flags, source, every header): CONTENT, not mtimes --
    a snapshot of the tree (gpurun, a checkout) does not keep mtimes, and `build()` then proves that the library on disk was compiled
    from the sources on disk."""
    try:
        with open(stamp or target + '.stamp') as f:
            return os.path.exists(target) and f.read().strip() == digest
    except OSError:
        return False


def plan(force=False):
    """-> (jobs, objects, stamps, audited): what build() would compile right now.  jobs: (command, object, digest) per STALE source -- its stamp does not hold
    the digest of (flags, source, headers), or it is an AUDITED source with neither an audit stamp of that digest nor its device assembly on disk."""
    hipcc = _hipcc()
    jobs, objs, stamps, audited = [], [], [], {}
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        objs.append(o)
        flags = FLAGS + (['-save-temps=obj'] if src in AUDITED else [])
        dg = _digest([s] + HEADERS, ' '.join(flags))
        stamps.append(dg)
        # an AUDITED object must be provably audited: either its audit stamp holds this digest (the assembly of exactly this object passed -- the
        # stamp travels to the GPU box, the 40 MB of compiler temporaries do not) or its device assembly is here to be read; otherwise it is stale
        if src in AUDITED:
            audited[src] = dg
        if force or not _fresh(o, dg) or (src in AUDITED and not _fresh(o, dg, o + '.audit.stamp') and not os.path.exists(_asm_of(src))):
            jobs.append(([hipcc] + flags + ['-c', s, '-o', o], o, dg))
    return jobs, objs, stamps, audited


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs, objs, stamps, audited = plan(force)

    def run(job):
        cmd, obj, dg = job[:3]
        stamp = job[3] if len(job) > 3 else obj + '.stamp'
        for st in (stamp, obj + '.audit.stamp'):
            if os.path.exists(st):
                os.remove(st)
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stderr))
        with open(stamp, 'w') as f:
            f.write(dg)
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    # the register audits read the device assembly of THIS build and run before the link: a library whose asm-addressed kernels the
    # compiler broke (spills, accumulator-file traffic, copies of registers with loads in flight) must not be left on disk
    try:
        _audit(audited)
    except Exception:
        if os.path.exists(LIB):
            os.remove(LIB)
        raise
    ldg = hashlib.sha256(' '.join(stamps).encode()).hexdigest()
    # the library's stamp lives with the objects (csrc/build/, git-ignored): a checkout cannot deliver a stamp for a library it does not hold
    lstamp = os.path.join(OBJ, 'libomp355.so.stamp')
    if force or jobs or not _fresh(LIB, ldg, lstamp):
        run(([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs, LIB, ldg, lstamp))
    return LIB


def _audit(audited):
    """gemm_4w names all 256 accumulator registers in asm; gemm_4w_r / _p and the decoders' row-owner kernel (csrc/dec_rows.hip) load
    operand fragments with asm statements and count the waits by hand: refuse a build in which the compiler spilled, used the
    accumulator file itself or copied a register with a load in flight (advancedliteratemachinery_amd/audit.py; silent corruption otherwise)."""
    from . import audit
    for src, dg in audited.items():
        asm, stem = _asm_of(src), src.replace('.hip', '')
        astamp = os.path.join(OBJ, stem + '.o.audit.stamp')
        if _fresh(os.path.join(OBJ, stem + '.o'), dg, astamp):
            continue   # the assembly of exactly this object passed before
        if not os.path.exists(asm):   # build() recompiles an audited source with neither stamp nor assembly: cannot happen
            raise RuntimeError('%s register audit: %s is missing' % (stem, asm))
        n, bad = audit.audit(asm) if src == 'gemm.hip' else audit.audit_dec_rows(asm)
        if bad or n == 0:
            raise RuntimeError('%s register audit failed (%d kernels):\n%s' % (stem, n, '\n'.join(bad) or 'no audited kernel found in ' + asm))
        with open(astamp, 'w') as f:
            f.write(dg)


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
