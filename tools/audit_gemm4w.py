"""Build audit of gemm_4w (csrc/gemm4w.inc): the kernel addresses all 256 accumulator registers literally from asm statements, so the
compiler must neither spill nor touch the accumulator file itself (cdna_hip_programming.md 5.7 item 4).  Reads the device assembly hipcc
leaves next to the object (advancedliteratemachinery_amd/build.py compiles gemm.hip with -save-temps=obj) and demands, for EVERY gemm_4w
instantiation: no scratch, no spill, and no v_accvgpr_* / scratch_* instruction outside an ;;#ASMSTART ... ;;#ASMEND block.
    python tools/audit_gemm4w.py [path/to/gemm-hip-amdgcn-amd-amdhsa-gfx950.s]   -> exit status 1 on a violation"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, 'advancedliteratemachinery_amd', 'csrc', 'build', 'gemm-hip-amdgcn-amd-amdhsa-gfx950.s')


def audit(path=DEFAULT):
    """-> (number of gemm_4w kernels seen, list of violation strings)"""
    text = open(path).read()
    bad, seen = [], 0
    # kernel bodies: from the label to s_endpgm
    for m in re.finditer(r'^(_ZN\S*gemm_4w\S*):[^\n]*\n(.*?)\ts_endpgm', text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        seen += 1
        inasm = False
        for ln in body.split('\n'):
            if 'ASMSTART' in ln:
                inasm = True
            elif 'ASMEND' in ln:
                inasm = False
            elif not inasm and ('v_accvgpr' in ln or 'scratch_' in ln):
                bad.append('%s: compiler-generated `%s`' % (name, ln.strip()))
                break
    for m in re.finditer(r'\.name:\s+(_ZN\S*gemm_4w\S*)\n(.*?)\.wavefront_size', text, re.S):
        name, meta = m.group(1), m.group(2)
        # (scalar-register spills are allowed: hipcc parks them in lanes of a vector register it owns, not in scratch or the accumulator file)
        for key in ('.private_segment_fixed_size', '.vgpr_spill_count'):
            v = re.search(re.escape(key) + r':\s+(\d+)', meta)
            if v and int(v.group(1)) != 0:
                bad.append('%s: %s = %s' % (name, key, v.group(1)))
        a = re.search(r'\.agpr_count:\s+(\d+)', meta)
        if a and int(a.group(1)) != 256:
            bad.append('%s: .agpr_count = %s (the kernel names a0..a255)' % (name, a.group(1)))
    return seen, bad


if __name__ == '__main__':
    n, bad = audit(sys.argv[1] if len(sys.argv) > 1 else DEFAULT)
    print('gemm_4w audit: %d kernels, %d violations' % (n, len(bad)))
    for b in bad:
        print('  ' + b)
    sys.exit(1 if bad or n == 0 else 0)
