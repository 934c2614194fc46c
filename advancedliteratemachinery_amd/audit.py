"""Build audit of gemm_4w (csrc/gemm4w.inc): the kernel addresses all 256 accumulator registers literally from asm statements, so the
compiler must neither spill nor touch the accumulator file itself (cdna_hip_programming.md 5.7 item 4).  Reads the device assembly hipcc
leaves next to the object (advancedliteratemachinery_amd/build.py compiles gemm.hip with -save-temps=obj) and demands, for EVERY gemm_4w
instantiation: no scratch, no spill, and no v_accvgpr_* / scratch_* instruction outside an ;;#ASMSTART ... ;;#ASMEND block; for gemm_4w_r /
gemm_4w_p (asm fragment loads, hand-counted waits) also no register copy inside the stage loop.
    python -m advancedliteratemachinery_amd.audit [path/to/gemm-hip-amdgcn-amd-amdhsa-gfx950.s]   -> exit status 1 on a violation
Part of the package (round 5; it lived under tools/): build.py runs it on the assembly BEFORE it links, so a library whose asm-addressed
kernels the compiler broke is never left on disk."""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT = os.path.join(HERE, 'csrc', 'build', 'gemm-hip-amdgcn-amd-amdhsa-gfx950.s')


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
    # gemm_4w_r / gemm_4w_p load fragments with asm statements and count the waits by hand: the compiler believes a fragment register
    # holds its value as soon as the statement is issued, so a register-to-register copy inside the stage loop (phi resolution,
    # rematerialisation) could read it before the data arrives.  Demand that the innermost loop (the stages) holds no v_mov at all.
    for m in re.finditer(r'^(_ZN\S*gemm_4w_[rp]\S*):[^\n]*\n(.*?)\ts_endpgm', text, re.S | re.M):
        name, lines = m.group(1), m.group(2).split('\n')
        if not re.search(r'ELi0E(Lb[01]E)?EEvNS', name):
            continue   # ablation / trace instantiations (wrong results by construction or development only)
        # innermost loops: the header block carries "Inner Loop Header", its other blocks "in Loop: Header=BBx_y Depth=d" (LLVM's comments)
        inner = set()
        for i, ln in enumerate(lines):
            if 'Inner Loop Header' in ln:
                for j in range(i, max(-1, i - 4), -1):
                    mm = re.match(r'^\.L(BB\d+_\d+):', lines[j])
                    if mm:
                        inner.add(mm.group(1))
                        break
        if not inner:
            bad.append('%s: no inner loop found (the stage loop)' % name)
            continue
        in_loop, n_mfma = False, 0
        for i, ln in enumerate(lines):
            if re.match(r'^(\.LBB\d+_\d+:|; %bb\.\d+:)', ln):
                ctx = ' '.join(lines[i:i + 3])
                mm = re.match(r'^\.L(BB\d+_\d+):', ln)
                in_loop = ('Inner Loop Header' in ctx and mm is not None and mm.group(1) in inner) or any('Header=%s ' % h in ctx for h in inner)
            elif in_loop:
                if 'v_mfma' in ln:
                    n_mfma += 1
                if re.search(r'\bv_mov_b(32|64)(_e32|_e64)?\s+v\S*,\s*v', ln):   # vector-register source
                    bad.append('%s: register copy inside the stage loop: `%s`' % (name, ln.strip()))
                    break
        if n_mfma < 512:
            bad.append('%s: the stage loop was not recognised (%d MFMAs inside the innermost loop, 512 expected)' % (name, n_mfma))
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


def _regs(text):
    out = set()
    for r in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', text):
        if r.group(3) is not None:
            out.add(int(r.group(3)))
        else:
            out.update(range(int(r.group(1)), int(r.group(2)) + 1))
    return out


def audit_dec_rows(path):
    """csrc/dec_rows.hip (round 5): every wave of the decoders' row-owner kernels walks a linear stream of weight fragments that it loads with
    asm statements into a register ring (`global_load_dwordx4 v[a:b], ... ; ring-load`) and waits for with hand-counted waits
    (`s_waitcnt vmcnt(7) ; ring-take v[a:b]`).  The compiler believes such a register holds its value from the moment the load is issued; any
    compiler-generated instruction that touches it between the load and its take (a copy for phi resolution, a spill to scratch, a
    rematerialisation) would read or destroy data in flight.  Walking the kernel text in layout order with the set of PENDING registers
    (loaded, not yet taken), the audit demands: no instruction outside the ;;#ASMSTART ... ;;#ASMEND blocks names a pending register; every
    take names registers that are pending; a load never overwrites a pending register.  (Loop bodies are laid out once, and the ring is
    periodic over every loop in these kernels, so layout order sees every load / take pair.)  Ordinary spills elsewhere are allowed.
    -> (number of dec_rows kernels seen, list of violation strings)"""
    text = open(path).read()
    bad, seen = [], 0
    # the whole function body, up to its .Lfunc_end label: a kernel may hold several s_endpgm (round 6: the blocks of an XCD outside a launch's
    # mask retire before the first ring load; the compiler lays that exit out wherever it likes)
    for m in re.finditer(r'^(_ZN\S*_rows_\S*kernel\S*):[^\n]*\n(.*?)^\.Lfunc_end', text, re.S | re.M):
        name, lines = m.group(1), m.group(2).split('\n')
        seen += 1
        pending, landed = set(), set()
        n_load = n_take = 0
        inasm = False
        err = None
        for ln in lines:
            if 'ASMSTART' in ln:
                inasm = True
                continue
            if 'ASMEND' in ln:
                inasm = False
                continue
            code = ln.split(';')[0]
            if inasm:
                if 'ring-load' in ln:
                    mm = re.match(r'\s*global_load_dwordx4\s+(v\[\d+:\d+\])', ln)
                    dst = _regs(mm.group(1)) if mm else set()
                    if not dst:
                        err = 'unparsed ring load `%s`' % ln.strip()
                    elif dst & pending:
                        err = 'ring load overwrites a fragment still in flight: `%s`' % ln.strip()
                    pending |= dst
                    landed -= dst
                    n_load += 1
                elif 'ring-drain' in ln:
                    landed |= pending   # everything requested has arrived; the registers stay reserved (operands of the drain) until taken
                    pending = set()
                elif 'ring-take' in ln:
                    regs = _regs(ln.split('ring-take')[1])
                    if not regs or not regs <= (pending | landed):
                        err = 'take of registers that no ring load filled: `%s`' % ln.strip()
                    pending -= regs
                    landed -= regs
                    n_take += 1
            elif pending and re.match(r'\s+[a-z]', ln) and (_regs(code) & pending):
                err = 'compiler-generated instruction touches a fragment in flight: `%s`' % ln.strip()
            if err:
                bad.append('%s: %s' % (name, err))
                break
        if not err and pending:
            bad.append('%s: the kernel ends with fragments in flight (no ring-drain): the registers they land in are not reserved' % name)
        if not err and (n_load < 8 or n_take < 8):
            bad.append('%s: ring not recognised (%d loads, %d takes)' % (name, n_load, n_take))
    return seen, bad


if __name__ == '__main__':
    path = sys.argv[1] if len(sys.argv) > 1 else DEFAULT
    n, bad = audit_dec_rows(path) if 'dec_rows' in os.path.basename(path) else audit(path)
    print('register audit of %s: %d kernels, %d violations' % (os.path.basename(path), n, len(bad)))
    for b in bad:
        print('  ' + b)
    sys.exit(1 if bad or n == 0 else 0)
