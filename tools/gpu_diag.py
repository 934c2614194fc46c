"""Run every GPU parity check and print ALL records (does not stop at the first failure).
Usage (on the GPU box):  python tools/gpu_diag.py [--quick] > gpurun_out/diag.log"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import gpu_checks as C  # noqa: E402


def run(fn, *a):
    t = time.time()
    try:
        recs = fn(*a)
    except Exception:
        print('!! %s%s raised:\n%s' % (fn.__name__, a, traceback.format_exc()), flush=True)
        return [dict(name='%s%s' % (fn.__name__, a), err=float('inf'), tol=0, ok=False, note='exception')]
    torch.cuda.synchronize()
    for r in recs:
        print('%s %-70s err=%.3e tol=%.1e %s' % ('ok  ' if r['ok'] else 'FAIL', r['name'], r['err'], r['tol'], r['note']), flush=True)
    print('   (%s: %.1fs)' % (fn.__name__, time.time() - t), flush=True)
    return recs


def main():
    print(torch.cuda.get_device_name(0), torch.__version__, flush=True)
    allr = []
    for fn in C.ALL_OP_CHECKS:
        allr += run(fn)
    for dt in ('fp32', 'bf16'):
        for pre in (True, False):
            allr += run(C.check_decoder, dt, pre, True)
    allr += run(C.check_decoder, 'fp32', True, False)
    for name in ('spot_224', 'spot_odd', 'kie_sroie', 'postnorm_nofpn'):
        allr += run(C.check_e2e, name, 'fp32')
    for name in ('spot_224', 'spot_odd'):
        allr += run(C.check_e2e, name, 'bf16')
    for name in ('spot_odd', 'kie_sroie'):
        allr += run(C.check_e2e, name, 'fp32', True)
    allr += run(C.check_batch_equivalence, 'fp32')
    allr += run(C.check_batch_equivalence, 'bf16')
    allr += run(C.check_graph_matches_eager, 'fp32')
    allr += run(C.check_lanes, 'fp32')
    nbad = sum(1 for r in allr if not r['ok'])
    print('SUMMARY: %d checks, %d failed' % (len(allr), nbad))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'diag.json'), 'w') as f:
        json.dump(allr, f, indent=1)


if __name__ == '__main__':
    main()
