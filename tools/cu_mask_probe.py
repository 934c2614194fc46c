"""CU-partitioned streams: (1) which XCDs / CUs a mask selects, (2) cross-attention bandwidth and GEMM rate on CU subsets,
(3) an HBM-bound stream (point-decoder cross-attention, 256 images per launch) next to a matrix-core-bound stream
(stage-2 Swin GEMMs) -- sequential vs concurrent unmasked vs concurrent on disjoint CU sets.
    python tools/cu_mask_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from advancedliteratemachinery_amd import _lib, ops  # noqa: E402

DEV = 'cuda:0'


def where(words, tag):
    st = ops.masked_stream(words) if words is not None else torch.cuda.Stream()
    with torch.cuda.stream(st):
        w = ops.where_probe(4096)
    torch.cuda.synchronize()
    w = w.cpu()
    xcc = (w[:, 0] & 15).tolist()
    hw = w[:, 1].tolist()
    per = {}
    for x, h in zip(xcc, hw):
        per.setdefault(x, set()).add(((h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15))   # (SE, SH, CU) of HW_ID
    print('%-28s XCCs %s ; distinct (se, sh, cu) per XCC: %s ; total %d' % (tag, sorted(per), [len(per[k]) for k in sorted(per)], sum(len(v) for v in per.values())), flush=True)
    return st


def cross_setup(B=256):
    nH, M, d = 8, 4096, 512
    q = torch.randn(B, d, device=DEV).to(torch.bfloat16)
    K = torch.randn(B, nH, M, 64, device=DEV).to(torch.bfloat16)
    Vt = torch.randn(B, nH, M // 32, 64, 32, device=DEV).to(torch.bfloat16)
    groups = torch.tensor([(b, 1, b) for b in range(B)], dtype=torch.int32, device=DEV)
    out = torch.empty(B, d, device=DEV, dtype=torch.bfloat16)
    by = B * 2 * M * d * 2

    def fn():
        ops.dec_cross_attn_step(q, K, Vt, nH * M * 64, M, None, groups, B, 1, None, out, M, nH, 1)
    return fn, by


def gemm_setup(Mrows=131072, N=1536, Kd=512):
    A = torch.randn(Mrows, Kd, device=DEV).to(torch.bfloat16)
    W = (torch.randn(N, Kd, device=DEV) / Kd ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    out = torch.empty(Mrows, N, device=DEV, dtype=torch.bfloat16)

    def fn():
        ops.gemm(A, W, bias, out=out)
    return fn, 2.0 * Mrows * N * Kd


def timed(st, fn, iters):
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    _lib.lib()
    print(torch.cuda.get_device_name(0), flush=True)
    where(None, 'no mask')
    where([0xffffffff, 0, 0, 0, 0, 0, 0, 0], 'bits 0..31')
    where([0xff] * 8, 'bits (i mod 32) < 8')
    where([0x01010101] * 8, 'bits (i mod 8) == 0')
    cross, cby = cross_setup()
    gemm, gfl = gemm_setup()
    streams = {}
    for n in (8, 16, 24, 32):
        streams[n] = ops.masked_stream(ops.cu_mask_words(n)) if n < 32 else torch.cuda.Stream()
        us = timed(streams[n], cross, 20)
        print('cross-attention 256 images on %3d CUs/XCD : %7.1f us  %5.0f GB/s' % (n, us, cby / us / 1e3), flush=True)
        us = timed(streams[n], gemm, 20)
        print('GEMM 131072x1536x512 on %3d CUs/XCD       : %7.1f us  %5.0f TF/s' % (n, us, gfl / us / 1e6), flush=True)
    # concurrent: NG GEMMs on one stream, NC cross-attention launches on another
    NG, NC = 60, 40

    def both(sg, sc, tag):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(sg):
            for _ in range(NG):
                gemm()
        with torch.cuda.stream(sc):
            for _ in range(NC):
                cross()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('%-44s : %7.2f ms for %d GEMMs + %d cross-attention launches' % (tag, dt * 1e3, NG, NC), flush=True)
        return dt
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    both(s0, s0, 'sequential (one stream)')
    both(s0, s0, 'sequential (one stream)')
    both(s0, s1, 'concurrent, no masks')
    both(s0, s1, 'concurrent, no masks')
    for nd in (8, 16):
        sg = ops.masked_stream(ops.cu_mask_words(nd, complement=True))
        sc = ops.masked_stream(ops.cu_mask_words(nd))
        both(sg, sc, 'concurrent, cross on %d CUs/XCD, GEMM on %d' % (nd, 32 - nd))
        both(sg, sc, 'concurrent, cross on %d CUs/XCD, GEMM on %d' % (nd, 32 - nd))
        both(s0, sc, 'concurrent, cross on %d CUs/XCD, GEMM unmasked' % nd)


if __name__ == '__main__':
    main()
