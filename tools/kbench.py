"""Per-kernel micro-benchmarks on the GPU box (development aid; not part of the product or the tests).

  python tools/kbench.py cross|gemm|dec_gemm|selfattn|all

Times each libomp355 entry point in isolation with HIP events over many back-to-back launches on one
stream (so launch gaps are included, as they are in a captured decoder step) and prints achieved
GB/s / TFLOP/s next to the algorithmic bytes / flops."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from advancedliteratemachinery_amd import _lib, ops  # noqa: E402

DEV = 'cuda'
GEMM_VARIANTS = [int(v) for v in os.environ.get('KBENCH_GEMM_VARIANTS', '5,7,8').split(',')]


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def bench_cross(dtype=torch.bfloat16):
    B, M, nH, d = 8, 4096, 8, 512
    KB = 32 if dtype == torch.bfloat16 else 16
    esz = 2 if dtype == torch.bfloat16 else 4
    Mpad = (M + KB - 1) // KB * KB
    # several layer slabs so that consecutive launches do not hit the same (L2 / MALL resident) memory
    NS = 6
    K = torch.randn(NS, B, nH, Mpad, 64, device=DEV).to(dtype)
    V = torch.randn(NS, B, nH, Mpad // KB, 64, KB, device=DEV).to(dtype)
    for rows_per_img in (1, 64):
        R = B * rows_per_img
        qt = 1 if rows_per_img <= 16 else (2 if rows_per_img <= 32 else 4)
        groups = []
        for b in range(B):
            for o in range(0, rows_per_img, 16 * qt):
                groups.append((b * rows_per_img + o, min(16 * qt, rows_per_img - o), b))
        g = torch.tensor(groups, dtype=torch.int32, device=DEV)
        q = torch.randn(R, d, device=DEV).to(dtype)
        out = torch.empty(R, d, device=DEV, dtype=dtype)
        alg = B * 2 * M * d * esz + 2 * R * d * esz
        for S, ring in [(S, r) for r in ((0, 2, 1) if rows_per_img > 32 else (0,)) for S in (1, 2, 4, 8, 16)]:   # ring: q4 mode (0 = register streaming, 2 = LDS ring one block per step, 1 = 64-key chunks)
            ops.cross_q4(ring)
            partial = torch.empty(R, nH, S, 68, device=DEV)
            state = [0]

            def fn():
                i = state[0] % NS
                state[0] += 1
                ops.dec_cross_attn_step(q, K[i], V[i], nH * Mpad * 64, Mpad, None, g, len(groups), qt, partial, out, M, nH, S)
            us = timeit(fn, iters=60)
            print('cross[%s] rows/img=%-3d qt=%d ring=%d S=%-2d groups=%d : %7.1f us  %6.0f GB/s (%.2f of 8 TB/s)'
                  % (str(dtype)[6:], rows_per_img, qt, ring, S, len(groups), us, alg / us / 1e3, alg / us / 1e3 / 8000), flush=True)
    ops.cross_q4(1)


def bench_cross128(dtype=torch.bfloat16):
    """the two cross-attention kernels at the bench's engine-call size (KBENCH_CROSS_IMAGES images, default 128, M = 4096)"""
    I, M, nH, d, KB = int(os.environ.get('KBENCH_CROSS_IMAGES', '128')), 4096, 8, 512, 32
    from advancedliteratemachinery_amd.model.transformer import Decoder
    g = torch.Generator(device='cpu').manual_seed(0)
    K = torch.randn(2, I, nH, M, 64, generator=g).to(DEV, dtype)
    Vt = torch.randn(2, I, nH, M // KB, 64, KB, generator=g).to(DEV, dtype)
    h = _lib.lib()
    for rows_per_img, splits in ((1, (1, 2, 4)), (64, (1, 2))):
        counts = [rows_per_img] * I
        groups, qt = Decoder.make_tiles(counts)
        gd = torch.tensor(groups, dtype=torch.int32, device=DEV)
        R = sum(counts)
        q = torch.randn(R, d, device=DEV).to(dtype)
        out = torch.empty(R, d, device=DEV, dtype=dtype)
        alg = I * 2 * M * d * 2 + 2 * R * d * 2
        for S in splits:
            partial = torch.empty(R, nH, S, 68, device=DEV)
            for nt in ((0, 1) if rows_per_img == 1 else (2, 4, 1, 5, 6)):   # 64 rows: nt column = q4 mode (2 = block per step, 4 = 64-key chunks, 1 = + non-temporal DMA)
                if rows_per_img == 1:
                    h.omp_debug_cross_nt(nt)
                else:
                    h.omp_debug_cross_q4(nt)
                st = [0]

                def fn():
                    st[0] += 1
                    ops.dec_cross_attn_step(q, K[st[0] & 1], Vt[st[0] & 1], nH * M * 64, M, None, gd, len(groups), qt, partial, out, M, nH, S)
                us = timeit(fn, iters=30, warm=4)
                print('cross128 rows/img=%-2d S=%d nt=%d : %7.1f us  %6.0f GB/s (%.2f of 8 TB/s)' % (rows_per_img, S, nt, us, alg / us / 1e3, alg / us / 1e3 / 8000), flush=True)
    h.omp_debug_cross_nt(0)
    h.omp_debug_cross_q4(1)


def bench_cross_split():
    """fp32-grade cross-attention at the bench's engine-call size (KBENCH_CROSS_IMAGES images, default 160, M = 4096): the fp32 slabs
    (16-key blocks, fp32 matrix cores) vs the split-plane slabs (32-key blocks of [hi | lo] bf16, three bf16 products), same bytes."""
    I, M, nH, d = int(os.environ.get('KBENCH_CROSS_IMAGES', '160')), 4096, 8, 512
    from advancedliteratemachinery_amd.model.transformer import Decoder
    g = torch.Generator(device='cpu').manual_seed(0)
    h = _lib.lib()
    slabs = {
        'f32': (torch.randn(2, I, nH, M, 64, generator=g).to(DEV), torch.randn(2, I, nH, M // 16, 64, 16, generator=g).to(DEV), nH * M * 64),
        'split': (torch.randn(2, I, nH, M // 32, 2, 32, 64, generator=g).to(DEV, torch.bfloat16),
                  torch.randn(2, I, nH, M // 32, 2, 64, 32, generator=g).to(DEV, torch.bfloat16), nH * M * 128),
    }
    for rows_per_img, splits in ((1, (1, 2)), (64, (1, 2))):
        counts = [rows_per_img] * I
        groups, qt = Decoder.make_tiles(counts)
        gd = torch.tensor(groups, dtype=torch.int32, device=DEV)
        R = sum(counts)
        q = torch.randn(R, d, device=DEV)
        out = torch.empty(R, d, device=DEV)
        alg = I * 2 * M * d * 4 + 2 * R * d * 4
        for name, (K, Vt, stride) in slabs.items():
            for S in splits:
                partial = torch.empty(R, nH, S, 68, device=DEV)
                for ring in ((1, 4, 5, 6) if (rows_per_img == 64 and name == 'split') else (1,)):
                    h.omp_debug_cross_q4(ring)
                    st = [0]

                    def fn():
                        st[0] += 1
                        ops.dec_cross_attn_step(q, K[st[0] & 1], Vt[st[0] & 1], stride, M, None, gd, len(groups), qt, partial, out, M, nH, S)
                    us = timeit(fn, iters=30, warm=4)
                    print('cross_fp32grade[%-5s] images=%d rows/img=%-2d S=%d ring=%d : %7.1f us  %6.0f GB/s (%.2f of 8 TB/s)'
                          % (name, I, rows_per_img, S, ring, us, alg / us / 1e3, alg / us / 1e3 / 8000), flush=True)
    h.omp_debug_cross_q4(1)


def bench_gemm(dtype=torch.bfloat16):
    # (M, N, K, act, residual) of the Swin-B stages at B=8 1024x1024 and the K/V projection
    shapes = [(524288, 384, 128, 0, 0), (524288, 128, 128, 0, 1), (524288, 512, 128, 1, 0), (524288, 128, 512, 0, 1),
              (131072, 768, 256, 0, 0), (131072, 256, 256, 0, 1), (131072, 1024, 256, 1, 0), (131072, 256, 1024, 0, 1),
              (32768, 1536, 512, 0, 0), (32768, 512, 512, 0, 1), (32768, 2048, 512, 1, 0), (32768, 512, 2048, 0, 1),
              (8192, 3072, 1024, 0, 0), (8192, 1024, 1024, 0, 1), (8192, 4096, 1024, 1, 0), (8192, 1024, 4096, 0, 1),
              (32768, 6144, 512, 0, 0)]
    ms = int(os.environ.get('KBENCH_GEMM_MSCALE', '1'))   # 4 = the encoder's 32-image chunks
    f32res = os.environ.get('KBENCH_GEMM_F32RES', '1') == '1'   # the engines keep the residual stream in fp32
    lib = os.environ.get('KBENCH_GEMM_LIB', '0') == '1'         # calibration: torch's library GEMM (hipBLASLt) on the same product
    x3 = os.environ.get('KBENCH_GEMM_X3', '0') == '1'           # the parity engine's products: split-pair A, [hi | hi | lo] weight image, K' = 3 K
    only = os.environ.get('KBENCH_GEMM_ONLY')                   # e.g. "8,9,10,11": indices into the shape list
    if only:
        shapes = [shapes[int(i)] for i in only.split(',')]
    if os.environ.get('KBENCH_GEMM_SHAPES'):   # "M,N,K,act,res;..." instead of the Swin list
        shapes = [tuple(int(v) for v in sh.split(',')) for sh in os.environ['KBENCH_GEMM_SHAPES'].split(';')]
    for (M, N, K, act, res) in shapes:
        M = M * ms
        if x3:
            A = ops.split_bf16(torch.randn(M, K, device=DEV))
            W = ops.split_weight3(torch.randn(N, K, device=DEV) / K ** 0.5)
        else:
            A = torch.randn(M, K, device=DEV).to(dtype)
            W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(dtype)
        pad = int(os.environ.get('KBENCH_GEMM_PAD', '0'))   # elements added to the row strides of A and W (L2 channel experiment)
        if pad:
            Ap = torch.zeros(M, A.shape[1] + pad, device=DEV, dtype=A.dtype); Ap[:, :A.shape[1]] = A; A = Ap[:, :A.shape[1]]
            Wp = torch.zeros(N, W.shape[1] + pad, device=DEV, dtype=W.dtype); Wp[:, :W.shape[1]] = W; W = Wp[:, :W.shape[1]]
        bias = torch.randn(N, device=DEV)
        odt = torch.float32 if ((res and f32res) or (x3 and not act)) else dtype
        split = x3 and act
        out = torch.empty(M, 2 * N if split else N, device=DEV, dtype=odt)
        r = torch.randn(M, N, device=DEV).to(odt) if res else None
        kw = dict(a_wrap=2 * K, out_dtype=(ops.SPLIT if split else odt)) if x3 else {}
        fl = 2.0 * M * N * K * (3 if x3 else 1)
        by = (M * K + N * K) * 2 * (2 if x3 else 1) + M * N * (2 if res else 1) * out.element_size() * (2 if split else 1)
        for which in GEMM_VARIANTS:
            ops.force_gemm_kernel(which)
            try:
                us = timeit(lambda: ops.gemm(A, W, bias, residual=r, act=act, out=out, **kw), iters=20, warm=3)
            except Exception as e:   # a selector that does not take this shape
                print('gemm[k%d] %dx%dx%d : %s' % (which, M, N, K, str(e)[:100]), flush=True)
                continue
            print('gemm[%s,k%d] %7dx%5dx%5d act=%d res=%d out=%s : %8.1f us  %6.1f TF/s  %6.0f GB/s' % ('x3' if x3 else str(dtype)[6:], which, M, N, K * (3 if x3 else 1), act, res, 'split' if split else str(odt)[6:], us, fl / us / 1e6, by / us / 1e3),
                  flush=True)
        ops.force_gemm_kernel(0)
        if lib:
            o2 = torch.empty(M, N, device=DEV, dtype=dtype)
            Wt = W.t()
            us = timeit(lambda: torch.mm(A, Wt, out=o2), iters=20, warm=3)
            print('gemm[lib     ] %7dx%5dx%5d plain bf16 product (no bias / act / residual) : %8.1f us  %6.1f TF/s' % (M, N, K, us, fl / us / 1e6), flush=True)


def bench_kvproj(dtype=torch.bfloat16):
    """K / V^T projection of the encoder memory into the blocked slabs (12 (decoder, layer) slabs x 512 features)."""
    B, tok, K, nH, NL = 64, 4096, 512, 8, 12
    mem = torch.randn(B * tok, K, device=DEV).to(dtype)
    W = (torch.randn(NL * 512, K, device=DEV) / K ** 0.5).to(dtype)
    bias = torch.randn(NL * 512, device=DEV)
    Kd = torch.zeros(NL, B, nH, tok, 64, dtype=dtype, device=DEV)
    Vd = torch.zeros(NL, B, nH, tok // 32, 64, 32, dtype=dtype, device=DEV)
    geom = (B, tok, tok, nH, 32)
    fl = 2.0 * B * tok * NL * 512 * K
    for which in GEMM_VARIANTS:
        ops.force_gemm_kernel(which)
        us = timeit(lambda: ops.gemm(mem, W, bias, out=Kd, store_mode=_lib.STORE_KBLK, kv=geom), iters=5, warm=2)
        print('kvproj[k%d] K slabs   %d images : %8.1f us  %6.1f TF/s' % (which, B, us, fl / us / 1e6), flush=True)
        us = timeit(lambda: ops.gemm(W, mem, bias, out=Vd, store_mode=_lib.STORE_VBLK, kv=geom, bias_along_m=True, M=NL * 512, N=B * tok, K=K), iters=5, warm=2)
        print('kvproj[k%d] V^T slabs %d images : %8.1f us  %6.1f TF/s' % (which, B, us, fl / us / 1e6), flush=True)
    ops.force_gemm_kernel(0)


def bench_mlp(dtype=torch.bfloat16):
    """Swin MLP per stage at B=8 1024x1024: LayerNorm + fc1(GELU) + fc2(+residual) as three launches vs omp_swin_mlp_fused."""
    from advancedliteratemachinery_amd.model.packing import pack_mlp
    for (M, C) in ((524288, 128), (131072, 256), (32768, 512)):
        Hd = 4 * C
        x = torch.randn(M, C, device=DEV).to(dtype)
        g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        w1 = (torch.randn(Hd, C, device=DEV) / C ** 0.5).to(dtype)
        w2 = (torch.randn(C, Hd, device=DEV) / Hd ** 0.5).to(dtype)
        b1, b2 = torch.randn(Hd, device=DEV) * 0.1, torch.randn(C, device=DEV) * 0.1
        pack = pack_mlp(w1, b1, w2)
        y = torch.empty_like(x)
        hbuf = torch.empty(M, Hd, device=DEV, dtype=dtype)
        out = torch.empty_like(x)
        fl = 2.0 * M * C * Hd * 2

        def unfused():
            ops.layernorm(x, g, b, out=y)
            ops.gemm(y, w1, b1, act=ops.ACT_GELU, out=hbuf)
            ops.gemm(hbuf, w2, b2, residual=x, out=out)
        us = timeit(unfused, iters=10, warm=2)
        print('mlp[C=%d] M=%d unfused (3 launches) : %8.1f us  %6.1f TF/s' % (C, M, us, fl / us / 1e6), flush=True)
        for v in range(4 if C == 128 else (3 if C == 256 else 2)):
            ops.swin_mlp_variant(v)
            us = timeit(lambda: ops.swin_mlp_fused(x, g, b, pack, b2, out=out), iters=10, warm=2)
            print('mlp[C=%d] M=%d fused v%d            : %8.1f us  %6.1f TF/s  %6.0f GB/s (x in + y out)'
                  % (C, M, v, us, fl / us / 1e6, 2 * M * C * 2 / us / 1e3), flush=True)
        ops.swin_mlp_variant(0)


def bench_swin_block(dtype=torch.bfloat16):
    """attention half of a stage-0 block (C = 128): one launch (omp_swin_attn_block) vs LayerNorm + qkv GEMM + window attention + proj GEMM."""
    from advancedliteratemachinery_amd.model.packing import pack_attn_block
    B = int(os.environ.get('KBENCH_SWIN_B', '32'))
    for (C, nH, H, W) in ((128, 4, 256, 256), (256, 8, 128, 128)):
        M = B * H * W
        x = torch.randn(M, C, device=DEV)
        g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        Wqkv = (torch.randn(3 * C, C, device=DEV) / C ** 0.5).to(dtype)
        bqkv = torch.randn(3 * C, device=DEV) * 0.1
        table = torch.randn(169, nH, device=DEV) * 0.2
        bexp = ops.swin_expand_bias(table)
        Wp, bp = (torch.randn(C, C, device=DEV) / C ** 0.5).to(dtype), torch.randn(C, device=DEV) * 0.1
        y = torch.empty(M, C, device=DEV, dtype=dtype)
        out = torch.empty_like(x)
        by = 2.0 * M * C * 4
        for shift in (0, 3):
            def chain():
                ops.layernorm(x, g, b, out=y, out_dtype=dtype)
                qkv = ops.gemm(y, Wqkv, bqkv)
                ops.swin_window_attn(qkv, bqkv, table, B, H, W, C, nH, shift, out=y, bias_expanded=bexp)
                ops.gemm(y, Wp, bp, residual=x, out=out)
            us_c = timeit(chain, iters=10, warm=2)
            wpack = pack_attn_block(Wqkv, Wp, nH) if C == 256 else None

            def fused():
                if C == 128:
                    ops.swin_attn_block(x, g, b, Wqkv, bqkv, bexp, Wp, bp, B, H, W, C, nH, shift, out=out)
                else:
                    ops.swin_attn_block_packed(x, g, b, wpack, bqkv, bexp, bp, B, H, W, C, nH, shift, out=out)
            us_f = timeit(fused, iters=10, warm=2)
            print('swin_attn_block C%d B%d %dx%d shift%d : unfused chain %8.1f us   fused %8.1f us  (%5.0f GB/s of x in + out)'
                  % (C, B, H, W, shift, us_c, us_f, by / us_f / 1e3), flush=True)
            if os.environ.get('KBENCH_SWIN_TRACE', '0') == '1':   # wave 0's phase cycles, median over the persistent workgroups
                h = _lib.lib()
                trace = torch.zeros(1024, 8, dtype=torch.int64, device=DEV)
                h.omp_debug_swin_mlp_trace(ops.ptr(trace))
                fused()
                torch.cuda.synchronize()
                h.omp_debug_swin_mlp_trace(None)
                t = trace.cpu().double()
                t = t[t[:, 0] > 0]
                med = t.median(dim=0).values
                names = ['whole workgroup', 'LayerNorm + prefetch issue', 'wait B1', 'fragments + q k v', 'attention', 'B2, O, loads, B3, fragments, B4', 'proj + stores']
                for i, n in enumerate(names):
                    print('    %-34s %9.0f cycles (%5.1f%%)' % (n, med[i].item(), 100 * med[i].item() / med[0].item()), flush=True)


def bench_dec_gemm(dtype=torch.bfloat16):
    """decoder-step GEMMs: weight streaming at R = 8 rows (point decoder) and R = 512 (polygon / recognition)."""
    for R, which in ((8, 0), (64, 0), (128, 0), (256, 0), (512, 6), (2048, 6), (8192, 6), (8192, 5), (16384, 0)):
        ops.force_gemm_kernel(which if R > 64 else 0)
        for (N, K, ln) in ((1536, 512, 1), (512, 512, 0), (2048, 512, 1), (512, 2048, 0), (1104, 512, 0)):
            W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(dtype)
            bias = torch.randn(N, device=DEV)
            x = torch.randn(R, K, device=DEV)
            g, b = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
            A = x.to(dtype)
            if ln and R <= 64:
                out = torch.empty(R, N, device=DEV, dtype=dtype)
                us = timeit(lambda: ops.gemm(x, W, bias, out=out, ln=(g, b)), iters=100)
            elif ln:
                out = torch.empty(R, N, device=DEV, dtype=dtype)
                y = torch.empty(R, K, device=DEV, dtype=dtype)

                def fn():
                    ops.layernorm(x, g, b, out=y)
                    ops.gemm(y, W, bias, out=out)
                us = timeit(fn, iters=100)
            else:
                out = torch.zeros(R, N, device=DEV)
                us = timeit(lambda: ops.gemm(A, W, bias, residual=out, out=out, out_dtype=torch.float32, small_m=True), iters=100)
            print('dec_gemm[%s,k%d] R=%-4d N=%-4d K=%-4d ln=%d : %6.1f us  (weights %.2f MB -> %5.0f GB/s, %5.1f TF/s)'
                  % (str(dtype)[6:], which, R, N, K, ln, us, N * K * 2 / 1e6, N * K * 2 / us / 1e3, 2.0 * R * N * K / us / 1e6), flush=True)


def bench_dec_gemm_rows():
    """the many-row decoder phases (polygon / recognition: 64 rows per image): every Linear of a step at R = KBENCH_DEC_ROWS rows, bf16 and
    bf16x3 (K' = 3 K, split-pair A), per kernel selector -- the dispatch table of launch_gemm for these shapes is read off this."""
    R = int(os.environ.get('KBENCH_DEC_ROWS', '10240'))
    bf = torch.bfloat16
    variants = [int(v) for v in os.environ.get('KBENCH_GEMM_VARIANTS', '0,5,6,9').split(',')]
    for x3 in (0, 1):
        for (N, K, res, name) in ((1536, 512, 0, 'sa_in'), (512, 512, 1, 'sa_out'), (512, 512, 0, 'ca_q'), (2048, 512, 0, 'ff1'), (512, 2048, 1, 'ff2'), (1104, 512, 0, 'head')):
            w = torch.randn(N, K, device=DEV) / K ** 0.5
            bias = torch.randn(N, device=DEV)
            a = torch.randn(R, K, device=DEV)
            if x3:
                W, A, kw = ops.split_weight3(w), ops.split_bf16(a), dict(a_wrap=2 * K)
            else:
                W, A, kw = w.to(bf), a.to(bf), {}
            f32o = res or name in ('ca_q', 'head') or x3 and name == 'sa_in'
            sp = x3 and name == 'ff1'
            out = (torch.empty(R, 2 * N, device=DEV, dtype=bf) if sp else torch.zeros(R, N, device=DEV, dtype=torch.float32 if f32o else bf))
            r = out if res else None
            fl = 2.0 * R * N * K * (3 if x3 else 1)
            for which in variants:
                ops.force_gemm_kernel(which)
                try:
                    us = timeit(lambda: ops.gemm(A, W, bias, residual=r, out=out, out_dtype=(ops.SPLIT if sp else out.dtype), **kw), iters=30, warm=3)
                except Exception as e:   # a selector that does not take this shape
                    print('dec_rows[%s,k%d] %s : %s' % ('x3' if x3 else 'bf16', which, name, str(e)[:90]), flush=True)
                    continue
                print('dec_rows[%s,k%d] R=%d %-6s N=%-4d K=%-4d out=%s : %7.1f us  %6.1f TF/s'
                      % ('x3' if x3 else 'bf16', which, R, name, N, K * (3 if x3 else 1), 'split' if sp else str(out.dtype)[6:], us, fl / us / 1e6), flush=True)
            ops.force_gemm_kernel(0)


def bench_dec_rows_fused():
    """round 5: the decoders' many-row phases as row-owner chains (csrc/dec_rows.hip) vs the launch-per-Linear path.
    (a) every chain kernel alone at R = KBENCH_DEC_ROWS rows; (b) the polygon || recognition phase of an engine call of
    KBENCH_DEC_IMAGES images x 64 instances (hipGraph replay, two streams, M = 4096 memory tokens per image) with the chains on / off."""
    from advancedliteratemachinery_amd.model import packing
    bf = torch.bfloat16
    R = int(os.environ.get('KBENCH_DEC_ROWS', '10240'))
    d, ff, V, P = 512, 2048, 1104, 40
    W = lambda n, k: (torch.randn(n, k, device=DEV) / k ** 0.5).to(bf)   # noqa: E731
    vec = lambda n: torch.randn(n, device=DEV) * 0.1                      # noqa: E731
    x = torch.randn(R, d, device=DEV)
    att = torch.randn(R, d, device=DEV).to(bf)
    dpos = torch.tensor([5, 0], dtype=torch.int32, device=DEV)
    g, b = 1 + vec(d), vec(d)
    tab3, tab1 = torch.randn(P, 3 * d, device=DEV), torch.randn(P, d, device=DEV)
    Wo, Wq, Wc, W1, W2, Win, H0, H1, H2 = W(d, d), W(d, d), W(d, d), W(ff, d), W(d, ff), W(3 * d, d), W(d, d), W(d, d), W(V, d)
    s_mid = packing.pack_rows_mid(Wo, Wq)
    s_qkv = packing.pack_rows_ffn_qkv(Wc, W1, W2, Win)
    s_head = packing.pack_rows_ffn_head(Wc, W1, W2, H0, H1, H2)
    s_emb = packing.pack_rows_embed_qkv(Win)
    q = torch.empty(R, d, device=DEV, dtype=bf)
    qkv = torch.empty(R, 3 * d, device=DEV, dtype=bf)
    lg = torch.empty(R, V, device=DEV)
    seq = torch.randint(0, V, (R, 9), device=DEV, dtype=torch.int32)
    word, ptab = torch.randn(V, d, device=DEV), torch.randn(P, d, device=DEV)
    common = dict(att=att, out_b=vec(d), ln_g=g, ln_b=b, ff1_b=vec(ff), ff2_b=vec(d))
    hb = (vec(d), vec(d), vec(V))
    cases = [('mid  (out_proj + LN + ca_q)', 4.0 * R * d * d, lambda: ops.dec_rows_mid(att, x, s_mid[0], s_mid[1], hb[0], g, b, tab1, dpos, q=q)),
             ('ffn  (out_proj + LN + FFN + LN + q k v)', 2.0 * R * d * (d + 8 * d + 3 * d), lambda: ops.dec_rows_ffn(x, s_qkv[0], s_qkv[1], dpos, g, b, bias_tab=tab3, qkv=qkv, **common)),
             ('ffn  (out_proj + LN + FFN + LN + head)', 2.0 * R * d * (d + 8 * d + 2 * d + V), lambda: ops.dec_rows_ffn(x, s_head[0], s_head[1], dpos, g, b, head_b=hb, logits=lg, vocab=V, **common)),
             ('embed (embedding + LN + LN + q k v)', 2.0 * R * d * 3 * d, lambda: ops.dec_rows_ffn(x, s_emb[0], s_emb[1], dpos, g, b, embed=(seq, word, ptab, g, b), bias_tab=tab3, qkv=qkv))]
    tiles = [int(t) for t in os.environ.get('KBENCH_ROWS_TILES', '0,5,4,3,2').split(',')]
    for tile in tiles:
        ops.rows_tile(tile)
        for name, fl, fn in cases:
            us = timeit(fn, iters=30, warm=3)
            print('dec_rows[R=%d, tile %s] %-42s : %7.1f us  %7.1f TF/s' % (R, '16 x %d' % tile if tile else 'auto', name, us, fl / us / 1e6), flush=True)
    ops.rows_tile(0)
    # ---- (b) the phase as the engine runs it
    sys.path.insert(0, ROOT)
    import bench as B_
    I = int(os.environ.get('KBENCH_DEC_IMAGES', '160'))
    model, args, _ = B_.build_model('bf16', 1, torch.device(DEV))
    _, dec = model.engine()
    M = 4096
    gen = torch.Generator(device='cpu').manual_seed(0)
    mem = torch.randn(I * M, d, generator=gen).to(DEV, bf)
    kv = dec.project_memory(mem, mem, I, M, None)
    counts = [64] * I
    pts = torch.randint(0, args.num_bins, (64 * I, 2), generator=gen, dtype=torch.int32).to(DEV)
    dec.use_graph = True
    st = torch.cuda.Stream()
    sp, sr = torch.cuda.Stream(), torch.cuda.Stream()
    for label, thr, tile in [('row-owner chains, tile %s' % ('16 x %d' % t if t else 'auto'), 1, t) for t in tiles] + [('launch per Linear', 1 << 30, 0)]:
        dec.rows_min = thr
        ops.rows_tile(tile)
        for ph in dec._phases.values():      # the graphs captured at another tile
            ph.release_graphs()
        with torch.cuda.stream(st):
            for streams in ((sp, sr), None):
                def fn():
                    dec.decode_poly_and_rec(kv, pts, counts, args.poly_sos_index, args.rec_sos_index, args.rec_length, streams=streams)
                fn()
                torch.cuda.synchronize()
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    fn()
                b_.record()
                torch.cuda.synchronize()
                print('poly + rec phase, %d images x 64 rows, %-32s, %s : %8.2f ms per engine call'
                      % (I, label, 'two streams' if streams else 'one stream ', a.elapsed_time(b_) / 3), flush=True)
    ops.rows_tile(0)


def bench_swin_attn_trace():
    """round 5: the window attention kernel alone at a stage-2 chunk (C = 512, 16 heads, 64 x 64 tokens per image), W-MSA and SW-MSA, with the phase
    cycles of wave 0 of every workgroup (omp_debug_swin_mlp_trace -> swin_attn_mfma_kernel<bf16, expanded bias, TRACE>)."""
    bf = torch.bfloat16
    C, nH = 512, 16
    B = int(os.environ.get('KBENCH_SWIN_B', '32'))
    H = W = 64
    M = B * H * W
    qkv = torch.randn(M, 3 * C, device=DEV).to(bf)
    bqkv = torch.randn(3 * C, device=DEV) * 0.1
    table = torch.randn(169, nH, device=DEV) * 0.2
    bexp = ops.swin_expand_bias(table)
    att = torch.empty(M, C, device=DEV, dtype=bf)
    h = _lib.lib()
    nW = (H + 6) // 7
    names = ['whole window', 'token resolution + load issue of the NEXT window', 'own loads landed, V^T in LDS', 'four query tiles (S^T, softmax, PV, store)', '-']
    for shift in (0, 3):
        us = timeit(lambda: ops.swin_window_attn(qkv, bqkv, table, B, H, W, C, nH, shift, out=att, bias_expanded=bexp), iters=10, warm=2)
        byt = M * C * 2 * 4
        print('swin_attn[C=512, shift %d] %d tokens : %7.1f us  %6.2f TB/s (q, k, v in + o out)' % (shift, M, us, byt / us / 1e6), flush=True)
        nwg = B * nW * nW * (nH // 4)
        trace = torch.zeros(nwg, 8, dtype=torch.int64, device=DEV)
        h.omp_debug_swin_mlp_trace(ops.ptr(trace))
        ops.swin_window_attn(qkv, bqkv, table, B, H, W, C, nH, shift, out=att, bias_expanded=bexp)
        torch.cuda.synchronize()
        h.omp_debug_swin_mlp_trace(None)
        t = trace.cpu().double()
        t = t[t[:, 5] > 0]                      # the persistent grid writes one row per workgroup: sums over the windows it walked
        nwin = t[:, 5].sum().item()
        print('    %d workgroups x %.1f windows; per window (wave 0): ' % (t.shape[0], nwin / t.shape[0])
              + ', '.join('%s %.0f' % (names[i], t[:, i].sum().item() / nwin) for i in range(4)) + ' cycles', flush=True)


def bench_kv_rows():
    """round 5: the cross-attention memory projection of an engine call (KBENCH_KV_IMAGES images x 4096 memory tokens, 12 slabs): the two tiled GEMMs
    with slab epilogues vs the row-owner stream kernel (omp_kv_project_rows)."""
    sys.path.insert(0, ROOT)
    import bench as B_
    bf = torch.bfloat16
    I = int(os.environ.get('KBENCH_KV_IMAGES', '160'))
    M, d = 4096, 512
    model, args, _ = B_.build_model('bf16', 1, torch.device(DEV))
    _, dec = model.engine()
    mem = torch.randn(I * M, d, device=DEV).to(bf)
    mem_pos = torch.randn(I * M, d, device=DEV).to(bf)
    fl = 2 * 2.0 * I * M * d * d * dec.NL
    for on, label in ((False, 'two tiled GEMMs (OMP_STORE_KBLK / VBLK epilogues)'), (True, 'row-owner stream kernel, K + V^T launches')):
        dec.kv_rows = on
        us = timeit(lambda: dec.project_memory(mem, mem_pos, I, M, None), iters=5, warm=2)
        print('kv_project[%d images x %d keys, %d slabs] %-52s : %8.1f us  %7.1f TF/s' % (I, M, dec.NL, label, us, fl / us / 1e6), flush=True)
    dec.kv_rows = True
    # phase cycles of wave 0 (omp_debug_swin_mlp_trace): K launch then V^T launch
    h = _lib.lib()
    nwg = I * M // 64
    for vt, rows_, (sk, nk), bias, slab in ((False, mem_pos, dec._kv_streams[0], dec.bk_all, 0), (True, mem, dec._kv_streams[1], dec.bv_all, 1)):
        kv = dec.project_memory(mem, mem_pos, I, M, None)
        out = kv['Vt'] if vt else kv['K']
        trace = torch.zeros(nwg, 8, dtype=torch.int64, device=DEV)
        h.omp_debug_swin_mlp_trace(ops.ptr(trace))
        ops.kv_project_rows(rows_, sk, nk, bias, out, I, M, kv['Mpad'], dec.NL, vt)
        torch.cuda.synchronize()
        h.omp_debug_swin_mlp_trace(None)
        t = trace.cpu().double()
        print('    %s launch, wave 0 per workgroup (%d workgroups): products %.0f cycles, pack + stores %.0f cycles (12 slabs each); shader clock over that interval %.0f MHz (s_memtime / s_memrealtime)'
              % ('V^T' if vt else 'K', nwg, t[:, 2].mean().item(), t[:, 3].mean().item(), (t[:, 0] / (t[:, 4] * 0.01)).mean().item()), flush=True)


def bench_swin_rows():
    """round 5: a Swin stage-2 block (C = 512, 16 heads) at the encoder's chunk size (32 images of 1024 x 1024: 131 072 tokens): LayerNorm, qkv,
    window attention, proj, LayerNorm, fc1 + GELU, fc2 as seven launches vs window attention + ONE row-owner chain (omp_swin_rows_block)."""
    from advancedliteratemachinery_amd.model import packing
    bf = torch.bfloat16
    C, Hd, nH = 512, 2048, 16
    B = int(os.environ.get('KBENCH_SWIN_B', '32'))
    H = W = 64
    M = B * H * W
    x = torch.randn(M, C, device=DEV)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    Wm = lambda n, k: (torch.randn(n, k, device=DEV) / k ** 0.5).to(bf)   # noqa: E731
    wqkv, wp, w1, w2 = Wm(3 * C, C), Wm(C, C), Wm(Hd, C), Wm(C, Hd)
    bqkv, bp, b1, b2 = (torch.randn(n, device=DEV) * 0.1 for n in (3 * C, C, Hd, C))
    table = torch.randn(169, nH, device=DEV) * 0.2
    bexp = ops.swin_expand_bias(table)
    y = torch.empty(M, C, device=DEV, dtype=bf)
    att = torch.empty(M, C, device=DEV, dtype=bf)
    hbuf = torch.empty(M, Hd, device=DEV, dtype=bf)
    qkv = torch.empty(M, 3 * C, device=DEV, dtype=bf)
    fl = 2.0 * M * C * (3 * C + C + 2 * Hd)

    def seven():
        ops.layernorm(x, g, b, out=y, out_dtype=bf)
        ops.gemm(y, wqkv, bqkv, out=qkv)
        ops.swin_window_attn(qkv, bqkv, table, B, H, W, C, nH, 3, out=att, bias_expanded=bexp)
        ops.gemm(att, wp, bp, residual=x, out=x)
        ops.layernorm(x, g, b, out=y, out_dtype=bf)
        ops.gemm(y, w1, b1, act=ops.ACT_GELU, out=hbuf)
        ops.gemm(hbuf, w2, b2, residual=x, out=x)
    us7 = timeit(seven, iters=10, warm=2)
    print('swin_rows[C=512] %d tokens: LayerNorm, qkv, window attention, proj, LayerNorm, fc1 + GELU, fc2 (7 launches) : %8.1f us  %6.1f TF/s' % (M, us7, fl / us7 / 1e6), flush=True)
    x.normal_()
    st = packing.pack_rows_ffn_qkv(wp, w1, w2, wqkv)

    def two():
        ops.swin_window_attn(qkv, bqkv, table, B, H, W, C, nH, 3, out=att, bias_expanded=bexp)
        ops.swin_rows_block(x, att, st[0], st[1], bp, (g, b), b1, b2, next_n1=(g, b), next_qkv_b=bqkv, qkv=qkv)
    us2 = timeit(two, iters=10, warm=2)
    print('swin_rows[C=512] %d tokens: window attention + ONE chain (proj, norm2, fc1 + GELU, fc2, next norm1 + qkv)    : %8.1f us  %6.1f TF/s' % (M, us2, fl / us2 / 1e6), flush=True)
    if os.environ.get('KBENCH_ROWS_TRACE', '1') == '1':   # wave 0's cycles per phase, median over the workgroups
        h = _lib.lib()
        nwg = (M + 79) // 80
        trace = torch.zeros(nwg, 16, dtype=torch.int64, device=DEV)
        h.omp_debug_swin_mlp_trace(ops.ptr(trace))
        ops.swin_rows_block(x, att, st[0], st[1], bp, (g, b), b1, b2, next_n1=(g, b), next_qkv_b=bqkv, qkv=qkv)
        torch.cuda.synchronize()
        h.omp_debug_swin_mlp_trace(None)
        t = trace.cpu().double()
        t = t[t[:, 0] > 0]
        med = t.median(dim=0).values
        names = ['whole workgroup (wave 0)', 'prologue: attention rows -> LDS', 'proj product', 'residual + LayerNorm2 + bias', 'fc1 products (8 chunks)',
                 'barrier before the hidden tile is rewritten', 'GELU + LDS writes', 'barrier behind them', 'fc2 products (8 chunks)', 'x store',
                 'next norm1 (LayerNorm over accumulators)', 'next qkv products + stores']
        for i, n in enumerate(names):
            print('    %-46s %9.0f cycles (%5.1f%%)' % (n, med[i].item(), 100 * med[i].item() / med[0].item()), flush=True)
        print('    workgroups %d, sum of phases / whole = %.3f' % (t.shape[0], float(med[1:12].sum() / med[0])), flush=True)
    usa = timeit(lambda: ops.swin_window_attn(qkv, bqkv, table, B, H, W, C, nH, 3, out=att, bias_expanded=bexp), iters=10, warm=2)
    print('swin_rows[C=512]   of which the window attention kernel : %8.1f us; the chain : %8.1f us = %6.1f TF/s' % (usa, us2 - usa, fl / (us2 - usa) / 1e6), flush=True)


def bench_patch_embed():
    """PatchEmbed + LayerNorm of one 32-image encoder chunk at 1024x1024: the fp32 matrix-core kernel vs the thread-per-token kernel."""
    B, H, W, E = int(os.environ.get('KBENCH_PE_B', '32')), 1024, 1024, 128
    img = torch.randn(B, 3, H, W, device=DEV)
    w, b = torch.randn(E, 48, device=DEV) * 0.2, torch.randn(E, device=DEV) * 0.1
    g, be = torch.ones(E, device=DEV), torch.zeros(E, device=DEV)
    by = img.numel() * 4 + B * (H // 4) * (W // 4) * E * 4
    outs = {}
    for name, env in (('mfma', None), ('thread-per-token', '1')):
        if env:
            os.environ['OMP355_PATCH_EMBED_SCALAR'] = env
        else:
            os.environ.pop('OMP355_PATCH_EMBED_SCALAR', None)
        us = timeit(lambda: ops.patch_embed_ln(img, w, b, g, be, torch.float32), iters=10, warm=2)
        outs[name] = ops.patch_embed_ln(img, w, b, g, be, torch.float32)[0]
        print('patch_embed[%-16s] B%d %dx%d E%d : %8.1f us  %6.0f GB/s (%.2f of 8 TB/s)' % (name, B, H, W, E, us, by / us / 1e3, by / us / 1e3 / 8000), flush=True)
    os.environ.pop('OMP355_PATCH_EMBED_SCALAR', None)
    print('patch_embed max |mfma - thread-per-token| = %.3g' % (outs['mfma'] - outs['thread-per-token']).abs().max().item(), flush=True)


def bench_selfattn(dtype=torch.bfloat16):
    ops.force_gemm_kernel(0)
    d, nH = 512, 8
    for (R, Lmax, pos) in ((8, 140, 70), (128, 140, 70), (128, 140, 135), (512, 40, 20), (512, 40, 34), (2048, 40, 34), (8192, 36, 17), (8192, 36, 34)):
        qkv = torch.randn(R, 3 * d, device=DEV).to(dtype)
        kc = torch.randn(R, Lmax, d, device=DEV).to(dtype)
        vc = torch.randn(R, Lmax, d, device=DEV).to(dtype)
        out = torch.empty(R, d, device=DEV, dtype=dtype)
        dp = torch.tensor([pos], dtype=torch.int32, device=DEV)
        by = R * (pos + 1) * d * 2 * 2
        for impl in (1, 2):
            _lib.lib().omp_debug_self_attn_impl(impl)
            us = timeit(lambda: ops.dec_self_attn_step(qkv, kc, vc, out, dp, nH), iters=100)
            print('selfattn[%s] R=%-4d pos=%-3d impl=%d : %6.1f us  %6.0f GB/s' % (str(dtype)[6:], R, pos, impl, us, by / us / 1e3), flush=True)
        _lib.lib().omp_debug_self_attn_impl(0)


def bench_misc(dtype=torch.bfloat16):
    B, H, W = 8, 256, 256
    for (C, nH, hh) in ((128, 4, 256), (256, 8, 128), (512, 16, 64), (1024, 32, 32)):
        qkv = torch.randn(B * hh * hh, 3 * C, device=DEV).to(dtype)
        bias = torch.randn(3 * C, device=DEV)
        tab = torch.randn(169, nH, device=DEV)
        out = torch.empty(B * hh * hh, C, device=DEV, dtype=dtype)
        bexp = ops.swin_expand_bias(tab)
        for shift in (0, 3):
            by = B * hh * hh * 4 * C * 2
            for name, be in (('table', None), ('expanded', bexp)):
                ops.swin_attn_impl(0 if be is not None else 2)
                us = timeit(lambda: ops.swin_window_attn(qkv, bias, tab, B, hh, hh, C, nH, shift, out=out, bias_expanded=be), iters=10, warm=2)
                print('swin_attn C=%-4d shift=%d %-8s : %8.1f us  %6.0f GB/s' % (C, shift, name, us, by / us / 1e3), flush=True)
        ops.swin_attn_impl(0)
        x = torch.randn(B * hh * hh, C, device=DEV).to(dtype)
        g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        y = torch.empty_like(x)
        us = timeit(lambda: ops.layernorm(x, g, b, out=y), iters=10, warm=2)
        print('layernorm rows=%d C=%-4d : %8.1f us  %6.0f GB/s' % (B * hh * hh, C, us, 2 * x.numel() * 2 / us / 1e3), flush=True)


if __name__ == '__main__':
    what = sys.argv[1:] or ['all']
    print(torch.cuda.get_device_name(0), flush=True)
    _lib.lib()
    if 'swin_block' in what or 'all' in what:
        bench_swin_block()
    if 'cross' in what or 'all' in what:
        bench_cross()
    if 'dec_gemm' in what or 'all' in what:
        bench_dec_gemm()
    if 'swin_rows' in what:
        bench_swin_rows()
    if 'swin_attn_trace' in what:
        bench_swin_attn_trace()
    if 'kv_rows' in what:
        bench_kv_rows()
    if 'dec_rows_fused' in what:
        bench_dec_rows_fused()
    if 'patch_embed' in what:
        bench_patch_embed()
    if 'cross_split' in what:
        bench_cross_split()
    if 'dec_rows' in what:
        bench_dec_gemm_rows()
    if 'selfattn' in what or 'all' in what:
        bench_selfattn()
    if 'gemm' in what or 'all' in what:
        bench_gemm()
    if 'misc' in what or 'all' in what:
        bench_misc()
    if 'mlp' in what or 'all' in what:
        bench_mlp()
    if 'kvproj' in what:
        bench_kvproj()
    if 'cross128' in what:
        bench_cross128()
