"""One encoder chunk (Swin-B + FPN + input_proj on 80 images of 1024x1024: what bench.py's engine calls encode per pass) plus the K / V^T
projection of those images, launched eagerly for the rocprofv3 --pmc passes that measure the HBM traffic of the matrix-core classes of the
bench line (FETCH_SIZE, WRITE_SIZE; separate passes):
  * encoder-sized tile GEMMs (gemm_256 / gemm_4w* / gemm_dma<128,128>): what is left on them after round 5 -- stage 3, PatchMerging reductions, FPN
    laterals, input_proj, and (parity engine only) the K / V^T projection;
  * fused MLP (stages 0 / 1) + the Swin stage-2 row-owner chain (dec_rows_ffn_kernel): the largest matrix-core class by GPU time;
  * the memory projection (kv_rows_kernel).
`run [images] [engine]` prints the ALGORITHMIC bytes of every launch of those classes -- GEMMs: M K + N K + M N (+ M N residual / second destination), each at the
element size it is launched with; fused MLP: x in + y out + packed weights; chains: the per-launch formula of csrc/dec_rows.hip (rows in and out +
the weight stream once); projection: rows in + slabs out + weights -- and `summarise <fetch.csv> <write.csv> <alg.json>` puts the measured bytes
(FETCH_SIZE KiB x 2: gfx950 counts 16-byte-per-lane streaming reads at half their bytes, MI355X_MICROARCH.md; + WRITE_SIZE KiB) next to them ->
profiles/pmc_gemm.json.  The encoder runs twice (warm-up + counted: not separable under rocprofv3), the summary halves its kernels' totals."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    from advancedliteratemachinery_amd import ops
    from advancedliteratemachinery_amd.model import OmniParser
    from advancedliteratemachinery_amd.utils import synthetic as weights
    from advancedliteratemachinery_amd.utils.parser import make_args
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    engine = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
    args = make_args(tfm_pre_norm=True, use_fpn=True, use_char_window_prompt=True)
    model = OmniParser(args, engine_dtype=engine)
    model.load_state_dict(weights.make_state_dict(args, seed=0))
    model = model.to('cuda:0')
    enc, dec = model.engine()
    img = torch.randn(B, 3, 1024, 1024, generator=torch.Generator().manual_seed(3)).to('cuda:0')
    mask = torch.zeros(B, 1024, 1024, dtype=torch.bool, device='cuda:0')
    log = collections.defaultdict(float)
    real = dict(gemm=ops.gemm, mlp=ops.swin_mlp_fused, rows=ops.swin_rows_block, qkv0=ops.swin_rows_qkv, kv=ops.kv_project_rows)
    x3 = engine == 'bf16x3'

    def gemm(A, W, bias=None, residual=None, **kw):
        K = kw.get('K') or A.shape[-1]
        N = kw.get('N') or W.shape[0]
        M = kw.get('M') or A.numel() // A.shape[-1]
        ea, ew = A.element_size(), W.element_size()
        out = kw.get('out')
        od = kw.get('out_dtype') or W.dtype
        eo = out.element_size() if out is not None and kw.get('store_mode') is None else (4 if (od == ops.SPLIT or od == torch.float32) else 2)
        mn = M * N * eo
        if residual is not None:
            mn += M * N * residual.element_size()
        if kw.get('out_noresidual') is not None:
            mn += M * N * kw['out_noresidual'].element_size()
        log['gemm_launches'] += 1
        log['gemm_alg_bytes'] += float(M * K * ea + N * K * ew + mn)
        log['gemm_flops'] += 2.0 * M * N * K
        return real['gemm'](A, W, bias, residual=residual, **kw)

    def mlp(x, g, b, wpack, b2, out=None, eps=1e-5):
        log['mlp_launches'] += 1
        log['mlp_alg_bytes'] += 2.0 * x.numel() * x.element_size() + wpack.numel() * wpack.element_size()
        return real['mlp'](x, g, b, wpack, b2, out=out, eps=eps)

    def rows(x, att, wstream, wave_stride, *a, **kw):   # csrc/dec_rows.hip omp_swin_rows_block mode 1: `by`
        M = x.numel() // 512
        tail = kw.get('next_n1') is not None
        es = 4 if x3 else 2
        frags = 64 + 8 * 64 + (192 if tail else 0)
        log['chain_launches'] += 1
        log['chain_alg_bytes'] += float(M * 512 * (es + 4 + 4) + (M * 1536 * es if tail else 0) + frags * 8192 * (2 if x3 else 1))
        log['chain_flops'] += 2.0 * M * 512 * (9 * 512 + (3 * 512 if tail else 0)) * (3 if x3 else 1)
        return real['rows'](x, att, wstream, wave_stride, *a, **kw)

    def qkv0(x, *a, **kw):   # mode 0: the first block's norm1 + qkv
        M = x.numel() // 512
        es = 4 if x3 else 2
        log['chain_launches'] += 1
        log['chain_alg_bytes'] += float(M * 512 * 4 + M * 1536 * es + 192 * 8192 * (2 if x3 else 1))
        log['chain_flops'] += 2.0 * M * 512 * 1536 * (3 if x3 else 1)
        return real['qkv0'](x, *a, **kw)

    def kv(mem, stream, n, bias, out, B_, M_, Mpad, NL, swap):
        log['kv_launches'] += 1
        log['kv_alg_bytes'] += float(B_ * M_ * 512 * 2 + NL * B_ * Mpad * 512 * 2 + NL * 512 * 512 * 2)
        return real['kv'](mem, stream, n, bias, out, B_, M_, Mpad, NL, swap)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        e = enc.encode(img, mask)            # warm-up pass: counted by rocprofv3 all the same, the summary halves the encoder kernels
        torch.cuda.synchronize()
        ops.gemm, ops.swin_mlp_fused, ops.swin_rows_block, ops.swin_rows_qkv, ops.kv_project_rows = gemm, mlp, rows, qkv0, kv
        try:
            e = enc.encode(img, mask)
            n_enc = log['gemm_launches']
            dec.project_memory(e['memory'], e['mem_pos'], B, e['M'], None)
            torch.cuda.synchronize()
        finally:
            ops.gemm, ops.swin_mlp_fused, ops.swin_rows_block, ops.swin_rows_qkv, ops.kv_project_rows = (real['gemm'], real['mlp'], real['rows'], real['qkv0'], real['kv'])
    out = dict(log)
    out.update(images=B, engine=engine, gemm_launches_encoder=n_enc,
               note='second of two identical encoder passes + one K / V^T projection; the first pass has no projection')
    print(json.dumps(out))


def read(path, ctr):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get('Counter_Name') != ctr:
                continue
            k = r['Kernel_Name']
            if 'kv_rows_kernel' in k:
                c = 'kv'
            elif 'mlp_fused_kernel' in k:
                c = 'mlp'
            elif 'dec_rows_ffn_kernel' in k or 'dec_rows_x3_ffn_kernel' in k:
                c = 'chain'
            elif 'gemm_256' in k or 'gemm_4w' in k or ('gemm_dma' in k and ('Li128ELi128E' in k or '128, 128' in k)):
                c = 'gemm'
            else:
                continue
            agg[c][0] += 1
            agg[c][1] += float(r['Counter_Value'])
    return agg


def summarise(fetch_csv, write_csv, alg_json):
    alg = json.load(open(alg_json))
    f, w = read(fetch_csv, 'FETCH_SIZE'), read(write_csv, 'WRITE_SIZE')
    out = {c: dict(launches_in_trace=f[c][0], fetch_kib_total=f[c][1], write_kib_total=w[c][1]) for c in ('gemm', 'mlp', 'chain', 'kv')}
    meas = lambda c, div: (f[c][1] * 2.0 + w[c][1]) * 1024.0 / div     # noqa: E731
    n_proj = alg['gemm_launches'] - alg['gemm_launches_encoder']        # projection GEMMs ran once (parity engine), encoder GEMMs twice
    # exact split of the GEMM class is not possible from per-kernel totals when the projection is a GEMM too: bf16 engine -> kv_rows_kernel, n_proj == 0
    g_meas = meas('gemm', 2.0) if n_proj == 0 else None
    mc_alg = alg.get('mlp_alg_bytes', 0.0) + alg.get('chain_alg_bytes', 0.0)
    mc_meas = meas('mlp', 2.0) + meas('chain', 2.0)
    mc_n = int(alg.get('mlp_launches', 0) + alg.get('chain_launches', 0))
    s = dict(images=alg['images'], engine=alg['engine'],
             gemm_launches=int(alg['gemm_launches']), gemm_alg_bytes=alg['gemm_alg_bytes'], gemm_measured_bytes=g_meas,
             gemm_measured_over_alg=(g_meas / alg['gemm_alg_bytes']) if g_meas else None,
             gemm_measured_bytes_per_launch=(g_meas / alg['gemm_launches']) if g_meas else None,
             mlp_chain_launches=mc_n, mlp_chain_alg_bytes=mc_alg, mlp_chain_measured_bytes=mc_meas, mlp_chain_measured_over_alg=mc_meas / mc_alg if mc_alg else None,
             mlp_chain_measured_bytes_per_launch=mc_meas / mc_n if mc_n else None,
             mlp_launches=int(alg.get('mlp_launches', 0)), mlp_alg_bytes=alg.get('mlp_alg_bytes', 0.0), mlp_measured_bytes=meas('mlp', 2.0),
             chain_launches=int(alg.get('chain_launches', 0)), chain_alg_bytes=alg.get('chain_alg_bytes', 0.0), chain_measured_bytes=meas('chain', 2.0),
             chain_measured_over_alg=(meas('chain', 2.0) / alg['chain_alg_bytes']) if alg.get('chain_alg_bytes') else None,
             kv_launches=int(alg.get('kv_launches', 0)), kv_alg_bytes=alg.get('kv_alg_bytes', 0.0), kv_measured_bytes=meas('kv', 1.0),
             kv_measured_over_alg=(meas('kv', 1.0) / alg['kv_alg_bytes']) if alg.get('kv_alg_bytes') else None,
             note='one encoder chunk of %d images (%s engine) + its K / V^T projection; FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE; classes as bench.py brackets them: '
                  'gemm = tile GEMMs at M >= 32768 rows, mlp_chain = mlp_fused_kernel + the Swin stage-2 chain, kv = kv_rows_kernel' % (alg['images'], alg['engine']))
    out['summary'] = s
    out['command'] = 'rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python tools/encode_pmc.py run %d %s' % (alg['images'], alg['engine'])
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'summarise':
        summarise(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        run()
