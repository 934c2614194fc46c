"""GPU parity checks of the MGP-STR path (BASELINE config 5): libomp355 through the C ABI wrappers vs the CPU
oracle (oracle/mgp_str_ref.py, pinned to the reference in tests/test_oracle_mgp.py) and vs the golden fixture
produced by the reference's own code.  Same record format as tests/gpu_checks.py."""
import os

import torch
import torch.nn.functional as F

from advancedliteratemachinery_amd import ops
from advancedliteratemachinery_amd.model import mgp_str as M
from oracle import gen_golden_mgp as G
from oracle import mgp_str_ref as R
from tests.gpu_checks import DEV, DTYPES, maxerr, q, rec, rnd

# engine precisions of the MGP-STR end-to-end checks; bf16x3 (the parity engine, round 5) is held to the fp32 gates
ENGINES = dict(DTYPES, bf16x3='bf16x3')
F32_GRADE = ('fp32', 'bf16x3')


def build(c, sd, dtype):
    m = M.MGPSTR({k: c[k] for k in ('embed', 'depth', 'heads', 'mlp_ratio', 'img', 'patch', 'max_len', 'num_class')}, engine_dtype=dtype)
    m.load_reference_state_dict({'module.' + k: v for k, v in sd.items()})
    return m.to(DEV)


def check_vit_patch_embed():
    out = []
    c = R.cfg(depth=1)
    sd = R.make_state_dict(c, seed=2)
    for (B, H, W) in ((3, 32, 128), (2, 8, 20)):
        cc = dict(c, img=(H, W))
        T = (H // 4) * (W // 4) + 1
        sd2 = dict(sd)
        sd2['mgp_str.pos_embed'] = rnd(1, T, 768, seed=4) * 0.1
        img = rnd(B, 3, H, W, seed=H)
        ref = R.embed(sd2, cc, img)
        for dn, dt in DTYPES.items():
            y = ops.vit_patch_embed(img.to(DEV), sd2['mgp_str.patch_embed.proj.weight'].reshape(768, 48).contiguous().to(DEV),
                                    sd2['mgp_str.patch_embed.proj.bias'].to(DEV), sd2['mgp_str.cls_token'].reshape(768).contiguous().to(DEV),
                                    sd2['mgp_str.pos_embed'].reshape(T, 768).contiguous().to(DEV), dt)
            out.append(rec('vit_patch_embed[%s,B%d %dx%d]' % (dn, B, H, W), maxerr(y, ref), 2e-5 if dt == torch.float32 else 2e-2))
    return out


def check_a3_pool():
    out = []
    for (B, T, S, C) in ((3, 257, 27, 768), (2, 50, 5, 100), (1, 257, 27, 1024)):
        sel = rnd(B * T, S, seed=T) * 2.0
        for dn, dt in DTYPES.items():
            feat = q(rnd(B * T, C, seed=C), dt)
            maps = F.softmax(sel.reshape(B, T, S).transpose(1, 2), dim=-1)
            ref = torch.einsum('bsi,bid->bsd', maps, feat.reshape(B, T, C)).reshape(B * S, C)
            pooled, attn = ops.a3_pool(sel.to(DEV), feat.to(DEV, dt), B, T, S, True)
            out.append(rec('a3_pool[%s,B%d T%d S%d C%d] pooled' % (dn, B, T, S, C), maxerr(pooled, ref), 2e-5))
            out.append(rec('a3_pool[%s,B%d T%d S%d C%d] maps' % (dn, B, T, S, C), maxerr(attn, maps), 2e-6))
    return out


def check_row_argmax_prob():
    out = []
    for (Rn, V) in ((81, 50257), (54, 38), (7, 30522)):
        lg = rnd(Rn, V, seed=V) * 3.0
        ids, pr = ops.row_argmax_prob(lg.to(DEV))
        ref_p, ref_i = F.softmax(lg, dim=1).max(dim=1)
        out.append(rec('row_argmax_prob[%dx%d] ids' % (Rn, V), float((ids.cpu().long() != ref_i).sum()), 0))
        out.append(rec('row_argmax_prob[%dx%d] prob' % (Rn, V), maxerr(pr, ref_p), 1e-6))
    return out


def check_vit_attn():
    """csrc/vit.hip::vit_attn_kernel (bf16, one workgroup per (image, head)) vs softmax(q k^T / 8) v in fp32 on the SAME
    bf16 q / k / v, with the slabs written by the GEMM store modes the model uses; also vs the blocked cross-attention
    path on the same slabs (the fp32 engine's route)."""
    from advancedliteratemachinery_amd import _lib
    out = []
    nH, E = 12, 768
    for (B, T) in ((1, 257), (5, 257), (2, 100)):
        Mpad, KB = 288, 32
        y = q(rnd(B * T, E, seed=B * T), torch.bfloat16).to(DEV, torch.bfloat16)
        eye = torch.eye(E, dtype=torch.bfloat16, device=DEV)
        kk = q(rnd(B * T, E, seed=B * T + 1), torch.bfloat16).to(DEV, torch.bfloat16)
        vv = q(rnd(B * T, E, seed=B * T + 2), torch.bfloat16).to(DEV, torch.bfloat16)
        K = torch.zeros(1, B, nH, Mpad, 64, dtype=torch.bfloat16, device=DEV)
        Vt = torch.zeros(1, B, nH, Mpad // KB, 64, KB, dtype=torch.bfloat16, device=DEV)
        geom = (B, T, Mpad, nH, KB)
        # identity projections copy k / v into the slabs through the epilogues under test elsewhere (exact in bf16)
        ops.gemm(kk, eye, None, out=K, store_mode=_lib.STORE_KBLK, kv=geom)
        ops.gemm(eye, vv, None, out=Vt, store_mode=_lib.STORE_VBLK, kv=geom, M=E, N=B * T, K=E)
        att = torch.empty_like(y)
        ops.vit_attn(y, K[0], Vt[0], att, B, T, nH, Mpad)
        qf = y.float().cpu().reshape(B, T, nH, 64).permute(0, 2, 1, 3)
        kf = kk.float().cpu().reshape(B, T, nH, 64).permute(0, 2, 1, 3)
        vf = vv.float().cpu().reshape(B, T, nH, 64).permute(0, 2, 1, 3)
        ref = (F.softmax(qf @ kf.transpose(-1, -2) * 0.125, dim=-1) @ vf).permute(0, 2, 1, 3).reshape(B * T, E)
        out.append(rec('vit_attn[B%d,T%d] vs fp32 attention' % (B, T), maxerr(att, ref), 1.5e-2, 'max|ref|=%.2f' % ref.abs().max().item()))
        groups = []
        for b in range(B):
            for o in range(0, T, 64):
                groups.append((b * T + o, min(64, T - o), b))
        g = torch.tensor(groups, dtype=torch.int32, device=DEV)
        att2 = torch.empty_like(y)
        ops.dec_cross_attn_step(y, K[0], Vt[0], nH * Mpad * 64, Mpad, None, g, len(groups), 4, None, att2, T, nH, 1)
        out.append(rec('vit_attn[B%d,T%d] vs blocked cross-attention' % (B, T), maxerr(att, att2.float().cpu()), 1.5e-2))
    return out


def check_vit_block(dtype_name='fp32'):
    """one encoder block (LN, q/k/v projections into the blocked slabs, 257-token attention on the cross-attention
    kernels, proj + residual, MLP) vs the oracle's block, and the slab path at two batch sizes"""
    dt = ENGINES[dtype_name]
    out = []
    c = R.cfg(depth=1)
    sd = R.make_state_dict(c, seed=8)
    model = build(c, sd, dt)
    for B in (1, 3):
        img = rnd(B, 3, 32, 128, seed=B)
        with torch.no_grad():
            ref = R.encoder(sd, c, img)
        x, _, T = model.encode(img.to(DEV))
        out.append(rec('vit_block[%s,B%d]' % (dtype_name, B), maxerr(x.reshape(B, T, -1), ref), {'fp32': 2e-4, 'bf16x3': 5e-4}.get(dtype_name, 0.15),
                       'max|ref|=%.1f' % ref.abs().max().item()))
    return out


def check_mgp_e2e(dtype_name='fp32', depth=2):
    dt = ENGINES[dtype_name]
    f32 = dtype_name in F32_GRADE
    c = R.cfg(depth=depth)
    sd = R.make_state_dict(c, seed=21)
    model = build(c, sd, dt)
    img = rnd(3, 3, 32, 128, seed=5).clamp(-1, 1)
    with torch.no_grad():
        ratt, rch, rbp, rwp = R.forward(sd, c, img)
    att, ch, bp, wp = model(img.to(DEV), is_eval=True)
    out = []
    # fp32-grade engines: north_star's 1e-3 on logits.  bf16: a bound RELATIVE to the logit scale (round 5; it was 0.5 absolute, a gate
    # that tested wiring, not numerics -- VERDICT r4): measured 1.0-1.6 % of max|logit| at depth 2, held to 3 %
    atol = {'fp32': 2e-5, 'bf16x3': 1e-4}.get(dtype_name, 2e-2)   # attention maps are softmax outputs <= 1: the split products' 2^-17 operand error shows here
    for name, a, b in (('char', ch, rch), ('bpe', bp, rbp), ('wp', wp, rwp)):
        scale = b.abs().max().item()
        err = maxerr(a, b)
        out.append(rec('mgp_logits[%s,depth%d,%s]' % (dtype_name, depth, name), err if f32 else err / scale, 1e-3 if f32 else 0.03,
                       'max|logit|=%.1f abs %.3g' % (scale, err)))
    for name, a, b in zip(('char', 'bpe', 'wp'), att, ratt):
        out.append(rec('mgp_attn[%s,depth%d,%s]' % (dtype_name, depth, name), maxerr(a, b), atol))
    # result decoding on the device vs the oracle's restatement of test_final.py
    got = model.recognize(img.to(DEV))
    want = R.decode(rch, rbp, rwp)
    same = tot = 0
    for g, w in zip(got, want):
        for k in ('char_ids', 'bpe_ids', 'wp_ids'):
            tot += len(w[k])
            same += sum(int(x == y) for x, y in zip(g[k], w[k]))
    frac = same / max(1, tot)
    out.append(rec('mgp_greedy_ids[%s,depth%d]' % (dtype_name, depth), 1.0 - frac, 0.0 if f32 else 0.3, 'agreement %.3f' % frac))
    if f32:
        worst = max(max(abs(x - y) for x, y in zip(g['conf'], w['conf'])) for g, w in zip(got, want))
        out.append(rec('mgp_confidence[%s,depth%d]' % (dtype_name, depth), worst, 1e-4 if dtype_name == 'fp32' else 1e-3))
        out.append(rec('mgp_choice[%s,depth%d]' % (dtype_name, depth), sum(int(g['choice'] != w['choice']) for g, w in zip(got, want)), 0))
    return out


def check_mgp_golden(dtype_name='fp32'):
    """full ViT-B (12 blocks, full vocabularies) in an fp32-grade engine (fp32, or bf16x3 = the parity engine) vs the fixture written by the
    reference's own code"""
    fix = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mgp_str_base.pt'), weights_only=False)
    c, ref = fix['cfg'], fix['ref']
    sd = R.make_state_dict(c, seed=fix['seed_w'])
    model = build(c, sd, ENGINES[dtype_name])
    img = fix['img']
    x, B, T = model.encode(img.to(DEV))
    att, ch, bp, wp = model(img.to(DEV), is_eval=True)
    got = G.reduce_outputs(x.reshape(B, T, -1).float().cpu(), [a.cpu() for a in att], ch.cpu(), bp.cpu(), wp.cpu())
    out = [rec('mgp_golden enc_proj', maxerr(got['enc_proj'], ref['enc_proj']), 1e-3),
           rec('mgp_golden char logits', maxerr(got['char'], ref['char']), 1e-3),
           rec('mgp_golden bpe logits (64 cols)', maxerr(got['bpe_sub'], ref['bpe_sub']), 1e-3),
           rec('mgp_golden wp logits (64 cols)', maxerr(got['wp_sub'], ref['wp_sub']), 1e-3)]
    for name, a, b in zip(('char', 'bpe', 'wp'), got['attens'], ref['attens']):
        out.append(rec('mgp_golden attn %s' % name, maxerr(a, b), 2e-5 if dtype_name == 'fp32' else 1e-4))
    for name in ('char', 'bpe', 'wp'):
        out.append(rec('mgp_golden %s ids' % name, float((got[name + '_ids'] != ref[name + '_ids']).sum()), 0))
        out.append(rec('mgp_golden %s prob' % name, maxerr(got[name + '_prob'], ref[name + '_prob']), 1e-4 if dtype_name == 'fp32' else 1e-3))
    return out


def check_mgp_b512(dtype_name='fp32', B=512, probe=(0, 1, 63, 64, 255, 256, 300, 511)):
    """BASELINE config 5 at its stated shape: full ViT-B (12 blocks, full vocabularies), batch 512.  Words are independent,
    so the oracle (CPU fp32) is run on a probe set of batch positions (first / last / tile edges) and compared with
    the engine's rows of the 512-batch: logits, attention maps and greedy ids of all three heads.
    bf16 (the benchmarked precision of config 5): errors relative to max|logit|, measured values in
    profiles/r02_parity_report.json; ids must agree wherever the oracle's top-1/top-2 margin exceeds 2x the logit error."""
    from tests.gpu_checks import REPORT
    dt = ENGINES[dtype_name]
    f32 = dtype_name in F32_GRADE
    c = R.cfg()
    sd = R.make_state_dict(c, seed=33)
    model = build(c, sd, dt)
    img = rnd(B, 3, 32, 128, seed=77).clamp(-1, 1)
    idx = torch.tensor([p for p in probe if p < B])
    with torch.no_grad():
        ratt, rch, rbp, rwp = R.forward(sd, c, img[idx])
    att, ch, bp, wp = model(img.to(DEV), is_eval=True)
    out = []
    for name, a, b in (('char', ch, rch), ('bpe', bp, rbp), ('wp', wp, rwp)):
        a = a.float().cpu()[idx]
        scale = b.abs().max().item()
        err = (a - b).abs().max().item()
        REPORT.append(dict(name='mgp_b%d[%s] %s logits' % (B, dtype_name, name), abs_err=err, rel_err=err / scale, ref_absmax=scale))
        out.append(rec('mgp_b%d[%s] %s logits' % (B, dtype_name, name), err if f32 else err / scale, 1e-3 if f32 else 0.03,
                       'max|logit|=%.1f abs %.3g' % (scale, err)))
        top2 = b.topk(2, dim=-1).values
        margin = top2[..., 0] - top2[..., 1]
        agree = a.argmax(-1) == b.argmax(-1)
        clear = margin > 2.0 * err
        REPORT.append(dict(name='mgp_b%d[%s] %s argmax' % (B, dtype_name, name), agree=float(agree.float().mean()), positions=int(agree.numel()),
                           inside_noise_band=int((~clear).sum())))
        out.append(rec('mgp_b%d[%s] %s ids (margin > 2x err)' % (B, dtype_name, name), float((clear & ~agree).sum()), 0,
                       'agree %.4f of %d' % (float(agree.float().mean()), agree.numel())))
    for name, a, b in zip(('char', 'bpe', 'wp'), att, ratt):
        out.append(rec('mgp_b%d[%s] %s attention maps' % (B, dtype_name, name), maxerr(a.float().cpu()[idx], b), {'fp32': 2e-5, 'bf16x3': 1e-4}.get(dtype_name, 2e-2)))
    return out


def check_two_stage():
    """SURVEY 8f row 4 end to end in fp32: uint8 images -> device pre-processing -> OmniParser -> polygon boxes -> device
    bicubic crops -> MGP-STR -> fused decoding, against the oracle chain built from the reference's pieces and the real
    Pillow (oracle/two_stage_ref.py).  Demands identical boxes, identical greedy ids of all three heads, identical fused
    choice, confidences within 1e-4."""
    import numpy as np
    from advancedliteratemachinery_amd.engine.two_stage import spot_and_recognize
    from advancedliteratemachinery_amd.utils.parser import make_args
    from oracle import two_stage_ref as T
    from advancedliteratemachinery_amd.utils import synthetic as weights
    from tests.gpu_checks import build_model
    depths = (2, 2, 2, 2)
    args = make_args(tfm_pre_norm=True, use_fpn=True, use_char_window_prompt=True, pt_seq_length=6, test_min_size=64, test_max_size=112)
    sd = weights.make_state_dict(args, seed=5, depths=depths)
    omni = build_model(args, sd, depths, torch.float32)
    c = R.cfg(depth=2)
    sdm = R.make_state_dict(c, seed=17)
    mgp = build(c, sdm, torch.float32)
    rng = np.random.RandomState(3)
    images = [rng.randint(0, 256, (90, 140, 3), dtype=np.uint8), rng.randint(0, 256, (70, 100, 3), dtype=np.uint8)]
    got, _, _ = spot_and_recognize(omni, mgp, [torch.from_numpy(i) for i in images], args)
    want = T.chain(sd, args, depths, sdm, c, images, args.test_min_size, args.test_max_size)
    out = []
    n_words = 0
    for b, (g, w) in enumerate(zip(got, want)):
        out.append(rec('two_stage img%d: detections' % b, abs(len(g) - len(w)), 0, '%d vs %d' % (len(g), len(w))))
        for i, (rg, rw) in enumerate(zip(g, w)):
            n_words += 1
            tag = 'two_stage img%d word%d' % (b, i)
            out.append(rec(tag + ' box', 0 if tuple(rg['box']) == tuple(rw['box']) else 1, 0, '%s vs %s' % (rg['box'], rw['box'])))
            # the OmniParser record itself (predict_images -> decode_pred_seq, engine/val.py:70-100) against the oracle's
            out.append(rec(tag + ' polygon (original-image pixels)', (torch.tensor(rg['polys']) - torch.tensor(rw['polys'])).abs().max().item(), 1e-3))
            chars = []
            for t in rw['rec_ids']:
                if t == args.recog_pad_index or t == args.rec_eos_index:
                    break
                if t == args.recog_pad_index - 1:
                    continue
                chars.append(args.chars[t - args.num_bins])
            out.append(rec(tag + ' OmniParser text', 0 if rg['rec'] == ''.join(chars) else 1, 0, '%r' % rg['rec']))
            for k in ('char', 'bpe', 'wp'):
                out.append(rec(tag + ' %s ids' % k, sum(int(x != y) for x, y in zip(rg['mgp_ids'][k], rw[k + '_ids'])), 0))
            out.append(rec(tag + ' choice', 0 if rg['mgp_choice'] == rw['choice'] else 1, 0))
            out.append(rec(tag + ' confidences', max(abs(x - y) for x, y in zip(rg['mgp_conf'], rw['conf'])), 1e-4))
            out.append(rec(tag + ' text', 0 if rg['mgp_text'] == rw['char_text'] else 1, 0))
    out.append(rec('two_stage: words recognised', 0 if n_words > 0 else 1, 0, '%d words' % n_words))
    return out


def check_crop_resizer():
    """device bicubic crops == PIL crop + resize(BICUBIC) + ToTensor, every float equal"""
    import numpy as np
    from PIL import Image
    from advancedliteratemachinery_amd.utils.preprocess import CropResizer
    rng = np.random.RandomState(9)
    imgs = [rng.randint(0, 256, (120, 200, 3), dtype=np.uint8), rng.randint(0, 256, (64, 48, 3), dtype=np.uint8)]
    boxes = [(0, 0, 0, 200, 120), (0, 17, 5, 150, 37), (0, 100, 60, 101, 61), (1, 3, 2, 47, 60), (1, 0, 0, 48, 64), (0, 10, 10, 138, 42)]
    cr = CropResizer(DEV)
    got = cr([torch.from_numpy(i).to(DEV) for i in imgs], boxes).cpu()
    worst = 0.0
    for n, (bi, x0, y0, x1, y1) in enumerate(boxes):
        ref = np.asarray(Image.fromarray(imgs[bi]).crop((x0, y0, x1, y1)).resize((128, 32), Image.BICUBIC)).astype(np.float32) / np.float32(255.0)
        worst = max(worst, float((got[n] - torch.from_numpy(ref).permute(2, 0, 1)).abs().max()))
    return [rec('crop_resizer == Pillow bicubic', worst, 0.0)]


def check_gemm_row_stats():
    """omp_gemm_bias_act(OMP_STORE_ROWSTAT) + omp_row_stat_merge (round 6: greedy decoding without the logits tensor) against the path it replaces --
    the same product written as fp32 logits, then omp_row_argmax_prob -- on ragged shapes: ids identical (the product bits are the same; ties go to
    the lowest index in both), probabilities within 2e-6 relative (a different summation order of the exponentials).  Engines: bf16 operands, fp32
    operands, and the bf16x3 form (split-pair rows against the [hi | hi | lo] weight image)."""
    out = []
    for (M, N, K) in ((300, 50257, 768), (129, 1104, 512), (64, 128, 256), (1000, 30522, 768)):
        A, W, b = rnd(M, K, seed=M + 1), rnd(N, K, seed=N + 2) / K ** 0.5, rnd(N, seed=N + 3, scale=0.2)
        b[7] += 3.0       # a clear winner somewhere, a planted exact tie elsewhere (rows 0 / 1 below)
        for tag, mk in (('bf16', lambda t: t.to(DEV, torch.bfloat16)), ('fp32', lambda t: t.to(DEV))):
            Ad, Wd, bd = mk(A), mk(W), b.to(DEV)
            if M > 1 and N > 200:   # an exact tie: two identical weight rows with the same bias -> the lower index must win in both paths
                Wd[150] = Wd[40]
                bd[150] = bd[40] = 9.0
            lg = ops.gemm(Ad, Wd, bd, out_dtype=torch.float32)
            i0, p0 = ops.row_argmax_prob(lg)
            i1, p1 = ops.gemm_row_argmax_prob(Ad, Wd, bd)
            out.append(rec('gemm_row_stats[%s %dx%dx%d] ids identical' % (tag, M, N, K), float((i0 != i1).sum()), 0))
            out.append(rec('gemm_row_stats[%s %dx%dx%d] probabilities (relative)' % (tag, M, N, K), ((p0 - p1).abs() / p0).max().item(), 2e-6))
            if N > 200:
                out.append(rec('gemm_row_stats[%s %dx%dx%d] the planted tie goes to the lower index' % (tag, M, N, K), float((i1 != 40).sum()) if tag == 'fp32' else 0.0, 0))
        # bf16x3: split-pair activation rows against the [hi | hi | lo] image of the fp32 weight
        As, W3, bd = ops.split_bf16(A.to(DEV)), ops.split_weight3(W.to(DEV)), b.to(DEV)
        lg = ops.gemm(As, W3, bd, out_dtype=torch.float32, a_wrap=2 * K)
        i0, p0 = ops.row_argmax_prob(lg)
        i1, p1 = ops.gemm_row_argmax_prob(As, W3, bd, a_wrap=2 * K)
        # (the logits path may take the fused three-product kernel, whose summation order differs in the last bits: ids may differ only at a near-tie)
        lg_s = lg.sort(dim=-1, descending=True).values
        near = (lg_s[:, 0] - lg_s[:, 1]) < 1e-4
        out.append(rec('gemm_row_stats[bf16x3 %dx%dx%d] ids identical outside exact near-ties' % (M, N, K), float(((i0 != i1) & ~near).sum()), 0, '%d near-ties' % int(near.sum())))
        out.append(rec('gemm_row_stats[bf16x3 %dx%dx%d] probabilities (relative)' % (M, N, K), ((p0 - p1).abs() / p0).max().item(), 1e-4))
    return out


def check_mgp_greedy_fused(dtype_name='bf16', B=64):
    """MGPSTR.recognize with the wide heads decoding from the product's row statistics (greedy_fused, the default) against the logits + arg-max
    pass it replaces: full ViT-B with the full vocabularies (BPE 50 257, WordPiece 30 522 classes), batch 64 -- ids of all three granularities,
    the fused choice and the text identical, confidences within 1e-5."""
    c = R.cfg()
    sd = R.make_state_dict(c, seed=41)
    model = build(c, sd, ENGINES[dtype_name])
    img = rnd(B, 3, 32, 128, seed=78).clamp(-1, 1).to(DEV)
    model.greedy_fused = False
    ref = model.recognize(img)
    model.greedy_fused = True
    got = model.recognize(img)
    bad = {k: 0 for k in ('char_ids', 'bpe_ids', 'wp_ids', 'choice', 'char_text')}
    cerr = 0.0
    for g, r in zip(got, ref):
        for k in bad:
            bad[k] += 0 if g[k] == r[k] else 1
        cerr = max(cerr, max(abs(x - y) for x, y in zip(g['conf'], r['conf'])))
    out = [rec('mgp_greedy_fused[%s] %s identical over %d words' % (dtype_name, k, B), v, 0) for k, v in bad.items()]
    out.append(rec('mgp_greedy_fused[%s] confidences' % dtype_name, cerr, 1e-5))
    return out


def check_vit_attn_qkv(B=5):
    """omp_vit_attn_qkv (round 6: q | k | v as ONE token-major product, keys by strided DMA, the blocked V^T image built in LDS by the kernel) against
    the blocked-slab path it replaces (three projections with OMP_STORE_KBLK / OMP_STORE_VBLK epilogues + omp_vit_attn): the products are the same
    bits, the attention arithmetic is the same kernel body, so the encoder's token stream must be IDENTICAL bit for bit; and against the oracle's
    encoder under the bf16 gate.  Full ViT-B width, 2 blocks, 5 images (257 tokens: the ragged last key block, the padded keys)."""
    c = R.cfg(depth=2)
    sd = R.make_state_dict(c, seed=51)
    model = build(c, sd, torch.bfloat16)
    img = rnd(B, 3, 32, 128, seed=79).clamp(-1, 1)
    res = {}
    for on in (False, True):
        model.vit_qkv_fused = on
        x, _, T = model.encode(img.to(DEV))
        torch.cuda.synchronize()
        res[on] = x.float().cpu().clone()
    with torch.no_grad():
        ref = R.encoder(sd, c, img).reshape(res[True].shape)
    scale = ref.abs().max().item()
    return [rec('vit_attn_qkv: encoder tokens identical to the blocked-slab path, bit for bit', 0 if torch.equal(res[True], res[False]) else (res[True] - res[False]).abs().max().item() + 1e-9, 0),
            rec('vit_attn_qkv: encoder tokens vs oracle (relative)', (res[True] - ref).abs().max().item() / scale, 0.03),
            rec('vit_attn_qkv: the output is not trivially zero', 0 if res[True].abs().sum().item() > 0 else 1, 0)]
