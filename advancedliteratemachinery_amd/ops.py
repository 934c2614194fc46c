"""Thin tensor -> pointer wrappers over the C ABI (include/omp355.h).

PyTorch is used for device memory and streams only; every computation below happens inside
libomp355.so.  All wrappers launch on torch's CURRENT stream and never synchronise.
"""
import ctypes

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, OMP_BF16, OMP_BF16X2, OMP_F32  # noqa: F401

_DT = {torch.float32: OMP_F32, torch.bfloat16: OMP_BF16}
SPLIT = 'bf16x2'   # out_dtype of producers that write split-bf16 pair rows [hi | lo] (OMP_BF16X2): a bf16 tensor [rows, 2C]


def dt(t):
    try:
        return _DT[t if isinstance(t, torch.dtype) else t.dtype]
    except KeyError:
        raise TypeError('libomp355 supports float32/bfloat16 tensors only, got %s' % (t,))


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('libomp355 ops need device tensors (got a CPU tensor); there is no CPU fallback')
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _c(t, name):
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)
    return t


def layernorm(x, gamma, beta, out_dtype=None, out=None, out_f32=None, eps=1e-5, want_out=True):
    """x [rows, C] -> LN in `out_dtype` (default x.dtype) and/or fp32 copy."""
    _c(x, 'x')
    rows, C = x.numel() // x.shape[-1], x.shape[-1]
    out_dtype = out_dtype or x.dtype
    split = out_dtype == SPLIT
    if out is None and want_out:
        out = (torch.empty((rows, 2 * C), dtype=torch.bfloat16, device=x.device) if split
               else torch.empty(x.shape, dtype=out_dtype, device=x.device))
    rc = _lib.lib().omp_layernorm(ptr(x), dt(x), ptr(gamma), ptr(beta), ptr(out),
                                  OMP_BF16X2 if split else (dt(out) if out is not None else OMP_F32), ptr(out_f32), rows, C,
                                  float(eps), stream())
    _lib.check(rc, 'omp_layernorm')
    return out


def gemm(A, W, bias=None, residual=None, act=ACT_NONE, out=None, out_dtype=None, M=None, lda=None,
         ldw=None, ldc=None, N=None, K=None, bias_row=None, bias_row_stride=0, trans_rows=0, trans_ld=0,
         ln=None, ln_eps=1e-5, small_m=False, store_mode=0, kv=None, bias_along_m=False, out_noresidual=None, a_wrap=0):
    """out[M,N] = act(A[M,K] @ W[N,K]^T + bias) + residual.  A/W may be strided row views (lda/ldw).
    out_noresidual (optional, with a residual): also receives act(A W^T + bias) without the residual.
    a_wrap > 0: bf16x3 product -- A is a split-bf16 pair tensor [M, a_wrap] = [hi | lo], W the [N, K] = [hi | hi | lo] image of an
    fp32 weight (split_weight3), K = 3/2 a_wrap.  out_dtype=SPLIT: the result is written as split pairs [M, 2N]."""
    K = K or (W.shape[-1] if a_wrap else A.shape[-1])
    N = N or W.shape[0]
    M = M or A.numel() // A.shape[-1]
    lda = lda or A.stride(-2) if A.dim() > 1 else K
    ldw = ldw or W.stride(0)
    out_dtype = out_dtype or W.dtype
    split = out_dtype == SPLIT
    if out is None:
        if trans_rows:
            raise ValueError('trans_out needs a preallocated (zeroed) output')
        out = torch.empty((M, 2 * N if split else N), dtype=torch.bfloat16 if split else out_dtype, device=A.device)
    ldc = ldc or (out.stride(-2) if out.dim() > 1 and not store_mode else N)
    a = _lib.GemmArgs()
    a.A, a.lda, a.W, a.ldw = ptr(A), lda, ptr(W), ldw
    a.bias, a.bias_row, a.bias_row_stride = ptr(bias), ptr(bias_row), bias_row_stride
    a.residual, a.ldr = ptr(residual), (residual.stride(-2) if residual is not None else 0)
    a.C, a.ldc = ptr(out), ldc
    a.M, a.N, a.K = M, N, K
    a.dtype, a.out_dtype, a.act = dt(W), (OMP_BF16X2 if split else dt(out)), act
    a.a_wrap = int(a_wrap)
    if ln is not None:
        a.ln_gamma, a.ln_beta, a.ln_eps = ptr(ln[0]), ptr(ln[1]), float(ln_eps)
    a.small_m_splitk = 1 if small_m else 0
    a.store_mode, a.bias_along_m = store_mode, 1 if bias_along_m else 0
    if kv is not None:  # (images, tokens per image, padded tokens, heads, key block) of the blocked K / V^T slabs
        a.kv_images, a.kv_tokens, a.kv_mpad, a.kv_heads, a.kv_key_block = kv
    a.trans_out, a.trans_rows, a.trans_ld = (1 if trans_rows else 0), trans_rows, trans_ld
    if out_noresidual is not None:
        a.C2, a.ldc2 = ptr(out_noresidual), out_noresidual.stride(-2)
    rc = _lib.lib().omp_gemm_bias_act(ctypes.byref(a), stream())
    _lib.check(rc, 'omp_gemm_bias_act')
    return out


def gemm_choice(M, N, K, dtype=OMP_BF16, out_dtype=None, act=ACT_NONE, residual=False, bias=True, a_wrap=0, store_mode=0, kv=None,
                lda=None, ldw=None, ldc=None, two_destinations=False, ln=False, small_m=False, base=1 << 20):
    """Host logic only (no device, no launch): the kernel selector omp_gemm_bias_act would take for a product of this shape -- the
    dispatch table of csrc/gemm.hip as a function (tests/test_host_logic.py pins it).  Pointers are synthetic, 4 KB-aligned addresses:
    the library only tests them for null and alignment.  Returns the selector (csrc/omp355_debug.h) or raises on an argument error."""
    out_dtype = dtype if out_dtype is None else out_dtype
    a = _lib.GemmArgs()
    split = out_dtype == OMP_BF16X2
    ka = a_wrap if a_wrap else K
    a.A, a.lda, a.W, a.ldw = base, lda or ka, base + (1 << 30), ldw or K
    a.bias = base + (2 << 30) if bias else None
    a.residual, a.ldr = (base + (3 << 30), N) if residual else (None, 0)
    a.C, a.ldc = base + (4 << 30), ldc or (2 * N if split else N)
    a.M, a.N, a.K = M, N, K
    a.dtype, a.out_dtype, a.act, a.a_wrap = dtype, out_dtype, act, int(a_wrap)
    if ln:
        a.ln_gamma, a.ln_beta, a.ln_eps = base + (5 << 30), base + (6 << 30), 1e-5
    a.small_m_splitk = 1 if small_m else 0
    a.store_mode = store_mode
    if kv is not None:
        a.kv_images, a.kv_tokens, a.kv_mpad, a.kv_heads, a.kv_key_block = kv
    if two_destinations:
        a.C2, a.ldc2 = base + (7 << 30), N
    rc = _lib.lib().omp_debug_gemm_choice(ctypes.byref(a))
    if rc < 0:
        _lib.check(rc, 'omp_debug_gemm_choice')
    return rc


def swin_mlp_fused(x, ln_g, ln_b, wpack, b2, out=None, eps=1e-5):
    """x [M, C] bf16 or fp32 (fp32 residual stream) -> x + fc2(GELU(fc1(LN(x)))) in one launch, same type; wpack from
    model.packing.pack_mlp (bf16 matrix-core operands either way).  out may be x."""
    _c(x, 'x')
    if x.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError('swin_mlp_fused takes a bf16 or fp32 residual stream')
    M, C = x.numel() // x.shape[-1], x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    rc = _lib.lib().omp_swin_mlp_fused2(ptr(x), dt(x), C, ptr(ln_g), ptr(ln_b), float(eps), ptr(wpack), ptr(b2), ptr(out), C, M, C,
                                        wpack.shape[0] * 32, stream())
    _lib.check(rc, 'omp_swin_mlp_fused')
    return out


def patch_embed_ln(img, w, b, gamma, beta, out_dtype, eps=1e-5):
    _c(img, 'img')
    B, _, H, W = img.shape
    E = w.shape[0]
    Hp, Wp = (H + 3) // 4, (W + 3) // 4
    out = torch.empty((B, Hp * Wp, E), dtype=out_dtype, device=img.device)
    rc = _lib.lib().omp_patch_embed_ln(ptr(img), ptr(w), ptr(b), ptr(gamma), ptr(beta), ptr(out), dt(out),
                                       B, H, W, E, float(eps), stream())
    _lib.check(rc, 'omp_patch_embed_ln')
    return out, Hp, Wp


def swin_expand_bias(table):
    """relative_position_bias_table fp32 [169, nH] -> fp32 [nH, 64, 64] (bias / scale, -inf on padding keys), once per checkpoint."""
    nH = table.shape[1]
    out = torch.empty((nH, 64, 64), dtype=torch.float32, device=table.device)
    _lib.check(_lib.lib().omp_swin_expand_bias(ptr(_c(table, 'table')), nH, ptr(out), stream()), 'omp_swin_expand_bias')
    return out


def swin_window_attn(qkv, qkv_bias, table, B, H, W, C, nH, shift, out=None, window=7, bias_expanded=None, out_split=False):
    """out_split (fp32 qkv only): out is [B*H*W, 2C] bf16 split pairs [hi | lo] for the bf16x3 proj GEMM."""
    _c(qkv, 'qkv')
    if out is None:
        out = (torch.empty((B * H * W, 2 * C), dtype=torch.bfloat16, device=qkv.device) if out_split
               else torch.empty((B * H * W, C), dtype=qkv.dtype, device=qkv.device))
    rc = _lib.lib().omp_swin_window_attn2(ptr(qkv), ptr(qkv_bias), ptr(table), ptr(bias_expanded), ptr(out), dt(qkv),
                                          OMP_BF16X2 if out_split else dt(qkv), B, H, W, C, nH, window, shift, stream())
    _lib.check(rc, 'omp_swin_window_attn')
    return out


def swin_attn_block(x, ln_g, ln_b, qkv_w, qkv_b, bias_expanded, proj_w, proj_b, B, H, W, C, nH, shift, out=None, window=7, eps=1e-5):
    """x (fp32 [B*H*W, C]) -> x + proj(window attention(qkv(LayerNorm(x)))) in one launch (omp_swin_attn_block: C = 128, 4 heads);
    out defaults to x (in place)."""
    _c(x, 'x')
    _c(qkv_w, 'qkv_w')
    _c(proj_w, 'proj_w')
    if x.dtype != torch.float32 or qkv_w.dtype != torch.bfloat16 or proj_w.dtype != torch.bfloat16:
        raise TypeError('swin_attn_block: fp32 residual stream with bf16 weights')
    if x.numel() != B * H * W * C or tuple(qkv_w.shape) != (3 * C, C) or tuple(proj_w.shape) != (C, C) or bias_expanded.numel() != nH * 4096:
        raise ValueError('swin_attn_block: x [B*H*W, C], qkv_w [3C, C], proj_w [C, C], bias_expanded [nH, 64, 64] expected')
    out = x if out is None else _c(out, 'out')
    if out.dtype != torch.float32 or out.numel() != x.numel():
        raise ValueError('swin_attn_block: out must be fp32 with the shape of x')
    rc = _lib.lib().omp_swin_attn_block(ptr(x), ptr(out), ptr(ln_g), ptr(ln_b), float(eps), ptr(qkv_w), ptr(qkv_b), ptr(bias_expanded),
                                        ptr(proj_w), ptr(proj_b), B, H, W, C, nH, window, shift, stream())
    _lib.check(rc, 'omp_swin_attn_block')
    return out


def swin_attn_block_packed(x, ln_g, ln_b, wpack, qkv_b, bias_expanded, proj_b, B, H, W, C, nH, shift, out=None, window=7, eps=1e-5):
    """swin_attn_block for C = 256 with 8 heads: wpack = model.packing.pack_attn_block(qkv.weight, proj.weight, nH) (omp_swin_attn_block_packed)."""
    _c(x, 'x')
    if x.dtype != torch.float32 or wpack.dtype != torch.bfloat16:
        raise TypeError('swin_attn_block_packed: fp32 residual stream with a bf16 weight image')
    out = x if out is None else out
    rc = _lib.lib().omp_swin_attn_block_packed(ptr(x), ptr(out), ptr(ln_g), ptr(ln_b), float(eps), ptr(wpack), ptr(qkv_b), ptr(bias_expanded),
                                               ptr(proj_b), B, H, W, C, nH, window, shift, stream())
    _lib.check(rc, 'omp_swin_attn_block_packed')
    return out


def patch_merge_gather_ln(x, gamma, beta, B, H, W, C, eps=1e-5, out_dtype=None):
    """out_dtype: x.dtype (default), torch.bfloat16 for an fp32 x, or SPLIT (split pairs [rows, 8C])."""
    _c(x, 'x')
    H2, W2 = (H + 1) // 2, (W + 1) // 2
    out_dtype = out_dtype or x.dtype
    split = out_dtype == SPLIT
    out = torch.empty((B * H2 * W2, (8 if split else 4) * C), dtype=torch.bfloat16 if split else out_dtype, device=x.device)
    rc = _lib.lib().omp_patch_merge_gather_ln2(ptr(x), dt(x), ptr(gamma), ptr(beta), ptr(out), OMP_BF16X2 if split else dt(out),
                                               B, H, W, C, float(eps), stream())
    _lib.check(rc, 'omp_patch_merge_gather_ln')
    return out, H2, W2


def split_bf16(x, triple=False, out=None):
    """fp32 [rows, C] -> bf16 split pairs [rows, 2C] = [hi | lo] (triple: [rows, 3C] = [hi | hi | lo]); omp_split_bf16."""
    if x.dtype != torch.float32:
        raise TypeError('split_bf16 takes fp32 rows')
    rows, C = x.numel() // x.shape[-1], x.shape[-1]
    n = 3 if triple else 2
    if out is None:
        out = torch.empty((rows, n * C), dtype=torch.bfloat16, device=x.device)
    rc = _lib.lib().omp_split_bf16(ptr(x), x.stride(-2) if x.dim() > 1 else C, ptr(out), out.stride(-2), rows, C, 1 if triple else 0, stream())
    _lib.check(rc, 'omp_split_bf16')
    return out


def split_weight3(w):
    """fp32 weight [N, K] -> bf16 [N, 3K] = [w_hi | w_hi | w_lo]: the W-side image of a bf16x3 product (once per checkpoint)."""
    w = w.detach().float()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, hi, lo], dim=1).contiguous()


def split_weight2(w):
    """fp32 weight [N, K] -> bf16 [N, 2K] = [w_hi | w_lo]: the A-side operand of a bf16x3 product with swapped operands."""
    w = w.detach().float()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=1).contiguous()


def fpn_fuse(l2, l3, l4, l5, B, sizes, stride):
    (h2, w2), (h3, w3), (h4, w4), (h5, w5) = sizes
    ho, wo = (h3 + stride - 1) // stride, (w3 + stride - 1) // stride
    out = torch.empty((B * ho * wo, 1024), dtype=l2.dtype, device=l2.device)
    rc = _lib.lib().omp_fpn_fuse(ptr(l2), ptr(l3), ptr(l4), ptr(l5), ptr(out), dt(l2), B, h2, w2, h3, w3,
                                 h4, w4, h5, w5, stride, stream())
    _lib.check(rc, 'omp_fpn_fuse')
    return out, ho, wo


def mask_nearest(mask_u8, h, w):
    """uint8 [B,H,W] -> uint8 [B,h,w]: F.interpolate(..., mode='nearest') of the reference (swin_transformer.py:621)."""
    _c(mask_u8, 'mask')
    B, H, W = mask_u8.shape
    out = torch.empty((B, h, w), dtype=torch.uint8, device=mask_u8.device)
    rc = _lib.lib().omp_mask_nearest(ptr(mask_u8), ptr(out), B, H, W, h, w, stream())
    _lib.check(rc, 'omp_mask_nearest')
    return out


def sine_posembed(mask_u8, npf, out_dtype, temperature=10000.0):
    _c(mask_u8, 'mask')
    B, h, w = mask_u8.shape
    out = torch.empty((B, h * w, 2 * npf), dtype=out_dtype, device=mask_u8.device)
    rc = _lib.lib().omp_sine_posembed(ptr(mask_u8), ptr(out), dt(out), B, h, w, npf, float(temperature),
                                      stream())
    _lib.check(rc, 'omp_sine_posembed')
    return out


def dec_embed_ln(seq, d_pos, word, postab, gamma, beta, x=None, y=None, eps=1e-5):
    R, d = seq.shape[0], word.shape[1]
    rc = _lib.lib().omp_dec_embed_ln(ptr(seq), seq.stride(0), ptr(d_pos), ptr(word), ptr(postab), ptr(gamma),
                                     ptr(beta), ptr(x), ptr(y), dt(y) if y is not None else OMP_F32, R, d,
                                     float(eps), stream())
    _lib.check(rc, 'omp_dec_embed_ln')


def dec_self_attn_step(qkv, kcache, vcache, out, d_pos, nH):
    R, d3 = qkv.shape
    d = d3 // 3
    rc = _lib.lib().omp_dec_self_attn_step(ptr(qkv), ptr(kcache), ptr(vcache), ptr(out), ptr(d_pos), dt(qkv),
                                           R, nH, d, kcache.shape[1], stream())
    _lib.check(rc, 'omp_dec_self_attn_step')


def dec_cross_attn_step(q, K, Vt, img_stride, Mpad, key_mask, groups, n_groups, q_tiles, partial, out, M, nH, n_split):
    """K / Vt in q's dtype: the bf16 / fp32 slabs; fp32 q with bf16 K / Vt: SPLIT-PLANE slabs (OMP_BF16X2: 32-key blocks of
    [hi plane | lo plane], img_stride = nH * Mpad * 128 bf16 elements; fp32 out)."""
    R = q.shape[0]
    split = q.dtype == torch.float32 and K.dtype == torch.bfloat16
    if split and (Vt.dtype != torch.bfloat16 or out.dtype != torch.float32):
        raise TypeError('split-plane slabs: bf16 K and V^T planes, fp32 q and out')
    if not split and (K.dtype != q.dtype or Vt.dtype != q.dtype or out.dtype != q.dtype):
        raise TypeError('dec_cross_attn_step: q, K, V^T and out share one dtype (or fp32 q / out over split-plane bf16 slabs)')
    rc = _lib.lib().omp_dec_cross_attn_step(ptr(q), q.stride(0), ptr(K), ptr(Vt), img_stride, Mpad, ptr(key_mask),
                                            ptr(groups), n_groups, q_tiles, R, ptr(partial), ptr(out), out.stride(0),
                                            OMP_BF16X2 if split else dt(q), M, nH, n_split, stream())
    _lib.check(rc, 'omp_dec_cross_attn_step')


def head_sample(logits, cfg, seq, probs, finished, lengths, d_pos, advance=True):
    R = logits.shape[0]
    rc = _lib.lib().omp_head_softmax_mask_argmax(ptr(logits), logits.stride(0), R, ctypes.byref(cfg), ptr(seq),
                                                 ptr(probs), seq.stride(0), ptr(finished), ptr(lengths),
                                                 ptr(d_pos), 1 if advance else 0, stream())
    _lib.check(rc, 'omp_head_softmax_mask_argmax')


def pack_spotting(points, poly, rec, rprob, counts, N, rec_len):
    """Decoded rows (device int32 [R,2] / [R,>=32] / [R,>=rec_len], fp32 [R,>=rec_len]; rows sorted by image, counts per image)
    -> (ids int32 [B, N, 34 + rec_len], probs fp32 [B, N, rec_len], n_inst int32 [B]) in ONE launch (omp_pack_spotting)."""
    dev = points.device
    B = len(counts)
    c = torch.tensor(counts, dtype=torch.int32)
    row0 = (torch.cumsum(c, 0) - c).to(torch.int32)
    cd, r0 = torch.minimum(c, torch.tensor(N, dtype=torch.int32)).to(dev, non_blocking=True), row0.to(dev, non_blocking=True)
    ids = torch.empty((B, N, 34 + rec_len), dtype=torch.int32, device=dev)
    probs = torch.empty((B, N, rec_len), dtype=torch.float32, device=dev)
    for t in (points, poly, rec, rprob):
        if t.stride(-1) != 1:
            raise ValueError('pack_spotting: rows must be contiguous along the last dimension')
    rc = _lib.lib().omp_pack_spotting(ptr(points), ptr(poly), poly.stride(0), ptr(rec), rec.stride(0), ptr(rprob), rprob.stride(0),
                                      ptr(r0), ptr(cd), B, N, rec_len, ptr(ids), ptr(probs), stream())
    _lib.check(rc, 'omp_pack_spotting')
    return ids, probs, cd


def vit_patch_embed(img, w, bias, cls, pos, out_dtype):
    """img [B,3,H,W] fp32 -> tokens [B, T, E] with T = (H/4)*(W/4) + 1 (token 0 = cls), pos_embed added."""
    _c(img, 'img')
    B, _, H, W = img.shape
    E = w.shape[0]
    T = (H // 4) * (W // 4) + 1
    out = torch.empty((B, T, E), dtype=out_dtype, device=img.device)
    rc = _lib.lib().omp_vit_patch_embed(ptr(img), ptr(w), ptr(bias), ptr(cls), ptr(pos), ptr(out), dt(out), B, H, W, E, stream())
    _lib.check(rc, 'omp_vit_patch_embed')
    return out


def vit_attn(q, K, Vt, out, B, T, nH, Mpad):
    """bf16 ViT self-attention on one layer's blocked slabs: q [B*T, nH*64], K [B, nH, Mpad, 64], Vt [B, nH, Mpad/32, 64, 32]
    -> out [B*T, nH*64] (all heads, all images, one launch)."""
    rc = _lib.lib().omp_vit_attn(ptr(q), q.stride(0), ptr(K), ptr(Vt), Mpad, ptr(out), out.stride(0), dt(q), B, T, nH, stream())
    _lib.check(rc, 'omp_vit_attn')
    return out


def vit_attn_qkv(qkv, out, B, T, nH):
    """bf16 ViT self-attention on the token-major fused projection qkv [B*T, 3*nH*64] = [q | k | v] -> out [B*T, nH*64]: one workgroup per
    (image, head), keys by strided DMA, the blocked V^T image built in LDS by the kernel (omp_vit_attn_qkv)."""
    rc = _lib.lib().omp_vit_attn_qkv(ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), dt(qkv), B, T, nH, stream())
    _lib.check(rc, 'omp_vit_attn_qkv')
    return out


def a3_pool(sel, feat, B, T, S, want_attn=True):
    """sel fp32 [B*T, >=S], feat [B*T, C] -> (pooled fp32 [B*S, C], maps fp32 [B, S, T] or None)."""
    C = feat.shape[-1]
    pooled = torch.empty((B * S, C), dtype=torch.float32, device=feat.device)
    attn = torch.empty((B, S, T), dtype=torch.float32, device=feat.device) if want_attn else None
    rc = _lib.lib().omp_a3_pool(ptr(sel), sel.stride(0), ptr(feat), dt(feat), ptr(pooled), ptr(attn), B, T, S, C, stream())
    _lib.check(rc, 'omp_a3_pool')
    return pooled, attn


def row_argmax_prob(logits):
    """logits fp32 [R, V] -> (ids int32 [R], prob fp32 [R]): greedy id and its softmax probability."""
    R, V = logits.shape
    ids = torch.empty(R, dtype=torch.int32, device=logits.device)
    prob = torch.empty(R, dtype=torch.float32, device=logits.device)
    rc = _lib.lib().omp_row_argmax_prob(ptr(logits), logits.stride(0), R, V, ptr(ids), ptr(prob), stream())
    _lib.check(rc, 'omp_row_argmax_prob')
    return ids, prob


def row_argmax_prob_2d(logits, B, S):
    """row_argmax_prob of logits [B * S, V] as ([B, S] ids, [B, S] prob)"""
    i, p = row_argmax_prob(logits.reshape(B * S, -1))
    return i.view(B, S), p.view(B, S)


def gemm_row_argmax_prob(A, W, bias=None, a_wrap=0):
    """(ids int32 [M], prob fp32 [M]) of the rows of A W^T + bias WITHOUT materialising the [M, N] logits: the product stores per
    (row, 128-column tile) its maximum, arg-max and sum of exponentials (omp_gemm_bias_act, OMP_STORE_ROWSTAT), omp_row_stat_merge folds
    the tiles.  The same values as row_argmax_prob(gemm(A, W, bias, out_dtype=float32)) (identical product bits; the probability within
    the rounding of a different summation order).  a_wrap: a split-pair A against the [hi | hi | lo] image of an fp32 weight (bf16x3)."""
    M = A.numel() // A.shape[-1]
    N = W.shape[0]
    nt = 2 * ((N + 127) // 128)        # one record per 64-column half of a 128-column tile
    stats = torch.empty((M, nt, 4), dtype=torch.float32, device=A.device)
    gemm(A, W, bias, out=stats, out_dtype=torch.float32, store_mode=_lib.STORE_ROWSTAT, ldc=N, a_wrap=a_wrap)
    ids = torch.empty(M, dtype=torch.int32, device=A.device)
    prob = torch.empty(M, dtype=torch.float32, device=A.device)
    _lib.check(_lib.lib().omp_row_stat_merge(ptr(stats), M, nt, ptr(ids), ptr(prob), stream()), 'omp_row_stat_merge')
    return ids, prob


class Context(object):
    """An omp_ctx: independent kernel selectors, captured decoder graphs and measurement brackets (include/omp355.h).
    `with ctx:` makes it current for the calling thread and restores the previous one; pipeline lane threads inherit the
    context that was current where their LanePool was created."""

    def __init__(self):
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().omp_ctx_create(ctypes.byref(h)), 'omp_ctx_create')
        self.handle = h.value
        self._prev = []
        from .model.transformer import _GraphSlots
        _GraphSlots._dead_ctx.discard(self.handle)   # the allocator may hand a destroyed context's address out again

    def make_current(self):
        _lib.check(_lib.lib().omp_ctx_make_current(ctypes.c_void_p(self.handle)), 'omp_ctx_make_current')

    def __enter__(self):
        self._prev.append(current_context_handle())
        self.make_current()
        return self

    def __exit__(self, *exc):
        _lib.check(_lib.lib().omp_ctx_make_current(ctypes.c_void_p(self._prev.pop())), 'omp_ctx_make_current')
        return False

    def destroy(self):
        if self.handle:
            from .model.transformer import _GraphSlots
            _GraphSlots._dead_ctx.add(self.handle)   # graphs captured in it die with it: never reset them through it again
            _lib.check(_lib.lib().omp_ctx_destroy(ctypes.c_void_p(self.handle)), 'omp_ctx_destroy')
            self.handle = None


def current_context_handle():
    """raw omp_ctx* of the calling thread (the process default context unless one was made current)"""
    return _lib.lib().omp_ctx_current()


def make_context_current(handle):
    _lib.check(_lib.lib().omp_ctx_make_current(ctypes.c_void_p(handle)), 'omp_ctx_make_current')


class capture_gate(object):
    """`with ops.capture_gate():` -- bracket for NON-BLOCKING event calls on a stream another host thread drives through omp_decoder_run
    (omp_capture_gate_enter / _leave, include/omp355.h: no graph capture begins or is in progress inside the bracket)."""

    def __enter__(self):
        _lib.lib().omp_capture_gate_enter()
        return self

    def __exit__(self, *exc):
        _lib.lib().omp_capture_gate_leave()
        return False


def cu_mask_words(n_per_xcd, total_cus=256, n_xcd=8, complement=False):
    """CU mask with `n_per_xcd` compute units on every XCD: bit i is set iff (i mod 32) < n_per_xcd ... balanced under both
    numberings a runtime may use for the mask (XCD-major: xcd = i / 32; interleaved: xcd = i mod 8) when n_per_xcd is a
    multiple of 8.  complement=True returns the other CUs."""
    per = total_cus // n_xcd
    words = [0] * ((total_cus + 31) // 32)
    for i in range(total_cus):
        inside = (i % per) < n_per_xcd
        if inside != complement:
            words[i // 32] |= 1 << (i % 32)
    return words


def masked_stream(words, device=None):
    """torch.cuda.ExternalStream on the CUs of `words` (omp_stream_create_cu_mask); the HIP stream lives as long as
    the process (a handful per lane)."""
    arr = (ctypes.c_uint32 * len(words))(*words)
    out = ctypes.c_void_p()
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        rc = _lib.lib().omp_stream_create_cu_mask(ctypes.cast(arr, ctypes.c_void_p), len(words), ctypes.byref(out))
        _lib.check(rc, 'omp_stream_create_cu_mask')
        return torch.cuda.ExternalStream(out.value, device=device)


def where_probe(n_workgroups):
    """(xcc_id, hw_id) int32 [n_workgroups, 2] of a short probe grid on the current stream."""
    out = torch.zeros((n_workgroups, 2), dtype=torch.int32, device='cuda')
    _lib.check(_lib.lib().omp_debug_where(ptr(out), n_workgroups, stream()), 'omp_debug_where')
    return out


def force_gemm_kernel(which):
    """debug/testing: 0 auto, 3 row-streaming, 4 split-K small-M, 5 DMA 128x128 (2 stages), 6 DMA 64x64 ring, 9 gemm_256 (256x256 tiles,
    eight waves), 10 gemm_4w (four waves), 16 gemm_4w_r (four waves, weights streamed into registers), 20 gemm_4w_p (16 made persistent,
    register-only epilogue); the ablation / trace selectors are listed in csrc/omp355_debug.h."""
    _lib.check(_lib.lib().omp_debug_force_gemm_kernel(which), 'omp_debug_force_gemm_kernel')


def cross_q4(on):
    """debug/testing: 1 = LDS-ring cross-attention kernel for 33..64 rows per image (default: 64-key chunks on bf16 / fp32 slabs, three
    one-block stages on split-plane slabs), 2 = the ring consumed one 32-key block per step, 4 = chunks with temporal loads (split
    planes: eight stages in 64-key chunks), 5 / 6 = split planes with four / two stages, 0 = register-streaming kernel."""
    _lib.check(_lib.lib().omp_debug_cross_q4(int(on)), 'omp_debug_cross_q4')


def rows_tile_choice(R, mid=False):
    """host logic (no launch, no device): rows per workgroup a decoder row-owner chain launch of R rows takes (csrc/dec_rows.hip rows_rtt)."""
    rc = _lib.lib().omp_debug_rows_tile_choice(int(R), 1 if mid else 0)
    if rc < 0:
        _lib.check(rc, 'omp_debug_rows_tile_choice')
    return rc


def rows_tile(rtt):
    """debug/testing: rows per workgroup of the decoder row-owner chains (bf16 engine) = 16 x rtt; 0 = chosen from the launch's row count
    (csrc/dec_rows.hip rows_rtt), 2..5 forced."""
    _lib.check(_lib.lib().omp_debug_rows_tile(int(rtt)), 'omp_debug_rows_tile')


def dec_fused(mode):
    """debug/testing: 0 = fused few-row decoder step kernels where they apply (default), 1 = one launch per op everywhere."""
    _lib.check(_lib.lib().omp_debug_dec_fused(int(mode)), 'omp_debug_dec_fused')


def swin_attn_impl(which):
    """debug/testing: 0 = matrix-core window attention (default), 1 = scalar cross-check kernel, 2 = matrix cores with table lookups,
    3 = fp32 matrix cores also for split-pair output (the parity engine's call otherwise runs three bf16 products of split operands),
    4 = that split-product kernel also for fp32 output."""
    _lib.check(_lib.lib().omp_debug_swin_attn_impl(which), 'omp_debug_swin_attn_impl')


def swin_mlp_variant(v):
    """debug/testing: alternative (rows per wave, waves per workgroup, ring depth) instantiations of the fused MLP kernel."""
    _lib.check(_lib.lib().omp_debug_swin_mlp_variant(int(v)), 'omp_debug_swin_mlp_variant')


def kv_project_rows(rows, wstream, wave_stride, bias, out, B, M, Mpad, n_slabs, vt):
    """omp_kv_project_rows: the cross-attention memory projection of all (decoder, layer) slabs in one row-owner launch (csrc/kv_rows.hip).
    rows [B * M, 512] bf16; wstream / wave_stride from model/packing.py::pack_kv_rows_k / _v; out: the K (vt False) or V^T (vt True) slab tensor."""
    _c(rows, "rows"); _c(bias, "bias"); _c(out, "out")
    if rows.dtype != torch.bfloat16 or out.dtype != torch.bfloat16 or rows.shape != (B * M, 512) or not rows.is_contiguous():
        raise ValueError('kv_project_rows: [B * M, 512] contiguous bf16 rows and bf16 slabs')
    _lib.check(_lib.lib().omp_kv_project_rows(ptr(rows), ptr(wstream), int(wave_stride), ptr(bias), ptr(out), int(B), int(M), int(Mpad), int(n_slabs),
                                              1 if vt else 0, stream()), 'omp_kv_project_rows')
    return out


def dec_rows_mid(att, x, wstream, wave_stride, out_b, ln_g, ln_b, qbias_tab, d_pos, q=None, eps=1e-5, x3=False, xcd_mask=0):
    """x += att Wo^T + bo;  q = bf16(LayerNorm(x) Wq^T + qbias_tab[*d_pos])  -- one launch, a workgroup owns 80 rows (include/omp355.h).
    x3: the parity engine's chain -- att as split pairs bf16 [R, 1024], q fp32, split weight stream, 48 rows per workgroup."""
    R = x.shape[0]
    if q is None:
        q = torch.empty((R, 512), dtype=torch.float32 if x3 else torch.bfloat16, device=x.device)
    a = _lib.DecRowsArgs()
    a.x3, a.xcd_mask = 1 if x3 else 0, int(xcd_mask)
    a.R, a.eps, a.d_pos, a.x, a.att = R, float(eps), ptr(d_pos), ptr(_c(x, 'x')), ptr(_c(att, 'att'))
    a.wstream, a.wave_stride = ptr(wstream), int(wave_stride)
    a.out_b, a.ln_g, a.ln_b, a.qbias_tab, a.q = ptr(out_b), ptr(ln_g), ptr(ln_b), ptr(qbias_tab), ptr(q)
    _lib.check(_lib.lib().omp_dec_rows_mid(ctypes.byref(a), stream()), 'omp_dec_rows_mid')
    return q


def dec_rows_ffn(x, wstream, wave_stride, d_pos, lnt_g, lnt_b, att=None, out_b=None, ln_g=None, ln_b=None, ff1_b=None, ff2_b=None,
                 embed=None, bias_tab=None, qkv=None, head_b=None, logits=None, vocab=0, eps=1e-5, x3=False, xcd_mask=0):
    """The chain behind the cross-attention (or, embed=(seq, word_emb, pos_tab, emb_g, emb_b), the embedding of layer 0) and its tail:
    bias_tab given -> the next layer's q | k | v (bf16 [R, 1536]); head_b=(b0, b1, b2) -> the prediction head's logits (fp32 [R, vocab])."""
    R = x.shape[0]
    a = _lib.DecRowsArgs()
    a.x3, a.xcd_mask = 1 if x3 else 0, int(xcd_mask)
    a.R, a.eps, a.d_pos, a.x = R, float(eps), ptr(d_pos), ptr(_c(x, 'x'))
    a.wstream, a.wave_stride = ptr(wstream), int(wave_stride)
    a.lnt_g, a.lnt_b = ptr(lnt_g), ptr(lnt_b)
    if embed is not None:
        seq, word, pos_tab, eg, eb = embed
        a.prologue, a.seq, a.seq_ld, a.word_emb, a.pos_tab, a.emb_g, a.emb_b = 1, ptr(seq), seq.stride(0), ptr(word), ptr(pos_tab), ptr(eg), ptr(eb)
    else:
        a.prologue, a.att = 0, ptr(_c(att, 'att'))
        a.out_b, a.ln_g, a.ln_b, a.ff1_b, a.ff2_b = ptr(out_b), ptr(ln_g), ptr(ln_b), ptr(ff1_b), ptr(ff2_b)
    if head_b is None:
        if qkv is None:
            qkv = torch.empty((R, 1536), dtype=torch.float32 if x3 else torch.bfloat16, device=x.device)
        a.tail, a.bias_tab, a.qkv = 0, ptr(bias_tab), ptr(qkv)
        out = qkv
    else:
        if logits is None:
            logits = torch.empty((R, vocab), dtype=torch.float32, device=x.device)
        a.tail, a.h0_b, a.h1_b, a.h2_b, a.logits, a.vocab = 1, ptr(head_b[0]), ptr(head_b[1]), ptr(head_b[2]), ptr(logits), int(vocab)
        out = logits
    _lib.check(_lib.lib().omp_dec_rows_ffn(ctypes.byref(a), stream()), 'omp_dec_rows_ffn')
    return out


def swin_rows_qkv(x, n1, qkv_b, wstream, wave_stride, qkv=None, eps=1e-5, x3=False):
    """qkv = bf16(LayerNorm(x; n1) Wqkv^T + bqkv) for x [M, 512] fp32 (Swin-B stage 2, the stage's first block): omp_swin_rows_block mode 0.
    x3: the parity engine's chain (fp32 qkv)."""
    M = x.numel() // 512
    if qkv is None:
        qkv = torch.empty((M, 1536), dtype=torch.float32 if x3 else torch.bfloat16, device=x.device)
    a = _lib.SwinRowsArgs()
    a.x3 = 1 if x3 else 0
    a.M, a.eps, a.mode, a.x, a.qkv = M, float(eps), 0, ptr(_c(x, 'x')), ptr(qkv)
    a.wstream, a.wave_stride = ptr(wstream), int(wave_stride)
    a.n1_g, a.n1_b, a.qkv_b = ptr(n1[0]), ptr(n1[1]), ptr(qkv_b)
    _lib.check(_lib.lib().omp_swin_rows_block(ctypes.byref(a), stream()), 'omp_swin_rows_block')
    return qkv


def swin_rows_block(x, att, wstream, wave_stride, proj_b, n2, fc1_b, fc2_b, next_n1=None, next_qkv_b=None, qkv=None, eps=1e-5, x3=False):
    """x += att Wproj^T + bproj; x += fc2(GELU(fc1(LN(x; n2)))) in place on the fp32 residual stream [M, 512]; with next_n1 / next_qkv_b also
    the next block's qkv = bf16(LN(x; next_n1) Wqkv'^T + b') -> returned (else None).  omp_swin_rows_block mode 1."""
    if x.dtype != torch.float32 or x.shape[-1] != 512:
        raise TypeError('swin_rows_block takes the fp32 residual stream [M, 512]')
    M = x.numel() // 512
    a = _lib.SwinRowsArgs()
    a.x3 = 1 if x3 else 0
    a.M, a.eps, a.mode, a.x, a.att = M, float(eps), 1, ptr(_c(x, 'x')), ptr(_c(att, 'att'))
    a.wstream, a.wave_stride = ptr(wstream), int(wave_stride)
    a.proj_b, a.n2_g, a.n2_b, a.fc1_b, a.fc2_b = ptr(proj_b), ptr(n2[0]), ptr(n2[1]), ptr(fc1_b), ptr(fc2_b)
    if next_n1 is not None:
        if qkv is None:
            qkv = torch.empty((M, 1536), dtype=torch.float32 if x3 else torch.bfloat16, device=x.device)
        a.n1_g, a.n1_b, a.qkv_b, a.qkv = ptr(next_n1[0]), ptr(next_n1[1]), ptr(next_qkv_b), ptr(qkv)
    else:
        qkv = None
    _lib.check(_lib.lib().omp_swin_rows_block(ctypes.byref(a), stream()), 'omp_swin_rows_block')
    return qkv
