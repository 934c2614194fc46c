"""TEST DOUBLE (never on the product path): the libomp355 wrappers used by model/mgp_str.py, restated in plain
torch on the CPU from the semantics documented in include/omp355.h.  tests/test_mgp_host_flow.py swaps it in for
`ops` to exercise the HOST orchestration (argument order, shapes, slab geometry, row groups) without a GPU; the
kernels themselves are only ever validated on the MI355X (tests/test_gpu_*.py)."""
import torch
import torch.nn.functional as F

ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
STORE_PLAIN, STORE_KBLK, STORE_VBLK = 0, 2, 3


def layernorm(x, gamma, beta, out_dtype=None, out=None, out_f32=None, eps=1e-5, want_out=True):
    y = F.layer_norm(x.float(), (x.shape[-1],), gamma, beta, eps)
    if out is not None:
        out.copy_(y.to(out.dtype))
        return out
    return y.to(out_dtype or x.dtype)


def _slot(kl):
    return ((kl & 15) >> 2) * 8 + (kl >> 4) * 4 + (kl & 3)


def gemm(A, W, bias=None, residual=None, act=ACT_NONE, out=None, out_dtype=None, M=None, N=None, K=None,
         store_mode=0, kv=None, bias_along_m=False, **_):
    y = A.float() @ W.float().t()
    if bias is not None:
        y = y + (bias[:, None] if bias_along_m else bias)
    y = F.gelu(y) if act == ACT_GELU else (F.relu(y) if act == ACT_RELU else y)
    if store_mode == STORE_KBLK:
        Bn, T, Mpad, nH, KB = kv
        nl = y.shape[1] // (nH * 64)
        out[:, :, :, :T] = y.reshape(Bn, T, nl, nH, 64).permute(2, 0, 3, 1, 4).to(out.dtype)
        return out
    if store_mode == STORE_VBLK:            # y[m = (slab, head, dim), n = (image, token)]
        Bn, T, Mpad, nH, KB = kv
        nl = y.shape[0] // (nH * 64)
        v = y.reshape(nl, nH, 64, Bn, T).permute(0, 3, 1, 4, 2)            # [nl][B][nH][T][64]
        for t in range(T):
            blk, kl = t // KB, t % KB
            pos = _slot(kl) if KB == 32 else kl
            out[:, :, :, blk, :, pos] = v[:, :, :, t].to(out.dtype)
        return out
    if residual is not None:
        y = y + residual.float()
    if out is not None:
        out.copy_(y.to(out.dtype))
        return out
    return y.to(out_dtype or W.dtype)


def dec_cross_attn_step(q, K, Vt, img_stride, Mpad, key_mask, groups, n_groups, q_tiles, partial, out, M, nH, n_split):
    KB = Vt.shape[-1]
    assert img_stride == nH * Mpad * 64 and K.shape[-2] == Mpad and Vt.shape[-3] * KB == Mpad
    for (row0, nrows, img) in groups.tolist()[:n_groups]:
        assert nrows <= 16 * q_tiles
        kk = K[img, :, :M].float()                                          # [nH][M][64]
        vb = Vt[img].float()                                                # [nH][Mpad/KB][64][KB]
        if KB == 32:
            nat = torch.empty_like(vb)
            kl = torch.arange(32)
            nat[..., kl] = vb[..., _slot(kl)]
            vb = nat
        vv = vb.permute(0, 1, 3, 2).reshape(nH, Mpad, 64)[:, :M]
        qq = q[row0:row0 + nrows].float().reshape(nrows, nH, 64).permute(1, 0, 2) * 0.125
        att = (qq @ kk.transpose(-2, -1)).softmax(-1)
        out[row0:row0 + nrows] = (att @ vv).permute(1, 0, 2).reshape(nrows, nH * 64).to(out.dtype)


def vit_attn(q, K, Vt, out, B, T, nH, Mpad):
    """test double of omp_vit_attn: same slabs, all T queries of every image at once"""
    groups = torch.tensor([(b * T, T, b) for b in range(B)], dtype=torch.int32)
    dec_cross_attn_step(q, K, Vt, nH * Mpad * 64, Mpad, None, groups, B, (T + 15) // 16, None, out, T, nH, 1)
    return out


def vit_patch_embed(img, w, bias, cls, pos, out_dtype):
    B, E = img.shape[0], w.shape[0]
    x = F.conv2d(img, w.reshape(E, 3, 4, 4), bias, stride=4).flatten(2).transpose(1, 2)
    return (torch.cat((cls.reshape(1, 1, E).expand(B, -1, -1), x), 1) + pos).to(out_dtype)


def a3_pool(sel, feat, B, T, S, want_attn=True):
    maps = F.softmax(sel[:, :S].reshape(B, T, S).transpose(1, 2), dim=-1)
    pooled = torch.einsum('bsi,bid->bsd', maps, feat.float().reshape(B, T, -1)).reshape(B * S, -1)
    return pooled, (maps if want_attn else None)


def row_argmax_prob(logits):
    p, i = F.softmax(logits, dim=1).max(dim=1)
    return i.int(), p


def row_argmax_prob_2d(logits, B, S):
    i, p = row_argmax_prob(logits.reshape(B * S, -1))
    return i.view(B, S), p.view(B, S)


def gemm_row_argmax_prob(A, W, bias=None, a_wrap=0):
    """include/omp355.h: OMP_STORE_ROWSTAT + omp_row_stat_merge == the greedy id / probability of the rows of A W^T + bias"""
    return row_argmax_prob(gemm(A, W, bias, out_dtype=torch.float32, a_wrap=a_wrap))


def vit_attn_qkv(qkv, out, B, T, nH):
    """test double of omp_vit_attn_qkv: softmax(q k^T / 8) v per (image, head) on the token-major fused projection [B*T, 3*nH*64]"""
    E = nH * 64
    x = qkv.float().reshape(B, T, 3, nH, 64)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    att = F.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v
    out.copy_(att.permute(0, 2, 1, 3).reshape(B * T, E).to(out.dtype))
    return out
