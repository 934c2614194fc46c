"""Weight images for the fused Swin MLP kernel (csrc/mlp.hip): one-off re-layout of fc1 / fc2 at engine build time.

Pure data movement (index permutations of the checkpoint tensors); no arithmetic of the hot path lives here.

The kernel keeps a wave's LayerNorm'ed rows in registers as matrix-core B operands and streams the weights
through LDS in SUB-CHUNKS of 32 hidden units.  Sub-chunk `hc` is one contiguous block

    [ W1 image : C*64 B ][ W2 image : C*64 B ][ b1 slice : 128 B ][ pad to 1 KB ]          (block = C*128 + 1024 B)

copied to LDS linearly by DMA (global_load_lds_dwordx4), so all layout work happens HERE:

  * W1 image = rows 32*hc .. 32*hc+31 of fc1.weight [4C, C] as C/64 K-tiles of [32 rows][128 B]; inside a row the
    16-byte chunk c sits in slot c ^ (row & 7), which makes the kernel's fragment ds_read_b128s conflict-free
    (same swizzle as gemm_dma, there applied to the DMA source address);
  * W2 image = for every output feature n the 32 hidden units of the sub-chunk in MATRIX-CORE ORDER:
    position 8g + 4h + r  <-  hidden 32*hc + 16h + 4g + r   (g = 0..3, h = 0..1, r = 0..3).
    The first product delivers GELU(fc1) as accumulators in which lane group g holds hidden 4g..4g+3 of both
    16-wide tiles h = 0, 1; in this order those 8 values ARE the B operand fragment of the second product, so the
    hidden activations never leave registers.  The four 16-byte chunks of a 64-byte row are stored in slot
    g ^ f(n & 15), f(i) = (-(i >> 2)) & 3 (conflict-free ds_read_b128 over 16 rows x 64 B);
  * b1 slice = fc1.bias[32*hc .. 32*hc+31] as fp32.
"""
import torch


def mlp_block_bytes(C):
    return C * 128 + 1024


def pack_mlp(fc1_w, fc1_b, fc2_w):
    """fc1_w [4C, C], fc1_b [4C] fp32, fc2_w [C, 4C] (matrices already in bf16) -> uint8 [4C/32, C*128 + 1024]."""
    Hd, C = fc1_w.shape
    if fc2_w.shape != (C, Hd) or Hd % 32 or C % 64:
        raise ValueError('pack_mlp: unsupported shapes %s / %s' % (tuple(fc1_w.shape), tuple(fc2_w.shape)))
    if fc1_w.dtype != torch.bfloat16 or fc2_w.dtype != torch.bfloat16:
        raise TypeError('pack_mlp packs bf16 matrices (the fused kernel is bf16-only)')
    dev = fc1_w.device
    nsub = Hd // 32
    blk = mlp_block_bytes(C)
    out = torch.zeros(nsub, blk, dtype=torch.uint8, device=dev)
    # ---- W1 image: [hc][kt][r][slot][8 elems]  <- chunk (slot ^ (r & 7)) of K-tile kt of row 32*hc + r
    w1 = fc1_w.reshape(nsub, 32, C // 64, 8, 8)                       # [hc][r][kt][chunk][8]
    r = torch.arange(32, device=dev)
    slot = torch.arange(8, device=dev)
    chunk = slot[None, :] ^ (r[:, None] & 7)                            # [r][slot] -> source chunk
    w1 = w1.permute(0, 2, 1, 3, 4)                                      # [hc][kt][r][chunk][8]
    idx = chunk[None, None, :, :, None].expand(nsub, C // 64, 32, 8, 8)
    w1_img = torch.gather(w1, 3, idx).contiguous()                      # [hc][kt][r][slot][8]
    out[:, :C * 64] = w1_img.reshape(nsub, -1).view(torch.uint8).reshape(nsub, C * 64)
    # ---- W2 image: [hc][n][slot][8 elems]; permuted position 8g + 4h + rr <- hidden 16h + 4g + rr; slot = g ^ f(n & 15)
    w2 = fc2_w.reshape(C, nsub, 2, 4, 4)                                # [n][hc][h][g][rr]
    w2 = w2.permute(1, 0, 3, 2, 4).reshape(nsub, C, 4, 8)               # [hc][n][g][4h + rr]
    n = torch.arange(C, device=dev)
    f = (-((n & 15) >> 2)) & 3
    s4 = torch.arange(4, device=dev)
    g_of_slot = s4[None, :] ^ f[:, None]                                # [n][slot] -> source chunk g
    idx = g_of_slot[None, :, :, None].expand(nsub, C, 4, 8)
    w2_img = torch.gather(w2, 2, idx).contiguous()
    out[:, C * 64:C * 128] = w2_img.reshape(nsub, -1).view(torch.uint8).reshape(nsub, C * 64)
    # ---- b1 slice
    b1 = fc1_b.detach().float().reshape(nsub, 32).contiguous()
    out[:, C * 128:C * 128 + 128] = b1.view(torch.uint8).reshape(nsub, 128)
    return out


def pack_attn_block(qkv_w, proj_w, n_heads):
    """Fragment-major weight image for the fused attention half of a Swin block whose weights do not fit in LDS
    (csrc/swin_block.hip::swin_block256_kernel, C = 256 with 8 heads).  qkv_w [3C, C], proj_w [C, C] in bf16 ->
    bf16 [3C*C + C*C], two regions:

      qkv : fragment ((h * KS + ks) * 6 + sel * 2 + dt) of 64 lanes x 8 elements; lane = 16 g + li holds
            qkv_w[sel*C + 32 h + 16 dt + li][32 ks + 8 g .. + 8]          (h = head, sel = q / k / v, dt = 16-row tile of the head)
      proj: fragment ((w * 2 + nt) * KS + ks); lane 16 g + li holds proj_w[32 w + 16 nt + li][32 ks + 8 g .. + 8]

    i.e. every matrix-core operand fragment is ONE contiguous 1 KB read (a wave reads its head's fragments straight into registers),
    and the six fragments of a k-step sit next to each other.  Pure index permutation."""
    C3, C = qkv_w.shape
    if C3 != 3 * C or proj_w.shape != (C, C) or C != 32 * n_heads or C % 32:
        raise ValueError('pack_attn_block: unsupported shapes %s / %s for %d heads' % (tuple(qkv_w.shape), tuple(proj_w.shape), n_heads))
    if qkv_w.dtype != torch.bfloat16 or proj_w.dtype != torch.bfloat16:
        raise TypeError('pack_attn_block packs bf16 matrices')
    KS = C // 32
    # qkv_w[sel][h][dt][li][ks][g][e] -> [h][ks][sel][dt][g][li][e]
    q = qkv_w.reshape(3, n_heads, 2, 16, KS, 4, 8).permute(1, 4, 0, 2, 5, 3, 6).contiguous()
    # proj_w[w][nt][li][ks][g][e] -> [w][nt][ks][g][li][e]
    p = proj_w.reshape(n_heads, 2, 16, KS, 4, 8).permute(0, 1, 3, 4, 2, 5).contiguous()
    return torch.cat([q.reshape(-1), p.reshape(-1)])


# ---------------------------------------------------------------------------------------------------------------------
# Row-owner chains of the decoders' many-row phases (csrc/dec_rows.hip, include/omp355.h: omp_dec_rows_mid / omp_dec_rows_ffn)
#
# A workgroup's 8 waves each walk ONE linear stream of 1 KB matrix-core fragments: for a product with N output features over K,
# wave w owns features 64 w + 16 t (t = 0..3) of every 512-feature pass (16 w of a 128-feature pass) and consumes, k-step by
# k-step, the fragment of each of its feature tiles:  fragment[lane = 16 g + li][8] = W[f0 + li][32 ks + 8 g .. + 8].
# The packers below write those fragments in consumption order, wave-major: buffer = [8 waves][fragments of the chain][64 lanes][8]
# bf16 + 16 KB of slack (the kernels' register ring of 8 fragments in flight runs ahead of the stream's end).  Pure index permutation.
# ---------------------------------------------------------------------------------------------------------------------
ROWS_WAVES = 8
ROWS_SLACK = 8 * 1024


def _rows_pass(w, per_wave_tiles):
    """w [16 * 8 * per_wave_tiles, K] (one pass: the features of wave 0's tiles first, then wave 1's ...) -> [8 waves][K / 32 k-steps]
    [tiles][64 lanes][8]: the pass's fragments in the order every wave consumes its own.  fp32 w (the parity engine, x3): every fragment
    becomes a PAIR, the fragment of w_hi = bf16(w) followed by the fragment of w_lo = bf16(w - w_hi)."""
    n, K = w.shape
    t = per_wave_tiles
    if n != 16 * ROWS_WAVES * t or K % 32:
        raise ValueError('_rows_pass: %d features x %d are not %d waves x %d tiles of 16 x k-steps of 32' % (n, K, ROWS_WAVES, t))
    if w.dtype == torch.float32:
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        return torch.stack([_rows_pass(hi, t), _rows_pass(lo, t)], dim=2).reshape(ROWS_WAVES, 2 * (K // 32) * t, 512)
    # w[wave][tile][li][ks][g][e] -> [wave][ks][tile][g][li][e]
    return w.reshape(ROWS_WAVES, t, 16, K // 32, 4, 8).permute(0, 3, 1, 4, 2, 5).reshape(ROWS_WAVES, (K // 32) * t, 512)


def _rows_product(w):
    """Fragments of y = a W^T for W [N, K], N a multiple of 128: 512-feature passes (4 tiles per wave), then 128-feature passes (1 tile
    per wave), each over the whole K.  -> list of [8][fragments][512] pieces."""
    N = w.shape[0]
    if N % 128:
        raise ValueError('_rows_product: %d output features are not a multiple of 128' % N)
    out, f = [], 0
    while N - f >= 512:
        out.append(_rows_pass(w[f:f + 512], 4))
        f += 512
    while f < N:
        out.append(_rows_pass(w[f:f + 128], 1))
        f += 128
    return out


def _rows_finish(pieces):
    """[8][n_i][512] pieces -> (uint8 buffer, bytes per wave)."""
    s = torch.cat(pieces, dim=1).contiguous()            # [8 waves][fragments][512 bf16]
    per_wave = s.shape[1] * 1024
    buf = torch.zeros(ROWS_WAVES * per_wave + 2 * ROWS_SLACK, dtype=torch.uint8, device=s.device)   # slack: 16 fragments in flight (x3 ring)
    buf[:ROWS_WAVES * per_wave] = s.view(torch.uint8).reshape(-1)
    return buf, per_wave


def _same_kind(*ws):
    """the bf16 engine packs bf16 matrices, the parity engine (x3) the fp32 masters: no mixing"""
    kinds = set(w.dtype for w in ws)
    if kinds not in ({torch.bfloat16}, {torch.float32}):
        raise TypeError('the row-owner chains take bf16 matrices (bf16 engine) or fp32 matrices (bf16x3 engine), not %s' % sorted(str(k) for k in kinds))
    return ws[0].dtype == torch.float32


def pack_kv_rows_k(wk_all):
    """omp_kv_project_rows(vt = 0): the key projections of all (decoder, layer) slabs, [n_slabs * 512, 512] bf16 with rows ordered (slab, head,
    dim).  Inside every head the 64 dims are permuted so that matrix-core row 4 g + r of feature tile ft is dim 32 (ft / 2) + 8 g + 4 (ft % 2) + r:
    the lane that owns rows 4 g .. 4 g + 3 of the four tiles then holds dims 8 g .. 8 g + 7 (tiles 0, 1) and 32 + 8 g .. (tiles 2, 3) of a key --
    two 16-byte stores, each completing 64 contiguous bytes of the K slab row with its three neighbour lanes."""
    if wk_all.dtype != torch.bfloat16 or wk_all.shape[0] % 512 or wk_all.shape[1] != 512:
        raise ValueError('pack_kv_rows_k: [n_slabs * 512, 512] bf16')
    j = torch.arange(64)
    ft, row = j // 16, j % 16                       # position in the wave's stream: tile ft, matrix-core row
    src = (ft // 2) * 32 + (row // 4) * 8 + (ft % 2) * 4 + (row % 4)      # the dim that position computes
    w = wk_all.reshape(-1, 64, 512)[:, src.to(wk_all.device)].reshape(-1, 512)
    return _rows_finish(_rows_product(w))


def pack_kv_rows_v(wv_all):
    """omp_kv_project_rows(vt = 1): the value projections of all slabs, natural dim order (the operands are swapped: a lane owns one dim)."""
    if wv_all.dtype != torch.bfloat16 or wv_all.shape[0] % 512 or wv_all.shape[1] != 512:
        raise ValueError('pack_kv_rows_v: [n_slabs * 512, 512] bf16')
    return _rows_finish(_rows_product(wv_all))


def pack_rows_mid(sa_out_w, ca_q_w):
    """omp_dec_rows_mid: self_attn.out_proj [512, 512], then the query rows of multihead_attn.in_proj [512, 512]."""
    _same_kind(sa_out_w, ca_q_w)
    return _rows_finish(_rows_product(sa_out_w) + _rows_product(ca_q_w))


def _rows_ffn_pieces(ca_out_w, ff1_w, ff2_w):
    x3 = _same_kind(ca_out_w, ff1_w, ff2_w)
    chunk = 128 if x3 else 256      # hidden units per chunk (csrc/dec_rows_x3.hip: HC3; csrc/dec_rows.hip: HC)
    Hd, C = ff1_w.shape
    if ff2_w.shape != (C, Hd) or Hd % chunk or C != 512:
        raise ValueError('pack_rows_ffn: unsupported shapes %s / %s' % (tuple(ff1_w.shape), tuple(ff2_w.shape)))
    pieces = _rows_product(ca_out_w)
    for c in range(Hd // chunk):      # per chunk of hidden units: linear1's pass (N = chunk: chunk / 128 tiles per wave, K = 512), then linear2's (N = 512, K = chunk)
        pieces.append(_rows_pass(ff1_w[c * chunk:(c + 1) * chunk], chunk // (16 * ROWS_WAVES)))
        pieces += _rows_product(ff2_w[:, c * chunk:(c + 1) * chunk].contiguous())
    return pieces


def _qkv_tail_rows(w):
    """The q | k | v Linear behind a chain (tail 0).  bf16 engine: inside every 64-feature group (a wave's features of a 512-feature pass) the
    rows are ordered so that matrix-core row 4 g + r of feature tile ft computes feature 32 (ft / 2) + 8 g + 4 (ft % 2) + r: the lane that owns
    rows 4 g .. 4 g + 3 of the four tiles then holds features 8 g .. 8 g + 7 and 32 + 8 g .. of a row -- two 16-byte stores instead of four
    8-byte ones (csrc/dec_rows.hip store_bias_perm; a pass's stores share vmcnt with the weight ring: fewer of them, shorter stall).  The
    parity engine's chains (fp32 masters, fp32 outputs: 16 bytes per quad already) keep the natural order."""
    if w.dtype != torch.bfloat16:
        return w
    j = torch.arange(64)
    ft, row = j // 16, j % 16
    src = (ft // 2) * 32 + (row // 4) * 8 + (ft % 2) * 4 + (row % 4)
    return w.reshape(-1, 64, w.shape[1])[:, src.to(w.device)].reshape(w.shape)


def pack_rows_ffn_qkv(ca_out_w, ff1_w, ff2_w, next_sa_in_w):
    """omp_dec_rows_ffn(prologue 0, tail 0): multihead_attn.out_proj, linear1 / linear2 in chunks, then the NEXT layer's self_attn.in_proj [1536, 512]."""
    _same_kind(ca_out_w, next_sa_in_w)
    return _rows_finish(_rows_ffn_pieces(ca_out_w, ff1_w, ff2_w) + _rows_product(_qkv_tail_rows(next_sa_in_w)))


def _rows_head_pieces(h0_w, h1_w, h2_w):
    _same_kind(h0_w, h1_w, h2_w)
    V = h2_w.shape[0]
    vpad = (V + 127) // 128 * 128
    if vpad != V:
        h2_w = torch.cat([h2_w, torch.zeros(vpad - V, h2_w.shape[1], dtype=h2_w.dtype, device=h2_w.device)], 0)
    return _rows_product(h0_w) + _rows_product(h1_w) + _rows_product(h2_w)


def pack_rows_ffn_head(ca_out_w, ff1_w, ff2_w, h0_w, h1_w, h2_w):
    """omp_dec_rows_ffn(prologue 0, tail 1): the last layer's chain, then the 3-layer prediction head (vocabulary rows zero-padded to x128)."""
    _same_kind(ca_out_w, h0_w)
    return _rows_finish(_rows_ffn_pieces(ca_out_w, ff1_w, ff2_w) + _rows_head_pieces(h0_w, h1_w, h2_w))


def pack_rows_embed_qkv(sa_in_w):
    """omp_dec_rows_ffn(prologue 1, tail 0): layer 0's self_attn.in_proj behind the embedding."""
    _same_kind(sa_in_w)
    return _rows_finish(_rows_product(_qkv_tail_rows(sa_in_w)))


def pack_rows_ffn(out_w, ff1_w, ff2_w):
    """an attention out-projection and the FFN behind it with nothing after it (omp_swin_rows_block mode 1 for the last block of a stage)."""
    return _rows_finish(_rows_ffn_pieces(out_w, ff1_w, ff2_w))
