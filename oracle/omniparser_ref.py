"""TEST INFRASTRUCTURE -- CPU fp32 restatement of the reference OmniParser inference path.

This is the ORACLE: a plain-PyTorch, functional (state-dict in, tensors out) restatement of
what `OmniParser.forward(samples, seqs)` computes in eval mode in the reference
(OCR/OmniParser/model/*).  It is NOT part of the product: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it, and only as the checker.  The product path
(advancedliteratemachinery_amd/) never imports anything from oracle/.

Pinning: tests/test_oracle_vs_reference.py runs this file against the real reference classes
(imported through oracle/ref_import.py where /root/reference exists) and
tests/test_oracle_golden.py checks it against tests/golden/*.pt, which were produced from the
real reference by oracle/gen_golden.py.

Every function cites the reference lines it follows.  The algorithm is restated as the
reference runs it -- no KV cache, the full prefix re-decoded at every greedy step, the memory
broadcast to every text instance -- so it doubles as the "reference CPU path" timing baseline.
All LayerNorms use eps=1e-5 (nn.LayerNorm default), GELU is the erf form, decoder FFN is ReLU.
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-5

CORD_CLASSES = ['menu.cnt', 'menu.discountprice', 'menu.etc', 'menu.itemsubtotal', 'menu.nm',
                'menu.num', 'menu.price', 'menu.sub.cnt', 'menu.sub.nm', 'menu.sub.price',
                'menu.sub.unitprice', 'menu.unitprice', 'menu.vatyn', 'sub_total.discount_price',
                'sub_total.etc', 'sub_total.othersvc_price', 'sub_total.service_price',
                'sub_total.subtotal_price', 'sub_total.tax_price', 'total.cashprice',
                'total.changeprice', 'total.creditcardprice', 'total.emoneyprice',
                'total.menuqty_cnt', 'total.menutype_cnt', 'total.total_etc', 'total.total_price',
                'void_menu.nm', 'void_menu.price']
SROIE_CLASSES = ['company', 'address', 'date', 'total']


def _ln(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + '.weight'], sd[prefix + '.bias'], LN_EPS)


# --------------------------------------------------------------------------------------------
# Swin backbone
# --------------------------------------------------------------------------------------------
def patch_embed(sd, img, pfx='backbone.0.patch_embed.'):
    """swin_transformer.py:427-443: right/bottom zero-pad to x4, conv4x4 s4, LN over channels.
    Returns tokens (B, Wh*Ww, C) and the grid."""
    _, _, H, W = img.shape
    if W % 4:
        img = F.pad(img, (0, 4 - W % 4))
    if H % 4:
        img = F.pad(img, (0, 0, 0, 4 - H % 4))
    x = F.conv2d(img, sd[pfx + 'proj.weight'], sd[pfx + 'proj.bias'], stride=4)
    Wh, Ww = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    x = _ln(x, sd, pfx + 'norm')
    return x, Wh, Ww


def partition(x, ws):
    """swin_transformer.py:39-51."""
    B, H, W, C = x.shape
    x = x.reshape(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)


def unpartition(win, ws, H, W):
    """swin_transformer.py:54-68."""
    B = win.shape[0] // ((H // ws) * (W // ws))
    x = win.reshape(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def shift_mask(H, W, ws, shift):
    """swin_transformer.py:369-387: region ids on the PADDED grid, additive mask {0,-100}."""
    Hp = int(math.ceil(H / ws)) * ws
    Wp = int(math.ceil(W / ws)) * ws
    ids = torch.zeros((1, Hp, Wp, 1))
    cuts = ((0, -ws), (-ws, -shift), (-shift, None))
    n = 0
    for h0, h1 in cuts:
        for w0, w1 in cuts:
            ids[:, h0:h1, w0:w1, :] = n
            n += 1
    mw = partition(ids, ws).reshape(-1, ws * ws)
    diff = mw[:, None, :] - mw[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def window_attention(sd, p, xw, nH, mask):
    """swin_transformer.py:119-151. xw: (nW*B, N, C)."""
    Bw, N, C = xw.shape
    hd = C // nH
    qkv = F.linear(xw, sd[p + 'qkv.weight'], sd[p + 'qkv.bias'])
    qkv = qkv.reshape(Bw, N, 3, nH, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    att = q @ k.transpose(-2, -1)
    table = sd[p + 'relative_position_bias_table']
    idx = sd[p + 'relative_position_index'].reshape(-1)
    bias = table[idx].reshape(N, N, nH).permute(2, 0, 1)
    att = att + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        att = att.reshape(Bw // nW, nW, nH, N, N) + mask[None, :, None]
        att = att.reshape(-1, nH, N, N)
    att = att.softmax(-1)
    out = (att @ v).transpose(1, 2).reshape(Bw, N, C)
    return F.linear(out, sd[p + 'proj.weight'], sd[p + 'proj.bias'])


def swin_block(sd, p, x, H, W, nH, ws, shift, mask):
    """swin_transformer.py:196-253.  NB: zero padding happens AFTER norm1, padded tokens join
    attention unmasked (:209-217); crop after the reverse roll (:243-244)."""
    B, L, C = x.shape
    short = x
    y = _ln(x, sd, p + 'norm1').reshape(B, H, W, C)
    pr = (ws - W % ws) % ws
    pb = (ws - H % ws) % ws
    y = F.pad(y, (0, 0, 0, pr, 0, pb))
    Hp, Wp = H + pb, W + pr
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    yw = partition(y, ws).reshape(-1, ws * ws, C)
    aw = window_attention(sd, p + 'attn.', yw, nH, mask if shift > 0 else None)
    y = unpartition(aw.reshape(-1, ws, ws, C), ws, Hp, Wp)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    y = y[:, :H, :W, :].reshape(B, H * W, C)
    x = short + y
    h = _ln(x, sd, p + 'norm2')
    h = F.linear(h, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'])
    h = F.gelu(h)
    h = F.linear(h, sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'])
    return x + h


def patch_merging(sd, p, x, H, W):
    """swin_transformer.py:269-296: pad to even, gather (0,0),(1,0),(0,1),(1,1), LN(4C), Linear."""
    B, L, C = x.shape
    x = x.reshape(B, H, W, C)
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    parts = [x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]]
    x = torch.cat(parts, -1).reshape(B, -1, 4 * C)
    x = _ln(x, sd, p + 'norm')
    return F.linear(x, sd[p + 'reduction.weight'])


def swin_forward(sd, img, mask, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), ws=7):
    """swin_transformer.py:597-625 (+ BasicLayer :360-400).  Returns per-stage NCHW maps after
    norm{i} and the nearest-resized padding masks."""
    pfx = 'backbone.0.'
    x, H, W = patch_embed(sd, img)
    feats, masks = [], []
    for s, (dep, nH) in enumerate(zip(depths, num_heads)):
        m = shift_mask(H, W, ws, ws // 2)
        for b in range(dep):
            x = swin_block(sd, f'{pfx}layers.{s}.blocks.{b}.', x, H, W, nH, ws,
                           0 if b % 2 == 0 else ws // 2, m)
        out = _ln(x, sd, f'{pfx}norm{s}')
        C = out.shape[-1]
        feats.append(out.reshape(-1, H, W, C).permute(0, 3, 1, 2).contiguous())
        masks.append(F.interpolate(mask[None].float(), size=(H, W)).to(torch.bool)[0])
        if s < len(depths) - 1:
            x = patch_merging(sd, f'{pfx}layers.{s}.downsample.', x, H, W)
            H, W = (H + 1) // 2, (W + 1) // 2
    return feats, masks


def sine_position(mask, num_pos_feats=256, temperature=10000.0):
    """position_embedding.py:24-44 with normalize=True, scale=2*pi.  mask (B,h,w) bool."""
    nm = ~mask
    y = nm.cumsum(1, dtype=torch.float32)
    x = nm.cumsum(2, dtype=torch.float32)
    y = y / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
    i = torch.arange(num_pos_feats, dtype=torch.float32)
    div = temperature ** (2 * torch.div(i, 2, rounding_mode='floor') / num_pos_feats)
    px = x[..., None] / div
    py = y[..., None] / div
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def fpn(sd, feats):
    """fpn.py:21-45: 1x1 convs (no bias), nearest top-down adds, bilinear resample of p2,p4,p5
    to c3's size (align_corners=False), concat (p2,p3,p4,p5)."""
    c2, c3, c4, c5 = feats
    w = [sd[f'fpn.fpn_in.{i}.weight'] for i in range(4)]
    p5 = F.conv2d(c5, w[0])
    p4 = F.conv2d(c4, w[1]) + F.interpolate(p5, size=c4.shape[2:], mode='nearest')
    p3 = F.conv2d(c3, w[2]) + F.interpolate(p4, size=c3.shape[2:], mode='nearest')
    p2 = F.conv2d(c2, w[3]) + F.interpolate(p3, size=c2.shape[2:], mode='nearest')
    sz = c3.shape[2:]
    p2 = F.interpolate(p2, size=sz, mode='bilinear')
    p4 = F.interpolate(p4, size=sz, mode='bilinear')
    p5 = F.interpolate(p5, size=sz, mode='bilinear')
    return torch.cat((p2, p3, p4, p5), dim=1)


def encode(sd, args, img, mask, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), ws=7):
    """omniparser.py:19-31 + transformer.py:219-226: backbone -> (FPN) -> input_proj -> flatten.
    Returns dict with memory (HW,B,d), mask (B,HW), pos (HW,B,d) and the intermediates."""
    feats, masks = swin_forward(sd, img, mask, depths, num_heads, ws)
    npf = args.tfm_hidden_dim // 2
    if args.use_fpn:
        src = fpn(sd, feats)
        m = masks[-2]
        pos = sine_position(masks[-2], npf)
        proj = F.conv2d(src, sd['input_proj.weight'], sd['input_proj.bias'], stride=2)
    else:
        src = feats[-1]
        m = masks[-1]
        pos = sine_position(masks[-1], npf)
        proj = F.conv2d(src, sd['input_proj.weight'], sd['input_proj.bias'])
    return dict(feats=feats, masks=masks, src=src, proj=proj,
                memory=proj.flatten(2).permute(2, 0, 1), mask=m.flatten(1),
                pos=pos.flatten(2).permute(2, 0, 1), hw=tuple(proj.shape[2:]))


# --------------------------------------------------------------------------------------------
# Decoder
# --------------------------------------------------------------------------------------------
def embed(sd, seq, kind):
    """transformer.py:302-328: LN(word[x] + pos_kind[arange(L)]); also returns the RAW position
    embeddings, which the layers re-add to q/k."""
    L = seq.shape[1]
    pe = sd[f'transformer.embedding.{kind}_position_embeddings.weight'][:L]
    pe = pe[None].expand(seq.shape[0], L, -1)
    e = sd['transformer.embedding.word_embeddings.weight'][seq] + pe
    return _ln(e, sd, 'transformer.embedding.LayerNorm'), pe


def causal_mask(L):
    """transformer.py:331-337."""
    return torch.triu(torch.full((L, L), float('-inf')), diagonal=1)


def mha(sd, p, q_in, k_in, v_in, nH, attn_mask=None, key_padding_mask=None):
    """nn.MultiheadAttention forward (packed in_proj rows [0:E]=q,[E:2E]=k,[2E:3E]=v; q scaled
    by 1/sqrt(head_dim) before QK^T; masks additive; softmax; out_proj), as called from
    transformer.py:412-420,437-447.  Shapes (L,N,E)/(S,N,E)."""
    E = q_in.shape[-1]
    hd = E // nH
    W, bvec = sd[p + 'in_proj_weight'], sd[p + 'in_proj_bias']
    q = F.linear(q_in, W[:E], bvec[:E])
    k = F.linear(k_in, W[E:2 * E], bvec[E:2 * E])
    v = F.linear(v_in, W[2 * E:], bvec[2 * E:])
    L, N, _ = q.shape
    S = k.shape[0]
    q = q.reshape(L, N, nH, hd).permute(1, 2, 0, 3) * (1.0 / math.sqrt(hd))
    k = k.reshape(S, N, nH, hd).permute(1, 2, 0, 3)
    v = v.reshape(S, N, nH, hd).permute(1, 2, 0, 3)
    att = q @ k.transpose(-2, -1)
    if attn_mask is not None:
        att = att + attn_mask
    if key_padding_mask is not None:
        att = att.masked_fill(key_padding_mask[:, None, None, :], float('-inf'))
    att = att.softmax(-1)
    out = (att @ v).permute(2, 0, 1, 3).reshape(L, N, E)
    return F.linear(out, sd[p + 'out_proj.weight'], sd[p + 'out_proj.bias'])


def decoder_layer(sd, p, x, memory, mem_kpm, pos, qpos, tmask, nH, pre_norm):
    """transformer.py:430-454 (pre-norm) / :407-428 (post-norm)."""
    if pre_norm:
        y = _ln(x, sd, p + 'norm1')
        qk = y + qpos
        x = x + mha(sd, p + 'self_attn.', qk, qk, y, nH, attn_mask=tmask)
        y = _ln(x, sd, p + 'norm2')
        x = x + mha(sd, p + 'multihead_attn.', y + qpos, memory + pos, memory, nH,
                    key_padding_mask=mem_kpm)
        y = _ln(x, sd, p + 'norm3')
        y = F.linear(F.relu(F.linear(y, sd[p + 'linear1.weight'], sd[p + 'linear1.bias'])),
                     sd[p + 'linear2.weight'], sd[p + 'linear2.bias'])
        return x + y
    qk = x + qpos
    x = _ln(x + mha(sd, p + 'self_attn.', qk, qk, x, nH, attn_mask=tmask), sd, p + 'norm1')
    x = _ln(x + mha(sd, p + 'multihead_attn.', x + qpos, memory + pos, memory, nH,
                    key_padding_mask=mem_kpm), sd, p + 'norm2')
    y = F.linear(F.relu(F.linear(x, sd[p + 'linear1.weight'], sd[p + 'linear1.bias'])),
                 sd[p + 'linear2.weight'], sd[p + 'linear2.bias'])
    return _ln(x + y, sd, p + 'norm3')


def head(sd, kind, x):
    """block/mlp.py:4-14 via transformer.py:35-37: 512->512->512->V, ReLU."""
    p = f'transformer.{kind}_pred_layer.layers.'
    x = F.relu(F.linear(x, sd[p + '0.weight'], sd[p + '0.bias']))
    x = F.relu(F.linear(x, sd[p + '1.weight'], sd[p + '1.bias']))
    return F.linear(x, sd[p + '2.weight'], sd[p + '2.bias'])


def decode(sd, args, seq, memory, mask, pos, kind):
    """transformer.py:74-100 in eval mode: every one of the N sequences attends to the (same)
    image memory; full prefix re-decoded; head on all positions.  memory/pos (M,1,d), mask (1,M).
    Returns logits (N, L, V)."""
    x, qpos = embed(sd, seq, kind)
    x = x.permute(1, 0, 2)
    qpos = qpos.permute(1, 0, 2)
    L, N, _ = x.shape
    mem = memory.expand(-1, N, -1)
    pe = pos.expand(-1, N, -1)
    kpm = mask.expand(N, -1)
    tmask = causal_mask(L)
    for l in range(args.tfm_dec_layers):
        x = decoder_layer(sd, f'transformer.{kind}_decoder.layers.{l}.', x, mem, kpm, pe, qpos,
                          tmask, args.tfm_nheads, args.tfm_pre_norm)
    x = _ln(x, sd, f'transformer.{kind}_decoder.norm')
    return head(sd, kind, x.transpose(0, 1))


def prompt_len(args):
    return 7 if args.use_char_window_prompt else 5


def pt_step_filter(args, probs, i):
    """transformer.py:110-123: which tokens may be emitted at greedy step i (in-place style)."""
    nb, eos = args.num_bins, args.pt_eos_index
    period = 3 if args.infer_vie else 2
    r = i % period
    if r == 0:
        probs = probs.clone()
        probs[:, nb:eos] = 0
        probs[:, eos + 1:] = 0
        return probs
    if r == 1:
        return probs[:, :nb]
    probs = probs.clone()
    probs[:, :-args.vie_categories] = 0
    return probs


def decode_pt_seq(sd, args, prompt, memory, mask, pos, max_steps=None):
    """transformer.py:102-141.  Returns (ids (2N,), [probs])."""
    seq = prompt
    probs_out = []
    steps = args.pt_seq_length if max_steps is None else max_steps
    for i in range(steps):
        logits = decode(sd, args, seq, memory, mask, pos, 'pt')[:, -1, :]
        pr = pt_step_filter(args, logits.softmax(-1), i)
        p, tok = pr.topk(dim=-1, k=1)
        if tok[0] == args.pt_eos_index:
            break
        seq = torch.cat([seq, tok], dim=-1)
        probs_out.append(p)
    seq = seq[:, prompt_len(args):]
    if seq.shape[1] % 2 != 0:
        seq = seq[:, :-1]
    return seq[0], probs_out


def rec_filter(args, probs):
    """transformer.py:275-278 / :177-180."""
    probs = probs.clone()
    probs[:, :args.num_bins] = 0
    probs[:, args.pt_eos_index] = 0
    probs[:, args.poly_eos_index] = 0
    probs[:, args.rec_eos_index + 1:] = 0
    return probs


def spot(sd, args, pt_seq, poly_prompt, rec_prompt, memory, mask, pos):
    """transformer.py:247-286: 32 poly steps then rec_length rec steps for all N instances."""
    pts = pt_seq.reshape(-1, 2)
    N = pts.shape[0]
    poly = torch.cat((pts, poly_prompt.repeat(N, 1)), dim=-1)
    for _ in range(32):
        pr = decode(sd, args, poly, memory, mask, pos, 'poly')[:, -1, :].softmax(-1)
        _, tok = pr[:, :args.num_bins].topk(dim=-1, k=1)
        poly = torch.cat([poly, tok], dim=-1)
    poly = poly[:, 3:35]
    rec = torch.cat((pts, rec_prompt.repeat(N, 1)), dim=-1)
    rprobs = []
    for _ in range(args.rec_length):
        pr = rec_filter(args, decode(sd, args, rec, memory, mask, pos, 'rec')[:, -1, :].softmax(-1))
        p, tok = pr.topk(dim=-1, k=1)
        rec = torch.cat([rec, tok], dim=-1)
        rprobs.append(p)
    rec = rec[:, 3:].unsqueeze(0)
    return [pts.reshape(1, -1), poly.reshape(1, -1), rec], [torch.cat(rprobs, dim=-1)]


def index2class(args):
    """transformer.py:49-67."""
    names = None
    if args.val_dataset:
        if 'cord' in args.val_dataset[0]:
            names = CORD_CLASSES
        elif 'sroie' in args.val_dataset[0]:
            names = SROIE_CLASSES
    if names is None:
        return {}
    return {args.padding_index + 1 + i: n for i, n in enumerate(names)}


def kie(sd, args, pt_seq, pt_probs, poly_prompt, rec_prompt, image_size, memory, mask, pos):
    """transformer.py:143-217: walk (x, y, class) triplets; per word 32 poly + rec_length rec
    steps at N=1 with the class logits sliced off BEFORE softmax (:156,:176)."""
    i2c = index2class(args)
    vc = args.vie_categories
    nb = args.num_bins
    out, words, rects = [], [], []
    i = 0
    n = len(pt_seq)
    while i < n:
        if pt_seq[i].item() < nb:
            if i + 1 <= n - 1 and pt_seq[i + 1].item() < nb:
                pt = pt_seq[i:i + 2].unsqueeze(0)
                poly = torch.cat((pt, poly_prompt), dim=-1)
                for _ in range(32):
                    lg = decode(sd, args, poly, memory, mask, pos, 'poly')[:, -1, :-vc]
                    _, tok = lg.softmax(-1)[:, :nb].topk(dim=-1, k=1)
                    poly = torch.cat([poly, tok], dim=-1)
                ih, iw = image_size
                pp = poly[0, 3:35].reshape(-1, 2)
                rect = [iw.item() * pp[:, 0].min().item() / nb, ih.item() * pp[:, 1].min().item() / nb,
                        iw.item() * pp[:, 0].max().item() / nb, ih.item() * pp[:, 1].max().item() / nb]
                rec = torch.cat((pt, rec_prompt), dim=-1)
                for _ in range(args.rec_length):
                    lg = decode(sd, args, rec, memory, mask, pos, 'rec')[:, -1, :-vc]
                    _, tok = rec_filter(args, lg.softmax(-1)).topk(dim=-1, k=1)
                    rec = torch.cat([rec, tok], dim=-1)
                chars = []
                for t in rec[0, 3:]:
                    t = int(t)
                    if t == args.recog_pad_index or t == args.rec_eos_index:
                        break
                    if t == args.recog_pad_index - 1:
                        continue
                    chars.append(args.chars[t - nb])
                words.append(''.join(chars))
                rects.append(rect)
                i += 2
            else:
                i += 1
        else:
            out.append((' '.join(words), i2c[pt_seq[i].item()], pt_probs[i].item(), rects))
            i += 1
            words, rects = [], []
    return out


def forward(sd, args, img, mask, seqs, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32),
            pt_max_steps=None, return_encoded=False):
    """OmniParser.forward in eval mode for ONE image (reference asserts batch 1, val.py:22).
    img (1,3,H,W), mask (1,H,W) bool, seqs = [pt_prompt, poly_prompt, rec_prompt, (orig_size)]."""
    enc = encode(sd, args, img, mask, depths, num_heads)
    mem, m, pos = enc['memory'], enc['mask'], enc['pos']
    pt_seq, pt_probs = decode_pt_seq(sd, args, seqs[0], mem, m, pos, pt_max_steps)
    if pt_seq.numel() == 0:
        res = None
    elif args.infer_vie:
        res = kie(sd, args, pt_seq, pt_probs, seqs[1], seqs[2], seqs[3], mem, m, pos)
    else:
        res = spot(sd, args, pt_seq, seqs[1], seqs[2], mem, m, pos)
    return (res, enc) if return_encoded else res


def default_prompts(args):
    """engine/val.py:25-33."""
    nb = args.num_bins
    if args.use_char_window_prompt:
        pt = torch.tensor([[0, 0, nb - 1, nb - 1, nb, nb + len(args.chars), args.pt_sos_index]])
    else:
        pt = torch.tensor([[0, 0, nb - 1, nb - 1, args.pt_sos_index]])
    return [pt.long(), torch.full((1, 1), args.poly_sos_index, dtype=torch.long),
            torch.full((1, 1), args.rec_sos_index, dtype=torch.long)]
