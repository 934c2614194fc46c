"""MGP-STR scene-text recogniser on libomp355 (BASELINE config 5): drop-in for the reference's
`Model(opt).mgp_str` on the inference path (OCR/MGP-STR/models.py:24-40, modules/mgp_str.py:46-101).

Same state-dict layout as the reference checkpoint (`mgp_str.*` keys of timm's VisionTransformer + the three
TokenLearners and heads; a DataParallel `module.` prefix is stripped by `load_reference_state_dict`), same
call contract: `forward(images, is_eval=False)` -> [char_out (B,27,38), bpe_out (B,27,50257), wp_out (B,27,30522)]
and with `is_eval=True` -> [[char_attn, bpe_attn, wp_attn], char_out, bpe_out, wp_out] (mgp_str.py:96-101).
`recognize()` adds the result decoding / fusion of test_final.py:145-240 on the device (ids + confidences; the
BPE / WordPiece strings need the GPT-2 / BERT vocabulary files and are left to the caller's tokenizers).

Execution: activations token-major [B*257, 768] in the engine dtype; the ViT-B encoder runs on the shared
kernels (omp_layernorm, omp_gemm_bias_act) with the k / v projections writing blocked K / V^T slabs; the 257-token
self-attention is csrc/vit.hip::vit_attn_kernel in bf16 (one workgroup per (image, head), keys and values in LDS)
and the blocked cross-attention kernels (omp_dec_cross_attn_step) in fp32.  There is no CPU fallback.
"""
import torch
import torch.nn as nn

from .. import _lib, ops

# 'bf16x3' = the PARITY engine (round 5, as OmniParser's): fp32 storage and fp32 non-GEMM kernels, every large product as three bf16
# matrix-core products of split operands (include/omp355.h, omp_gemm_args.a_wrap), the attention over split-plane K / V^T slabs
_DTYPES = {'bf16': torch.bfloat16, 'fp32': torch.float32, 'bf16x3': 'bf16x3', torch.bfloat16: torch.bfloat16, torch.float32: torch.float32}
BASE_CFG = dict(embed=768, depth=12, heads=12, mlp_ratio=4, img=(32, 128), patch=4, max_len=27, num_class=38,
                bpe_vocab=50257, wp_vocab=30522)
CHARACTER = '0123456789abcdefghijklmnopqrstuvwxyz'
CHAR_TABLE = ['[GO]', '[s]'] + list(CHARACTER)   # utils.py:15-21
BPE_EOS, WP_EOS = 2, 102                          # test_final.py:206,227
LN_EPS_BLOCK, LN_EPS_A3 = 1e-6, 1e-5              # timm 0.4.12 blocks / nn.LayerNorm default in TokenLearner
GRANULARITIES = ('char', 'bpe', 'wp')


def expected_state_dict(c, prefix='mgp_str.'):
    """name -> shape of every tensor of the reference checkpoint (incl. timm's unused final norm and head)."""
    E, L = c['embed'], c['max_len']
    Hd = int(E * c['mlp_ratio'])
    T = (c['img'][0] // c['patch']) * (c['img'][1] // c['patch']) + 1
    sp = {}
    p = prefix
    sp[p + 'cls_token'] = (1, 1, E)
    sp[p + 'pos_embed'] = (1, T, E)
    sp[p + 'patch_embed.proj.weight'] = (E, 3, c['patch'], c['patch'])
    sp[p + 'patch_embed.proj.bias'] = (E,)
    for i in range(c['depth']):
        b = '%sblocks.%d.' % (p, i)
        for n, shape in (('norm1.weight', (E,)), ('norm1.bias', (E,)), ('attn.qkv.weight', (3 * E, E)), ('attn.qkv.bias', (3 * E,)),
                         ('attn.proj.weight', (E, E)), ('attn.proj.bias', (E,)), ('norm2.weight', (E,)), ('norm2.bias', (E,)),
                         ('mlp.fc1.weight', (Hd, E)), ('mlp.fc1.bias', (Hd,)), ('mlp.fc2.weight', (E, Hd)), ('mlp.fc2.bias', (E,))):
            sp[b + n] = shape
    sp[p + 'norm.weight'] = (E,)
    sp[p + 'norm.bias'] = (E,)
    sp[p + 'head.weight'] = (c['num_class'], E)
    sp[p + 'head.bias'] = (c['num_class'],)
    for name, vocab in (('char', c['num_class']), ('bpe', c['bpe_vocab']), ('wp', c['wp_vocab'])):
        t = '%s%s_tokenLearner.' % (p, name)
        sp[t + 'token_norm.weight'] = (E,)
        sp[t + 'token_norm.bias'] = (E,)
        sp[t + 'tokenLearner.0.weight'] = (E, E // 8, 1, 1)
        sp[t + 'tokenLearner.1.weight'] = (L, E, 1, 1)
        sp[t + 'feat.weight'] = (E, E // 8, 1, 1)
        sp[t + 'norm.weight'] = (E,)
        sp[t + 'norm.bias'] = (E,)
        sp['%s%s_head.weight' % (p, name)] = (vocab, E)
        sp['%s%s_head.bias' % (p, name)] = (vocab,)
    return sp


def _grouped_to_dense(w, groups=8):
    """weight [E, E/groups, 1, 1] of a grouped 1x1 conv -> the equivalent dense [E, E] matrix (zeros off the
    diagonal blocks; adding exact zeros leaves every fp32 sum unchanged)."""
    E, per = w.shape[0], w.shape[1]
    dense = torch.zeros(E, per * groups, dtype=w.dtype, device=w.device)
    og = E // groups
    for g in range(groups):
        dense[g * og:(g + 1) * og, g * per:(g + 1) * per] = w[g * og:(g + 1) * og, :, 0, 0]
    return dense


class _Engine(object):
    """weights packed once: matrices in the engine dtype, vectors fp32"""

    def __init__(self, sd, c, dtype, prefix):
        self.x3 = dtype == 'bf16x3'
        if self.x3:
            dtype = torch.float32
        self.c, self.dtype = c, dtype
        E = c['embed']
        f32 = lambda k: sd[prefix + k].detach().float().contiguous()        # noqa: E731
        if self.x3:   # [w_hi | w_hi | w_lo] images (ops.split_weight3): the W side of a bf16x3 product
            mat = lambda t: ops.split_weight3(t)                            # noqa: E731
        else:
            mat = lambda t: t.detach().to(dtype).contiguous()               # noqa: E731
        self.pe_w = f32('patch_embed.proj.weight').reshape(E, -1).contiguous()
        self.pe_b = f32('patch_embed.proj.bias')
        self.cls = f32('cls_token').reshape(E).contiguous()
        self.pos = f32('pos_embed').reshape(-1, E).contiguous()
        self.blocks = []
        for i in range(c['depth']):
            b = 'blocks.%d.' % i
            qkv_w, qkv_b = f32(b + 'attn.qkv.weight'), f32(b + 'attn.qkv.bias')
            self.blocks.append(dict(
                n1=(f32(b + 'norm1.weight'), f32(b + 'norm1.bias')),
                wq=mat(qkv_w[:E]), bq=qkv_b[:E].contiguous(),
                wk=mat(qkv_w[E:2 * E]), bk=qkv_b[E:2 * E].contiguous(),
                wv=(ops.split_weight2(qkv_w[2 * E:]) if self.x3 else mat(qkv_w[2 * E:])), bv=qkv_b[2 * E:].contiguous(),   # V^T: swapped operands, the weight is the A side
                wo=mat(f32(b + 'attn.proj.weight')), bo=f32(b + 'attn.proj.bias'),
                n2=(f32(b + 'norm2.weight'), f32(b + 'norm2.bias')),
                w1=mat(f32(b + 'mlp.fc1.weight')), b1=f32(b + 'mlp.fc1.bias'),
                w2=mat(f32(b + 'mlp.fc2.weight')), b2=f32(b + 'mlp.fc2.bias')))
            if not self.x3 and dtype == torch.bfloat16:   # the fused projection of the token-major attention path (ops.vit_attn_qkv)
                self.blocks[-1].update(wqkv=mat(qkv_w), bqkv=qkv_b.contiguous())
        self.a3 = {}
        for name in GRANULARITIES:
            t = name + '_tokenLearner.'
            self.a3[name] = dict(
                tn=(f32(t + 'token_norm.weight'), f32(t + 'token_norm.bias')),
                wg=mat(_grouped_to_dense(f32(t + 'tokenLearner.0.weight'))),
                wsel=(f32(t + 'tokenLearner.1.weight').reshape(c['max_len'], E).contiguous() if self.x3   # 27 rows: stays an fp32 product
                      else mat(f32(t + 'tokenLearner.1.weight').reshape(c['max_len'], E))),
                wfeat=mat(_grouped_to_dense(f32(t + 'feat.weight'))),
                n=(f32(t + 'norm.weight'), f32(t + 'norm.bias')),
                hw=mat(f32(name + '_head.weight')), hb=f32(name + '_head.bias'))
        self._slabs = {}

    def slabs(self, B, T, dev):
        """K / V^T slabs of ONE layer (reused by every layer: a layer's attention is finished before the next
        layer's projections overwrite them, all on one stream).  Zero-initialised: the padded key tail stays 0."""
        KB = 16 if (self.dtype == torch.float32 and not self.x3) else 32
        Mpad = (T + KB - 1) // KB * KB
        key = (B, T)
        if key not in self._slabs:
            nH = self.c['heads']
            groups = []
            for b in range(B):
                for o in range(0, T, 64):
                    groups.append((b * T + o, min(64, T - o), b))
            if self.x3:   # split-plane slabs: every 32-key block = [hi plane | lo plane] of bf16 (the bytes of the fp32 slabs)
                slabs = (torch.zeros(1, B, nH, Mpad // 32, 2, 32, 64, dtype=torch.bfloat16, device=dev),
                         torch.zeros(1, B, nH, Mpad // 32, 2, 64, 32, dtype=torch.bfloat16, device=dev))
            else:
                slabs = (torch.zeros(1, B, nH, Mpad, 64, dtype=self.dtype, device=dev),
                         torch.zeros(1, B, nH, Mpad // KB, 64, KB, dtype=self.dtype, device=dev))
            self._slabs[key] = slabs + (torch.tensor(groups, dtype=torch.int32, device=dev), len(groups), Mpad, KB)
        return self._slabs[key]


class MGPSTR(nn.Module):
    def __init__(self, cfg=None, engine_dtype='bf16', prefix='mgp_str.'):
        super().__init__()
        self.cfg = dict(BASE_CFG)
        self.cfg.update(cfg or {})
        if self.cfg['embed'] != self.cfg['heads'] * 64:
            raise ValueError('libomp355 attention kernels are built for head_dim 64 (embed %d, heads %d)'
                             % (self.cfg['embed'], self.cfg['heads']))
        if self.cfg['patch'] != 4:
            raise ValueError('omp_vit_patch_embed is built for 4x4 patches')
        self.prefix = prefix
        self.engine_dtype = _DTYPES[engine_dtype]
        self._names = {}
        for k, shape in expected_state_dict(self.cfg, prefix).items():
            pname = k.replace('.', '__')
            self._names[k] = pname
            self.register_parameter(pname, nn.Parameter(torch.zeros(*shape), requires_grad=False))
        self._engine, self._engine_key = None, None
        self.vit_attn_kernel = True    # bf16, 257 tokens: csrc/vit.hip::vit_attn_kernel; False = blocked cross-attention kernels
        self.vit_qkv_fused = True      # bf16: q | k | v as ONE token-major product, V^T built in LDS by the attention kernel (False: blocked K / V^T slabs; A/B, tests)
        self.greedy_fused = True       # recognize(): wide heads decode from the head product's row statistics, no logits tensor (False: logits + arg-max pass; A/B, tests)
        self.eval()

    # reference key names in and out ----------------------------------------------------------------
    def state_dict(self, *a, **k):
        inner = super().state_dict(*a, **k)
        back = {v: key for key, v in self._names.items()}
        return {back.get(n, n): t for n, t in inner.items()}

    def load_state_dict(self, sd, strict=True):
        sd = {self._names.get(k, k): v for k, v in sd.items()}
        return super().load_state_dict(sd, strict=strict)

    def load_reference_state_dict(self, sd):
        """checkpoint as test_final.py:348-356 loads it: keys of DataParallel(Model) = `module.mgp_str.*`."""
        sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
        return self.load_state_dict(sd, strict=True)

    def set_engine_dtype(self, dtype):
        self.engine_dtype = _DTYPES[dtype]
        self._engine = None

    def engine(self):
        ps = list(self.parameters())
        key = (ps[0].device, self.engine_dtype, sum(p._version for p in ps), id(ps[0]))
        if self._engine is None or key != self._engine_key:
            if key[0].type != 'cuda':
                raise RuntimeError('MGPSTR runs on MI355X only: move the model to a cuda (HIP) device; there is no CPU fallback')
            with torch.cuda.device(key[0]):
                self._engine = _Engine(self.state_dict(), self.cfg, self.engine_dtype, self.prefix)
            self._engine_key = key
        return self._engine

    # encoder ---------------------------------------------------------------------------------------
    def encode(self, img):
        """img [B,3,32,128] fp32 (device) -> tokens [B*257, 768] after the 12 blocks (no final norm, mgp_str.py:73-74)."""
        e, c = self.engine(), self.cfg
        B, T = img.shape[0], (img.shape[2] // 4) * (img.shape[3] // 4) + 1
        if (img.shape[2], img.shape[3]) != tuple(c['img']):
            raise ValueError("Input image size (%d*%d) doesn't match model (%d*%d)." % (img.shape[2], img.shape[3], c['img'][0], c['img'][1]))
        E, nH, dt = c['embed'], c['heads'], self.engine_dtype
        if e.x3:
            return self._encode_x3(img, e, B, T)
        x = ops.vit_patch_embed(img.float().contiguous(), e.pe_w, e.pe_b, e.cls, e.pos, dt).view(B * T, E)
        K, Vt, groups, n_groups, Mpad, KB = e.slabs(B, T, img.device)
        geom = (B, T, Mpad, nH, KB)
        y = torch.empty_like(x)
        att = torch.empty_like(x)
        fused_qkv = self.vit_attn_kernel and self.vit_qkv_fused and dt == torch.bfloat16 and T <= 288 and 'wqkv' in e.blocks[0]
        qkv = torch.empty((B * T, 3 * E), dtype=dt, device=x.device) if fused_qkv else None
        for blk in e.blocks:
            ops.layernorm(x, blk['n1'][0], blk['n1'][1], out=y, eps=LN_EPS_BLOCK)
            if fused_qkv:
                # ONE plain product [B T, 3 E] (timm's fused qkv Linear); the attention kernel reads its three column groups and builds V^T in LDS
                ops.gemm(y, blk['wqkv'], blk['bqkv'], out=qkv)
                ops.vit_attn_qkv(qkv, att, B, T, nH)
            else:
                q = ops.gemm(y, blk['wq'], blk['bq'])
                ops.gemm(y, blk['wk'], blk['bk'], out=K, store_mode=_lib.STORE_KBLK, kv=geom)
                ops.gemm(blk['wv'], y, blk['bv'], out=Vt, store_mode=_lib.STORE_VBLK, kv=geom, bias_along_m=True, M=E, N=B * T, K=E)
                if self.vit_attn_kernel and dt == torch.bfloat16 and Mpad == 288:
                    ops.vit_attn(q, K[0], Vt[0], att, B, T, nH, Mpad)     # one workgroup per (image, head), K / V^T in LDS
                else:
                    ops.dec_cross_attn_step(q, K[0], Vt[0], nH * Mpad * 64, Mpad, None, groups, n_groups, 4, None, att, T, nH, 1)
            ops.gemm(att, blk['wo'], blk['bo'], residual=x, out=x)
            ops.layernorm(x, blk['n2'][0], blk['n2'][1], out=y, eps=LN_EPS_BLOCK)
            h = ops.gemm(y, blk['w1'], blk['b1'], act=ops.ACT_GELU)
            ops.gemm(h, blk['w2'], blk['b2'], residual=x, out=x)
        return x, B, T

    def _encode_x3(self, img, e, B, T):
        """The parity engine's encoder: fp32 residual stream, split-pair GEMM operands, split-plane K / V^T slabs, the 257-token
        attention on the blocked cross-attention kernels (three bf16 products per score / value block)."""
        c = self.cfg
        E, nH, S = c['embed'], c['heads'], ops.SPLIT
        f32 = torch.float32
        x = ops.vit_patch_embed(img.float().contiguous(), e.pe_w, e.pe_b, e.cls, e.pos, f32).view(B * T, E)
        K, Vt, groups, n_groups, Mpad, KB = e.slabs(B, T, img.device)
        geom = (B, T, Mpad, nH, KB)
        yf = torch.empty_like(x)
        att = torch.empty_like(x)
        ys = torch.empty((B * T, 2 * E), dtype=torch.bfloat16, device=x.device)
        for blk in e.blocks:
            ops.layernorm(x, blk['n1'][0], blk['n1'][1], out=ys, out_dtype=S, out_f32=yf, eps=LN_EPS_BLOCK)
            q = ops.gemm(ys, blk['wq'], blk['bq'], out_dtype=f32, a_wrap=2 * E)
            ops.gemm(ys, blk['wk'], blk['bk'], out=K, out_dtype=S, store_mode=_lib.STORE_KBLK, kv=geom, a_wrap=2 * E, M=B * T, N=E, K=3 * E)
            ops.gemm(blk['wv'], ops.split_bf16(yf, triple=True), blk['bv'], out=Vt, out_dtype=S, store_mode=_lib.STORE_VBLK, kv=geom, bias_along_m=True,
                     a_wrap=2 * E, M=E, N=B * T, K=3 * E)
            ops.dec_cross_attn_step(q, K[0], Vt[0], nH * Mpad * 128, Mpad, None, groups, n_groups, 4, None, att, T, nH, 1)
            ops.gemm(ops.split_bf16(att), blk['wo'], blk['bo'], residual=x, out=x, a_wrap=2 * E)
            ops.layernorm(x, blk['n2'][0], blk['n2'][1], out=ys, out_dtype=S, eps=LN_EPS_BLOCK)
            h = ops.gemm(ys, blk['w1'], blk['b1'], act=ops.ACT_GELU, out_dtype=S, a_wrap=2 * E)
            ops.gemm(h, blk['w2'], blk['b2'], residual=x, out=x, a_wrap=h.shape[1])
        return x, B, T

    def _a3_head(self, x, B, T, name, want_attn, greedy=False):
        """greedy=True (recognize): -> (attn, (ids [B, S], prob [B, S])) -- the head product keeps per-tile row statistics instead of writing the
        logits (ops.gemm_row_argmax_prob; the BPE / WordPiece heads are 50 257 / 30 522 classes wide: 2.8 / 1.7 GB of fp32 per 512-word batch)."""
        e, c = self.engine(), self.cfg
        a, S = e.a3[name], c['max_len']
        if e.x3:
            E, f32 = c['embed'], torch.float32
            ys = ops.layernorm(x, a['tn'][0], a['tn'][1], out_dtype=ops.SPLIT, eps=LN_EPS_A3)
            t = ops.gemm(ys, a['wg'], out_dtype=f32, a_wrap=2 * E)
            sel = ops.gemm(t, a['wsel'], out_dtype=f32)                          # [B*T, S]: a 27-row fp32 product
            feat = ops.gemm(ys, a['wfeat'], out_dtype=f32, a_wrap=2 * E)
            pooled, attn = ops.a3_pool(sel, feat, B, T, S, want_attn)
            zs = ops.layernorm(pooled, a['n'][0], a['n'][1], out_dtype=ops.SPLIT, eps=LN_EPS_A3)
            if greedy and self.greedy_fused and a['hw'].shape[0] >= self.GREEDY_FUSED_MIN_CLASSES:
                i, p = ops.gemm_row_argmax_prob(zs, a['hw'], a['hb'], a_wrap=2 * E)
                return attn, (i.view(B, S), p.view(B, S))
            logits = ops.gemm(zs, a['hw'], a['hb'], out_dtype=f32, a_wrap=2 * E)
            return attn, (ops.row_argmax_prob_2d(logits, B, S) if greedy else logits.view(B, S, -1))
        y = ops.layernorm(x, a['tn'][0], a['tn'][1], eps=LN_EPS_A3)
        t = ops.gemm(y, a['wg'])
        sel = ops.gemm(t, a['wsel'], out_dtype=torch.float32)                 # [B*T, S] fp32
        feat = ops.gemm(y, a['wfeat'])
        pooled, attn = ops.a3_pool(sel, feat, B, T, S, want_attn)
        z = ops.layernorm(pooled, a['n'][0], a['n'][1], out_dtype=self.engine_dtype, eps=LN_EPS_A3)
        if greedy and self.greedy_fused and a['hw'].shape[0] >= self.GREEDY_FUSED_MIN_CLASSES:
            i, p = ops.gemm_row_argmax_prob(z, a['hw'], a['hb'])
            return attn, (i.view(B, S), p.view(B, S))
        logits = ops.gemm(z, a['hw'], a['hb'], out_dtype=torch.float32)
        return attn, (ops.row_argmax_prob_2d(logits, B, S) if greedy else logits.view(B, S, -1))

    GREEDY_FUSED_MIN_CLASSES = 1024   # heads at least this wide decode greedily from the product's row statistics (the 38-class character head: one tile, nothing to save)

    @torch.no_grad()
    def forward(self, input, is_eval=False):
        if self.training:
            raise NotImplementedError('training is out of scope for the MI355X inference engine')
        if not input.is_cuda:
            raise RuntimeError('MGPSTR runs on MI355X only: pass device tensors; there is no CPU fallback')
        with torch.cuda.device(input.device):
            x, B, T = self.encode(input)
            attens, outs = [], []
            for name in GRANULARITIES:
                a, lg = self._a3_head(x, B, T, name, is_eval)
                attens.append(a)
                outs.append(lg)
        return [attens] + outs if is_eval else outs

    @torch.no_grad()
    def greedy(self, input):
        """The device part of recognition (test_final.py:145-170): -> [(ids int32 [B, S], prob fp32 [B, S])] for char / bpe / wp.  The wide heads
        (BPE 50 257, WordPiece 30 522 classes) decode from the head product's per-tile row statistics -- no logits tensor (greedy_fused)."""
        if self.training:
            raise NotImplementedError('training is out of scope for the MI355X inference engine')
        if not input.is_cuda:
            raise RuntimeError('MGPSTR runs on MI355X only: pass device tensors; there is no CPU fallback')
        with torch.cuda.device(input.device):
            x, B, T = self.encode(input)
            return [self._a3_head(x, B, T, name, False, greedy=True)[1] for name in GRANULARITIES]

    # result decoding (test_final.py:145-240) ---------------------------------------------------------
    @torch.no_grad()
    def recognize(self, input, bpe_vocab=None, wp_vocab=None):
        """-> list of dicts per image: greedy ids of the three granularities (position 0 dropped), their confidences,
        the fused choice (0 char / 1 bpe / 2 wp / -1 none) and the character-level string.  bpe_vocab / wp_vocab: paths of a LOCAL
        GPT-2 `vocab.json` / BERT `vocab.txt` (or utils.mgp_tokens.BpeVocab / WordPieceVocab objects): the records then also
        carry `bpe_text`, `wp_text` and the fused `text` of test_final.py:196-236 (the reference fetches both tokenizers from
        the hub, utils.py:23-24; without the files the engine stops at ids + confidences)."""
        ids, probs = [], []
        for i, p in self.greedy(input):
            ids.append(i[:, 1:].cpu())
            probs.append(p[:, 1:].cpu())
        res = decode_ids(ids, probs)
        if bpe_vocab is not None or wp_vocab is not None:
            from ..utils import mgp_tokens as MT
            bpe = MT.BpeVocab(bpe_vocab) if isinstance(bpe_vocab, str) else bpe_vocab
            wp = MT.WordPieceVocab(wp_vocab) if isinstance(wp_vocab, str) else wp_vocab
            MT.decode_strings(res, bpe, wp)
        return res


def decode_ids(ids, probs):
    """host part of test_final.py:172-240 on the greedy ids / max-softmax probabilities ([B, 26] each)."""
    res = []
    B = ids[0].shape[0]
    for b in range(B):
        s = ''.join(CHAR_TABLE[i] for i in ids[0][b].tolist())
        eos = s.find('[s]')          # the reference uses the STRING index as a token count (test_final.py:176-181)
        conf = []
        pr = probs[0][b][:eos + 1]
        conf.append(float(pr.cumprod(dim=0)[-1]) if pr.numel() else 0.0)
        for k, eos_id in ((1, BPE_EOS), (2, WP_EOS)):
            lst = ids[k][b].tolist()
            e = lst.index(eos_id) if eos_id in lst else -1
            pr = probs[k][b][:e + 1]
            conf.append(float(pr.cumprod(dim=0)[-1]) if pr.numel() else 0.0)
        best, which = 0.0, -1
        for k in range(3):
            if conf[k] > best:
                best, which = conf[k], k
        res.append(dict(char_ids=ids[0][b].tolist(), bpe_ids=ids[1][b].tolist(), wp_ids=ids[2][b].tolist(),
                        char_text=s[:eos], conf=conf, choice=which))
    return res
