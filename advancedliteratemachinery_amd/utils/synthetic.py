"""Procedural (seeded) checkpoints in the reference's state-dict layouts: the synthetic weights bench.py, the smoke
test and the parity tests run on.  Data generation only -- no model arithmetic lives here.

No OmniParser checkpoint exists offline and a Swin-B state-dict is 577 MB, so golden fixtures
cannot carry weights.  Instead every tensor of the reference's 610-key state-dict layout
(OCR/OmniParser: model/backbone/swin_transformer.py:479-590, model/fpn.py:16-19,
model/omniparser.py:13-17, model/transformer.py:20-37,289-300,383-396, model/block/mlp.py:5-9)
is generated from (key name, shape, seed) alone with a per-key torch.Generator.  The real
reference (in the build container), this repo's CPU oracle and the HIP engine (on the GPU box)
therefore all load bit-identical weights without shipping them.

Values are deliberately *not* the default init: LayerNorm gains/biases, all Linear biases and the
relative-position-bias table are non-trivial so every term of the forward pass is exercised, and
the last layer of each prediction head is scaled by `head_gain` so greedy argmax margins are wide
enough for token-exact comparisons (SURVEY.md section 7 "hard parts").
"""
import math
import re
import zlib
from collections import OrderedDict

import torch

_POST_ACT = re.compile(r'(pred_layer\.layers\.[12]\.weight|linear2\.weight|mlp\.fc2\.weight)$')

SWIN_B = dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window=7, mlp_ratio=4)


def relative_position_index(ws=7):
    """(ws*ws, ws*ws) int64 lookup into the (2ws-1)^2 bias table.

    reference swin_transformer.py:97-108: index = (dy + ws-1) * (2ws-1) + (dx + ws-1)."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing='ij')
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + ws - 1
    dx = xs[:, None] - xs[None, :] + ws - 1
    return (dy * (2 * ws - 1) + dx).long()


def state_dict_spec(args, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32),
                    window=7, mlp_ratio=4):
    """Ordered {key: shape} for the reference model built with `--use_fpn` (FPN present)
    or without it (no fpn.* keys; input_proj stride 1)."""
    d = args.tfm_hidden_dim
    ff = args.tfm_dim_feedforward
    V = args.num_classes
    spec = OrderedDict()
    bb = 'backbone.0.'
    spec[bb + 'patch_embed.proj.weight'] = (embed_dim, 3, 4, 4)
    spec[bb + 'patch_embed.proj.bias'] = (embed_dim,)
    spec[bb + 'patch_embed.norm.weight'] = (embed_dim,)
    spec[bb + 'patch_embed.norm.bias'] = (embed_dim,)
    for s, (dep, nh) in enumerate(zip(depths, num_heads)):
        C = embed_dim * 2 ** s
        for b in range(dep):
            p = f'{bb}layers.{s}.blocks.{b}.'
            spec[p + 'norm1.weight'] = (C,)
            spec[p + 'norm1.bias'] = (C,)
            spec[p + 'attn.relative_position_bias_table'] = ((2 * window - 1) ** 2, nh)
            spec[p + 'attn.relative_position_index'] = (window * window, window * window)
            spec[p + 'attn.qkv.weight'] = (3 * C, C)
            spec[p + 'attn.qkv.bias'] = (3 * C,)
            spec[p + 'attn.proj.weight'] = (C, C)
            spec[p + 'attn.proj.bias'] = (C,)
            spec[p + 'norm2.weight'] = (C,)
            spec[p + 'norm2.bias'] = (C,)
            spec[p + 'mlp.fc1.weight'] = (mlp_ratio * C, C)
            spec[p + 'mlp.fc1.bias'] = (mlp_ratio * C,)
            spec[p + 'mlp.fc2.weight'] = (C, mlp_ratio * C)
            spec[p + 'mlp.fc2.bias'] = (C,)
        if s < len(depths) - 1:
            p = f'{bb}layers.{s}.downsample.'
            spec[p + 'reduction.weight'] = (2 * C, 4 * C)
            spec[p + 'norm.weight'] = (4 * C,)
            spec[p + 'norm.bias'] = (4 * C,)
    for s in range(len(depths)):
        C = embed_dim * 2 ** s
        spec[f'{bb}norm{s}.weight'] = (C,)
        spec[f'{bb}norm{s}.bias'] = (C,)
    tr = 'transformer.'
    spec[tr + 'embedding.word_embeddings.weight'] = (V, d)
    for t in ('pt', 'poly', 'rec', 'other'):
        spec[tr + f'embedding.{t}_position_embeddings.weight'] = (1024, d)
    spec[tr + 'embedding.LayerNorm.weight'] = (d,)
    spec[tr + 'embedding.LayerNorm.bias'] = (d,)
    for dec in ('pt', 'poly', 'rec'):
        for l in range(args.tfm_dec_layers):
            p = f'{tr}{dec}_decoder.layers.{l}.'
            for att in ('self_attn', 'multihead_attn'):
                spec[p + att + '.in_proj_weight'] = (3 * d, d)
                spec[p + att + '.in_proj_bias'] = (3 * d,)
                spec[p + att + '.out_proj.weight'] = (d, d)
                spec[p + att + '.out_proj.bias'] = (d,)
            spec[p + 'linear1.weight'] = (ff, d)
            spec[p + 'linear1.bias'] = (ff,)
            spec[p + 'linear2.weight'] = (d, ff)
            spec[p + 'linear2.bias'] = (d,)
            for n in ('norm3', 'norm1', 'norm2'):
                spec[p + n + '.weight'] = (d,)
                spec[p + n + '.bias'] = (d,)
        spec[f'{tr}{dec}_decoder.norm.weight'] = (d,)
        spec[f'{tr}{dec}_decoder.norm.bias'] = (d,)
    for dec in ('pt', 'poly', 'rec'):
        dims = [(d, d), (d, d), (V, d)]
        for i, (o, k) in enumerate(dims):
            spec[f'{tr}{dec}_pred_layer.layers.{i}.weight'] = (o, k)
            spec[f'{tr}{dec}_pred_layer.layers.{i}.bias'] = (o,)
    if args.use_fpn:
        chans = [embed_dim * 2 ** s for s in range(len(depths))]
        for i, cin in enumerate(reversed(chans)):
            spec[f'fpn.fpn_in.{i}.weight'] = (256, cin, 1, 1)
        spec['input_proj.weight'] = (d, 1024, 1, 1)
    else:
        spec['input_proj.weight'] = (d, embed_dim * 2 ** (len(depths) - 1), 1, 1)
    spec['input_proj.bias'] = (d,)
    return spec


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) * 2654435761 + seed * 97 + 12345) % (2 ** 63 - 1))
    return g


def _fans(shape):
    rf = 1
    for s in shape[2:]:
        rf *= s
    return shape[1] * rf, shape[0] * rf


def make_state_dict(args, seed=0, head_gain=16.0, window=7, **swin):
    """Deterministic fp32 state-dict in the reference layout (see module docstring)."""
    cfg = dict(SWIN_B)
    cfg.update(swin)
    spec = state_dict_spec(args, cfg['embed_dim'], cfg['depths'], cfg['num_heads'],
                           cfg['window'], cfg['mlp_ratio'])
    sd = OrderedDict()
    shared_norm = None
    for key, shape in spec.items():
        g = _gen(key, seed)
        leaf = key.rsplit('.', 1)[-1]
        if key.endswith('relative_position_index'):
            t = relative_position_index(cfg['window'])
        elif key.endswith('relative_position_bias_table'):
            t = 0.3 * torch.randn(shape, generator=g)
        elif 'position_embeddings' in key:
            t = torch.randn(shape, generator=g)
        elif 'word_embeddings' in key:
            t = torch.randn(shape, generator=g)
            t[args.padding_index].zero_()  # nn.Embedding(padding_idx=...) keeps this row zero
        elif len(shape) == 1:
            is_norm = ('norm' in key.lower()) and leaf == 'weight'
            if is_norm:
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            elif leaf in ('bias', 'in_proj_bias'):
                t = 0.05 * torch.randn(shape, generator=g)
            else:
                raise KeyError(key)
        else:
            fi, fo = _fans(shape)
            bound = math.sqrt(6.0 / (fi + fo))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
            if _POST_ACT.search(key):
                # consumers of post-ReLU/GELU features: zero-mean rows, so the (positive) mean
                # activation does not map to a large input-independent offset that would make
                # every greedy step pick the same token (weak test signal).
                t = t - t.mean(dim=1, keepdim=True)
            if 'pred_layer.layers.2.weight' in key:
                t = t * head_gain
        # the reference shares ONE final LayerNorm module between the three decoders
        # (transformer.py:24-33) -> the three key pairs alias the same tensor values.
        if key.endswith('_decoder.norm.weight') or key.endswith('_decoder.norm.bias'):
            kind = key.rsplit('.', 1)[-1]
            if shared_norm is None:
                shared_norm = {}
            if kind not in shared_norm:
                shared_norm[kind] = t
            t = shared_norm[kind]
        sd[key] = t.contiguous()
    return sd


# ---------------------------------------------------------------------------------------------
# MGP-STR (OCR/MGP-STR): timm VisionTransformer keys + the three TokenLearners and heads (models.py:36,
# modules/mgp_str.py:46-61, modules/token_learner.py:13-19)
# ---------------------------------------------------------------------------------------------
MGP_BPE_VOCAB, MGP_WP_VOCAB = 50257, 30522
MGP_BASE = dict(embed=768, depth=12, heads=12, mlp_ratio=4, img=(32, 128), patch=4, max_len=27, num_class=38)


def mgp_cfg(**over):
    c = dict(MGP_BASE)
    c.update(over)
    return c


def make_mgp_state_dict(c, seed=0, prefix='mgp_str.'):
    """Seeded procedural checkpoint in the reference's key layout (Model.mgp_str.*, models.py:36): timm
    VisionTransformer keys + the three TokenLearners and heads.  Includes timm's unused `norm` / `head`."""
    g = torch.Generator().manual_seed(seed)
    E, D, L, V = c['embed'], c['depth'], c['max_len'], c['num_class']
    Hd = int(E * c['mlp_ratio'])
    T = (c['img'][0] // c['patch']) * (c['img'][1] // c['patch']) + 1
    r = lambda *s, std=0.02: torch.randn(*s, generator=g) * std          # noqa: E731
    sd = {}
    p = prefix
    sd[p + 'cls_token'] = r(1, 1, E)
    sd[p + 'pos_embed'] = r(1, T, E)
    sd[p + 'patch_embed.proj.weight'] = r(E, 3, c['patch'], c['patch'], std=0.1)
    sd[p + 'patch_embed.proj.bias'] = r(E, std=0.05)
    for i in range(D):
        b = '%sblocks.%d.' % (p, i)
        sd[b + 'norm1.weight'] = 1 + r(E, std=0.05); sd[b + 'norm1.bias'] = r(E, std=0.05)
        sd[b + 'attn.qkv.weight'] = r(3 * E, E, std=1.5 / math.sqrt(E)); sd[b + 'attn.qkv.bias'] = r(3 * E, std=0.05)
        sd[b + 'attn.proj.weight'] = r(E, E, std=0.7 / math.sqrt(E)); sd[b + 'attn.proj.bias'] = r(E, std=0.05)
        sd[b + 'norm2.weight'] = 1 + r(E, std=0.05); sd[b + 'norm2.bias'] = r(E, std=0.05)
        sd[b + 'mlp.fc1.weight'] = r(Hd, E, std=1.0 / math.sqrt(E)); sd[b + 'mlp.fc1.bias'] = r(Hd, std=0.05)
        sd[b + 'mlp.fc2.weight'] = r(E, Hd, std=0.7 / math.sqrt(Hd)); sd[b + 'mlp.fc2.bias'] = r(E, std=0.05)
    sd[p + 'norm.weight'] = torch.ones(E); sd[p + 'norm.bias'] = torch.zeros(E)          # unused by MGPSTR.forward
    sd[p + 'head.weight'] = r(V, E); sd[p + 'head.bias'] = torch.zeros(V)                 # timm's head, unused
    for name, vocab in (('char', V), ('bpe', c.get('bpe_vocab', MGP_BPE_VOCAB)),
                        ('wp', c.get('wp_vocab', MGP_WP_VOCAB))):
        t = '%s%s_tokenLearner.' % (p, name)
        sd[t + 'token_norm.weight'] = 1 + r(E, std=0.05); sd[t + 'token_norm.bias'] = r(E, std=0.05)
        sd[t + 'tokenLearner.0.weight'] = r(E, E // 8, 1, 1, std=1.0 / math.sqrt(E // 8))
        sd[t + 'tokenLearner.1.weight'] = r(L, E, 1, 1, std=2.0 / math.sqrt(E))
        sd[t + 'feat.weight'] = r(E, E // 8, 1, 1, std=1.0 / math.sqrt(E // 8))
        sd[t + 'norm.weight'] = 1 + r(E, std=0.05); sd[t + 'norm.bias'] = r(E, std=0.05)
        sd['%s%s_head.weight' % (p, name)] = r(vocab, E, std=3.0 / math.sqrt(E))
        sd['%s%s_head.bias' % (p, name)] = r(vocab, std=0.1)
    return sd


