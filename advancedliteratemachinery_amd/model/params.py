"""State-dict layout of the reference OmniParser model and a parameter container that reproduces it.

The drop-in contract is the reference's checkpoint format: 610 tensors whose
keys come from the reference module tree (OCR/OmniParser: model/omniparser.py:8-17,
model/backbone/joiner.py:6-8 -> 'backbone.0.*', model/backbone/swin_transformer.py:479-590,
model/fpn.py:16-19, model/transformer.py:20-37,289-300,383-396, model/block/mlp.py:5-9).
`attach_parameters` builds bare nn.Module containers along those dotted paths so that
`state_dict()`, `load_state_dict()` (strict), `.to()`, `.eval()` behave like the reference model.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

SWIN_B = dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window=7, mlp_ratio=4)
MAX_POSITION_EMBEDDINGS = 1024  # reference: build_transformer, transformer.py:475


def expected_state_dict(args, swin=None):
    """OrderedDict key -> (shape, dtype) for the model `build_model(args)` would create."""
    cfg = dict(SWIN_B)
    cfg.update(swin or {})
    E, depths, heads, ws, mr = cfg['embed_dim'], cfg['depths'], cfg['num_heads'], cfg['window'], cfg['mlp_ratio']
    d, ff, V, L = args.tfm_hidden_dim, args.tfm_dim_feedforward, args.num_classes, args.tfm_dec_layers
    f32, i64 = torch.float32, torch.int64
    spec = OrderedDict()

    def put(k, shape, dtype=f32):
        spec[k] = (tuple(shape), dtype)

    def lin(prefix, out_f, in_f, bias=True):
        put(prefix + '.weight', (out_f, in_f))
        if bias:
            put(prefix + '.bias', (out_f,))

    def norm(prefix, c):
        put(prefix + '.weight', (c,))
        put(prefix + '.bias', (c,))

    bb = 'backbone.0.'
    put(bb + 'patch_embed.proj.weight', (E, 3, 4, 4))
    put(bb + 'patch_embed.proj.bias', (E,))
    norm(bb + 'patch_embed.norm', E)
    for s, (dep, nh) in enumerate(zip(depths, heads)):
        C = E << s
        for b in range(dep):
            p = '%slayers.%d.blocks.%d.' % (bb, s, b)
            norm(p + 'norm1', C)
            put(p + 'attn.relative_position_bias_table', ((2 * ws - 1) ** 2, nh))
            put(p + 'attn.relative_position_index', (ws * ws, ws * ws), i64)
            lin(p + 'attn.qkv', 3 * C, C)
            lin(p + 'attn.proj', C, C)
            norm(p + 'norm2', C)
            lin(p + 'mlp.fc1', mr * C, C)
            lin(p + 'mlp.fc2', C, mr * C)
        if s + 1 < len(depths):
            p = '%slayers.%d.downsample.' % (bb, s)
            lin(p + 'reduction', 2 * C, 4 * C, bias=False)
            norm(p + 'norm', 4 * C)
    for s in range(len(depths)):
        norm('%snorm%d' % (bb, s), E << s)
    tr = 'transformer.'
    put(tr + 'embedding.word_embeddings.weight', (V, d))
    for kind in ('pt', 'poly', 'rec', 'other'):
        put('%sembedding.%s_position_embeddings.weight' % (tr, kind), (MAX_POSITION_EMBEDDINGS, d))
    norm(tr + 'embedding.LayerNorm', d)
    for kind in ('pt', 'poly', 'rec'):
        for l in range(L):
            p = '%s%s_decoder.layers.%d.' % (tr, kind, l)
            for att in ('self_attn', 'multihead_attn'):
                put(p + att + '.in_proj_weight', (3 * d, d))
                put(p + att + '.in_proj_bias', (3 * d,))
                lin(p + att + '.out_proj', d, d)
            lin(p + 'linear1', ff, d)
            lin(p + 'linear2', d, ff)
            for n in ('norm3', 'norm1', 'norm2'):
                norm(p + n, d)
        norm('%s%s_decoder.norm' % (tr, kind), d)
    for kind in ('pt', 'poly', 'rec'):
        p = '%s%s_pred_layer.layers.' % (tr, kind)
        lin(p + '0', d, d)
        lin(p + '1', d, d)
        lin(p + '2', V, d)
    if args.use_fpn:
        chans = [E << s for s in range(len(depths))]
        for i, cin in enumerate(reversed(chans)):
            put('fpn.fpn_in.%d.weight' % i, (256, cin, 1, 1))
        put('input_proj.weight', (d, 1024, 1, 1))
    else:
        put('input_proj.weight', (d, E << (len(depths) - 1), 1, 1))
    put('input_proj.bias', (d,))
    return spec


def relative_position_index(ws=7):
    """Lookup from (token i, token j) of a window to a row of the bias table:
    (yi - yj + ws-1) * (2ws-1) + (xi - xj + ws-1)   (reference swin_transformer.py:97-108)."""
    t = torch.arange(ws * ws)
    y, x = t // ws, t % ws
    return ((y[:, None] - y[None, :] + ws - 1) * (2 * ws - 1) + (x[:, None] - x[None, :] + ws - 1)).long()


class _Node(nn.Module):
    """Bare container; only exists so dotted state-dict keys match the reference."""


def attach_parameters(root, spec, shared=()):
    """Create Parameters/buffers under `root` following the dotted keys of `spec`.
    `shared`: groups of keys that must alias ONE tensor (the reference shares the final decoder
    LayerNorm between its three decoders, transformer.py:24-33)."""
    alias = {}
    for group in shared:
        for k in group[1:]:
            alias[k] = group[0]
    made = {}
    for key, (shape, dtype) in spec.items():
        parts = key.split('.')
        node = root
        for name in parts[:-1]:
            if name not in node._modules:
                node.add_module(name, _Node())
            node = node._modules[name]
        leaf = parts[-1]
        if key in alias and alias[key] in made:
            node.register_parameter(leaf, made[alias[key]])
            continue
        if dtype == torch.int64:
            node.register_buffer(leaf, relative_position_index(int(round(shape[0] ** 0.5))))
        else:
            p = nn.Parameter(torch.zeros(shape, dtype=dtype), requires_grad=False)
            node.register_parameter(leaf, p)
            made[key] = p
