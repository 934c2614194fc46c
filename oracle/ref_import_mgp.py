"""TEST INFRASTRUCTURE -- run the REAL reference MGP-STR model code on CPU as far as this container allows.

/root/reference/OCR/MGP-STR/modules/mgp_str.py subclasses timm's VisionTransformer (timm==0.4.12, not vendored,
not installed).  What CAN run unmodified is everything MGP-STR itself wrote: `MGPSTR.__init__` / `reset_classifier` /
`forward_features` / `forward` (mgp_str.py:46-101) and `TokenLearner` (token_learner.py).  This module installs a
stand-in `timm.models.vision_transformer.VisionTransformer` base class that only provides the attributes those
methods use (patch_embed, cls_token, pos_embed, pos_drop, blocks, norm, head, embed_dim); its blocks are
`transformers.models.mgp_str.modeling_mgp_str.MgpstrLayer` -- the ViT block of the MGP-STR authors' own port to
`transformers` (same parameter names as timm's Block), with LayerNorm eps = 1e-6 as in timm 0.4.12.
So: wiring, A^3 modules, heads = the reference's own code; block internals = the authors' port (third party).
Only usable where /root/reference exists."""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

MGP_ROOT = os.environ.get('MGPSTR_REF_ROOT', '/root/reference/OCR/MGP-STR')


def available():
    if not os.path.isdir(os.path.join(MGP_ROOT, 'modules')):
        return False
    try:
        importlib.import_module('transformers.models.mgp_str.modeling_mgp_str')
    except Exception:  # noqa: BLE001
        return False
    return True


class _Cfg(object):
    """the few MgpstrConfig fields MgpstrLayer reads"""

    def __init__(self, E, H, mlp_ratio):
        self.hidden_size, self.num_attention_heads, self.mlp_ratio = E, H, mlp_ratio
        self.layer_norm_eps, self.qkv_bias = 1e-6, True
        self.drop_rate = self.attn_drop_rate = 0.0


def _install():
    hf = importlib.import_module('transformers.models.mgp_str.modeling_mgp_str')

    class _PatchEmbed(nn.Module):
        def __init__(self, in_chans, E, patch):
            super().__init__()
            self.proj = nn.Conv2d(in_chans, E, kernel_size=patch, stride=patch)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class _Block(nn.Module):
        """MgpstrLayer returns (hidden, attention probabilities); timm's Block returns hidden only"""

        def __init__(self, cfg):
            super().__init__()
            inner = hf.MgpstrLayer(cfg, drop_path=None)
            for name in ('norm1', 'attn', 'norm2', 'mlp'):
                setattr(self, name, getattr(inner, name))
            self._inner = [inner]

        def forward(self, x):
            return self._inner[0](x)[0]

    class VisionTransformer(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                     num_heads=12, mlp_ratio=4., qkv_bias=True, **kw):
            super().__init__()
            self.num_classes, self.embed_dim, self.num_features = num_classes, embed_dim, embed_dim
            T = (img_size[0] // patch_size) * (img_size[1] // patch_size)
            self.patch_embed = _PatchEmbed(in_chans, embed_dim, patch_size)
            self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
            self.pos_embed = nn.Parameter(torch.zeros(1, T + 1, embed_dim))
            self.pos_drop = nn.Identity()
            cfg = _Cfg(embed_dim, num_heads, mlp_ratio)
            self.blocks = nn.ModuleList([_Block(cfg) for _ in range(depth)])
            self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
            self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def get(name):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
        return sys.modules[name]
    timm, tm = get('timm'), get('timm.models')
    vt, reg = get('timm.models.vision_transformer'), get('timm.models.registry')
    vt.VisionTransformer, vt._cfg = VisionTransformer, (lambda **kw: dict(kw))
    reg.register_model = lambda fn: fn
    tm.create_model = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('create_model is not available in the stub'))
    timm.models, tm.vision_transformer, tm.registry = tm, vt, reg


_cached = {}


def ref_module():
    if 'm' not in _cached:
        if not available():
            raise RuntimeError('MGP-STR reference / transformers port not available')
        _install()
        if 'modules' in sys.modules and not getattr(sys.modules['modules'], '__file__', '').startswith(MGP_ROOT):
            del sys.modules['modules']
        sys.path.insert(0, MGP_ROOT)
        try:
            _cached['m'] = importlib.import_module('modules.mgp_str')
            _cached['tl'] = importlib.import_module('modules.token_learner')
        finally:
            sys.path.remove(MGP_ROOT)
    return _cached['m']


def token_learner_class():
    ref_module()
    return _cached['tl'].TokenLearner


def build_reference_model(c, state_dict, prefix='mgp_str.'):
    """The reference's MGPSTR (mgp_str_base_patch4_3_32_128 arguments, mgp_str.py:190-194) on the stand-in base,
    classifier reset as create_mgp_str does (:43), weights loaded strictly."""
    m = ref_module()
    model = m.MGPSTR(c['max_len'], img_size=c['img'], patch_size=c['patch'], embed_dim=c['embed'], depth=c['depth'],
                     num_heads=c['heads'], mlp_ratio=c['mlp_ratio'], qkv_bias=True, num_classes=c['num_class'], in_chans=3)
    model.reset_classifier(num_classes=c['num_class'])
    own = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
    own = {k: v for k, v in own.items() if not k.startswith('head.')}   # reset_classifier leaves timm's head in place
    missing, unexpected = model.load_state_dict(own, strict=False)
    missing = [k for k in missing if not k.startswith('head.') and '_inner' not in k]
    if missing or unexpected:
        raise RuntimeError('state dict mismatch: missing %s unexpected %s' % (missing, unexpected))
    return model.eval()
