"""Model assembly, mirroring the reference's `model` package (OCR/OmniParser/model/__init__.py:7-20)."""
import os

import torch

from .omniparser import OmniParser
from .params import SWIN_B, expected_state_dict


def load_swin_pretrained(model, path, allow_unsafe_pickle=False):
    """ImageNet Swin-B init by key intersection, like build_swin_transformer_model
    (reference backbone/swin_transformer.py:636-656): file = {'model': {un-prefixed keys}}.  The official Swin files also carry
    a pickled 'config' object, which torch.load(weights_only=True) refuses: pass allow_unsafe_pickle (--allow_unsafe_pickle)
    for a file you trust."""
    from ..utils.checkpointer import load_checkpoint_file
    saved = load_checkpoint_file(path, allow_unsafe_pickle)['model']
    own = model.state_dict()
    hit = {}
    for k in own:
        if k.startswith('backbone.0.') and k[len('backbone.0.'):] in saved:
            hit[k] = saved[k[len('backbone.0.'):]]
    own.update(hit)
    model.load_state_dict(own)
    return len(hit)


def build_model(args, swin_cfg=None):
    """Same call as the reference's build_model(args).  The reference requires the ImageNet file
    even for --eval; here it is optional (the fine-tuned checkpoint overwrites it anyway)."""
    if 'swin' not in args.backbone:
        raise NotImplementedError('only the Swin backbone is on the MI355X hot path (reference default)')
    model = OmniParser(args, swin_cfg)
    pf = getattr(args, 'pretrained_file', None)
    if pf and os.path.isfile(pf):
        load_swin_pretrained(model, pf, bool(getattr(args, 'allow_unsafe_pickle', False)))
    if torch.cuda.is_available():
        model = model.to(torch.device('cuda'))
    return model
