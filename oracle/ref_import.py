"""TEST INFRASTRUCTURE -- load the *real* reference OmniParser model classes on CPU.

Only usable where /root/reference exists (the build container).  Used by gen_golden.py to
produce tests/golden/* and by the `not gpu` tests that pin oracle/omniparser_ref.py against the
reference itself.  Nothing on the GPU box imports this module.

The reference needs three helpers from `timm.models.layers` (swin_transformer.py:14) and imports
`torchvision` (utils/nested_tensor.py:2, backbone/resnet.py:2,7); neither package is installed,
so in-memory stub modules stand in for them (DropPath == identity in eval mode).
`build_model` hard-codes .to('cuda') (model/__init__.py:14) and `swin_base` requires a
pretrained file (swin_transformer.py:636), so the model is assembled from the same classes
directly.
"""
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get('OMNIPARSER_REF_ROOT', '/root/reference/OCR/OmniParser')


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'model'))


def _install_stubs():
    if 'timm.models.layers' not in sys.modules:
        timm = types.ModuleType('timm')
        tm = types.ModuleType('timm.models')
        tl = types.ModuleType('timm.models.layers')

        class DropPath(nn.Module):
            def __init__(self, p=0.0):
                super().__init__()

            def forward(self, x):
                return x

        tl.DropPath = DropPath
        tl.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
        tl.trunc_normal_ = nn.init.trunc_normal_
        sys.modules.update({'timm': timm, 'timm.models': tm, 'timm.models.layers': tl})
    if 'torchvision' not in sys.modules:
        tv = types.ModuleType('torchvision')
        tvm = types.ModuleType('torchvision.models')
        tvu = types.ModuleType('torchvision.models._utils')
        tvu.IntermediateLayerGetter = object
        tv.models = tvm
        sys.modules.update({'torchvision': tv, 'torchvision.models': tvm,
                            'torchvision.models._utils': tvu})


_cached = {}


def ref_modules():
    """Import the reference `model`/`utils` packages (they use absolute `utils.*` imports)."""
    if 'mods' in _cached:
        return _cached['mods']
    if not available():
        raise RuntimeError('reference tree not present at %s' % REF_ROOT)
    _install_stubs()
    # the reference packages are called `model` and `utils`; make sure nothing shadows them
    for name in ('model', 'utils'):
        if name in sys.modules and not getattr(sys.modules[name], '__file__', '').startswith(REF_ROOT):
            del sys.modules[name]
    sys.path.insert(0, REF_ROOT)
    try:
        import importlib
        swin = importlib.import_module('model.backbone.swin_transformer')
        joiner = importlib.import_module('model.backbone.joiner')
        backbone = importlib.import_module('model.backbone')
        transformer = importlib.import_module('model.transformer')
        omni = importlib.import_module('model.omniparser')
        nested = importlib.import_module('utils.nested_tensor')
    finally:
        sys.path.remove(REF_ROOT)
    mods = dict(swin=swin, joiner=joiner, backbone=backbone, transformer=transformer,
                omniparser=omni, nested=nested)
    _cached['mods'] = mods
    return mods


def build_reference_model(args, state_dict, embed_dim=128, depths=(2, 2, 18, 2),
                          num_heads=(4, 8, 16, 32), window=7):
    """Assemble reference classes exactly like model/__init__.py:7-13 + backbone/__init__.py:6-23
    (minus the cuda/pretrained-file hard-codes) and load `state_dict` strictly."""
    m = ref_modules()
    swin = m['swin'].SwinTransformer(embed_dim=embed_dim, depths=list(depths),
                                      num_heads=list(num_heads), window_size=window,
                                      drop_path_rate=0.3, use_checkpoint=False)
    pos = m['backbone'].build_position_embedding(args)
    bb = m['joiner'].Joiner(swin, pos)
    tr = m['transformer'].build_transformer(args)
    model = m['omniparser'].OmniParser(bb, tr, args.num_classes, args.use_fpn)
    # EXTENSION (Swin-T and other widths, BASELINE config 1): the reference hard-codes input_proj to 1024 input channels
    # (omniparser.py:13-17).  For a narrower backbone the 1x1 conv -- and nothing else -- is rebuilt with the backbone's
    # width: 'reference classes with patched widths' 
    w = state_dict.get('input_proj.weight')
    if w is not None and w.shape[1] != model.input_proj.in_channels:
        old = model.input_proj
        model.input_proj = nn.Conv2d(w.shape[1], old.out_channels, kernel_size=old.kernel_size, stride=old.stride)
    missing = model.load_state_dict(state_dict, strict=True)
    model.eval()
    return model


def nested(tensors, mask):
    return ref_modules()['nested'].NestedTensor(tensors, mask)


def ref_module(name):
    """Import one more module of the reference tree by its dotted name (e.g. 'utils.checkpointer')."""
    ref_modules()
    import importlib
    sys.path.insert(0, REF_ROOT)
    try:
        return importlib.import_module(name)
    finally:
        sys.path.remove(REF_ROOT)
