"""Checkpoint ingestion: utils/checkpointer.py and model.load_swin_pretrained against the
reference's OWN loaders run on the same files (OCR/OmniParser/utils/checkpointer.py:19-65,
model/backbone/swin_transformer.py:628-658), plus the MGP-STR `module.` prefix (MGP-STR/test_final.py:348-356).
CPU only; the comparisons with the real reference need /root/reference (build container) and are skipped elsewhere,
the format / safety checks run everywhere."""
import argparse
import copy
import os

import pytest
import torch

from advancedliteratemachinery_amd.model import OmniParser, load_swin_pretrained
from advancedliteratemachinery_amd.utils.checkpointer import Checkpointer, load_checkpoint_file
from advancedliteratemachinery_amd.utils.parser import make_args
from advancedliteratemachinery_amd.utils import synthetic
from oracle import ref_import

DEPTHS = (2, 2, 2, 2)
needs_ref = pytest.mark.skipif(not ref_import.available(), reason='reference tree not present')


def _args(**kw):
    a = make_args(tfm_pre_norm=True, use_fpn=True, use_char_window_prompt=True, **kw)
    for k, v in dict(continue_train=False, train_vie=False).items():
        if not hasattr(a, k):
            setattr(a, k, v)
    return a


def _ours(args):
    return OmniParser(args, dict(depths=DEPTHS), engine_dtype='fp32')


def _equal(sd_a, sd_b):
    assert set(sd_a) == set(sd_b)
    for k in sd_a:
        assert torch.equal(sd_a[k].cpu(), sd_b[k].cpu()), k


@pytest.fixture(scope='module')
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp('ckpt')
    args = _args()
    sd = synthetic.make_state_dict(args, seed=3, depths=DEPTHS)
    paths = {}
    paths['final'] = str(d / 'final.pth')                     # checkpointer.py:85-87: {'model': ...} only
    torch.save({'model': sd}, paths['final'])
    paths['bare'] = str(d / 'bare.pth')
    torch.save(sd, paths['bare'])
    paths['step'] = str(d / 'step.pth')                       # checkpointer.py:75-82: everything, incl. the args Namespace
    torch.save({'model': sd, 'optimizer': {'state': {}, 'param_groups': []}, 'lr_scheduler': {'last_epoch': 7}, 'epoch': 5,
                'global_step': 1234, 'args': argparse.Namespace(lr=1e-4, train_dataset=['x'])}, paths['step'])
    # ImageNet Swin file: {'model': un-prefixed backbone keys} + a classifier head the detector does not have
    swin = {k[len('backbone.0.'):]: v for k, v in sd.items() if k.startswith('backbone.0.') and 'norm0' not in k
            and 'norm1.' not in k.split('layers')[0]}
    swin = {k: v for k, v in swin.items() if not k.startswith('norm')}   # the classification Swin has one final `norm`, not norm0..3
    swin['head.weight'] = torch.randn(10, 1024)
    swin['norm.weight'] = torch.ones(1024)
    paths['swin'] = str(d / 'swin_base.pth')
    torch.save({'model': swin}, paths['swin'])
    return args, sd, paths


def test_final_and_bare_and_step_checkpoints(files):
    args, sd, paths = files
    for name, cont, want in (('final', False, (-1, 0)), ('bare', False, (-1, 0)), ('bare', True, (-1, 0)), ('step', False, (-1, 0)),
                             ('step', True, (5, 1234))):
        a = copy.copy(args)
        a.continue_train = cont
        m = _ours(a)
        got = Checkpointer(False).load(paths[name], m, a)
        assert got == want, (name, cont, got)
        _equal(m.state_dict(), sd)


def test_step_checkpoint_loads_without_executing_pickles(files):
    """The 'args' Namespace is allow-listed; an arbitrary object is refused unless the caller opts in."""
    args, sd, paths = files
    ck = load_checkpoint_file(paths['step'])
    assert isinstance(ck['args'], argparse.Namespace) and ck['global_step'] == 1234

    class Evil(object):
        def __reduce__(self):
            return (os.system, ('true',))
    bad = os.path.join(os.path.dirname(paths['step']), 'evil.pth')
    torch.save({'model': sd, 'payload': Evil()}, bad)
    with pytest.raises(RuntimeError):
        load_checkpoint_file(bad)
    assert 'model' in load_checkpoint_file(bad, allow_unsafe_pickle=True)


@needs_ref
def test_matches_reference_checkpointer(files):
    args, sd, paths = files
    RefCk = ref_import.ref_module('utils.checkpointer').Checkpointer
    for name, cont in (('final', False), ('step', False), ('step', True), ('bare', True)):
        a = copy.copy(args)
        a.continue_train = cont
        ref_model = ref_import.build_reference_model(a, synthetic.make_state_dict(a, seed=9, depths=DEPTHS), depths=DEPTHS)
        with torch.serialization.safe_globals([argparse.Namespace]):   # torch >= 2.6 defaults to weights_only=True
            want = RefCk(False).load(paths[name], ref_model, a)
        m = _ours(a)
        got = Checkpointer(False).load(paths[name], m, a)
        assert tuple(got) == tuple(want), (name, cont)
        _equal(m.state_dict(), ref_model.state_dict())


@needs_ref
def test_train_vie_vocabulary_growth_matches_reference(files):
    """--train_vie: a text-spotting checkpoint (V = 1104) into a KIE model (V = 1108): old rows copied, the new class rows
    keep the model's initial values (checkpointer.py:9-17,33-42)."""
    args, sd, paths = files
    a = _args(vie_categories=4, val_dataset=['sroie_val'])
    a.train_vie, a.continue_train = True, False
    init = synthetic.make_state_dict(a, seed=11, depths=DEPTHS)
    RefCk = ref_import.ref_module('utils.checkpointer').Checkpointer
    ref_model = ref_import.build_reference_model(a, init, depths=DEPTHS)
    RefCk(False).load(paths['final'], ref_model, a)
    m = _ours(a)
    m.load_state_dict(init)
    Checkpointer(False).load(paths['final'], m, a)
    got = m.state_dict()
    _equal(got, ref_model.state_dict())
    k = 'transformer.embedding.word_embeddings.weight'
    assert got[k].shape[0] == 1108 and torch.equal(got[k][:1104], sd[k]) and torch.equal(got[k][1104:], init[k][1104:])


@needs_ref
def test_swin_pretrained_matches_reference(files, monkeypatch):
    """Key-intersection load of the ImageNet file: every backbone tensor equals what build_swin_transformer_model gives
    (depths patched to the small test configuration on both sides)."""
    args, sd, paths = files
    swin_mod = ref_import.ref_modules()['swin']
    real = swin_mod.SwinTransformer
    monkeypatch.setattr(swin_mod, 'SwinTransformer', lambda **kw: real(**dict(kw, depths=list(DEPTHS))))
    torch.manual_seed(0)
    ref_swin = swin_mod.build_swin_transformer_model(paths['swin'])
    m = _ours(args)
    base = {k: v.clone() for k, v in m.state_dict().items()}
    n = load_swin_pretrained(m, paths['swin'])
    got = m.state_dict()
    saved = load_checkpoint_file(paths['swin'])['model']
    assert n == len([k for k in ref_swin.state_dict() if k in saved]) and n > 100
    for k, v in ref_swin.state_dict().items():
        if k in saved:
            assert torch.equal(got['backbone.0.' + k], v), k
        else:   # not in the file (norm0..3): untouched on both sides
            assert torch.equal(got['backbone.0.' + k], base['backbone.0.' + k]), k
    for k in got:
        if not k.startswith('backbone.0.'):
            assert torch.equal(got[k], base[k])


def test_mgp_str_dataparallel_prefix():
    """MGP-STR checkpoints are state-dicts of DataParallel(Model): keys `module.mgp_str.*` (test_final.py:343-356)."""
    from advancedliteratemachinery_amd.model.mgp_str import MGPSTR
    c = synthetic.mgp_cfg(depth=1, bpe_vocab=64, wp_vocab=48)
    sd = synthetic.make_mgp_state_dict(c, seed=2)
    m = MGPSTR(c, engine_dtype='fp32')
    m.load_reference_state_dict({'module.' + k: v for k, v in sd.items()})
    own = m.state_dict()
    for k, v in sd.items():
        assert torch.equal(own[k], v), k


def test_checkpoint_io_errors_are_not_reported_as_pickle_refusals(tmp_path):
    """only the weights-only unpickler's allow-list refusal becomes the "allow_unsafe_pickle" error (and only that
    triggers the unsafe second attempt); a missing or truncated file surfaces as what it is."""
    import pytest
    from advancedliteratemachinery_amd.utils.checkpointer import load_checkpoint_file
    with pytest.raises(FileNotFoundError):
        load_checkpoint_file(str(tmp_path / 'missing.pth'), allow_unsafe_pickle=True)
    good = tmp_path / 'good.pth'
    torch.save({'model': {'w': torch.ones(3)}}, str(good))
    data = good.read_bytes()
    (tmp_path / 'cut.pth').write_bytes(data[:len(data) // 2])
    with pytest.raises(Exception) as ei:
        load_checkpoint_file(str(tmp_path / 'cut.pth'))
    assert 'allow_unsafe_pickle' not in str(ei.value)
    bad = tmp_path / 'cfg.pth'      # like the official Swin files: a pickled 'config' object next to the weights
    torch.save({'model': {'w': torch.ones(3)}, 'config': _Refused()}, str(bad))
    with pytest.raises(RuntimeError, match='allow_unsafe_pickle'):
        load_checkpoint_file(str(bad))
    assert torch.equal(load_checkpoint_file(str(bad), allow_unsafe_pickle=True)['model']['w'], torch.ones(3))


class _Refused(object):
    """module-level class: picklable by reference, not on torch's weights-only allow-list"""
    def __init__(self):
        self.v = 3
