"""Run the oracle against the real reference classes where /root/reference exists (build
container only; skipped on the GPU box)."""
import pytest
import torch

from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason='reference tree not present')


def test_parser_flags_match_reference():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        'ref_parser', os.path.join(ref_import.REF_ROOT, 'utils', 'parser.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from advancedliteratemachinery_amd.utils import parser as mine
    ref = mod.DefaultParser().parser
    me = mine.DefaultParser().parser
    ref_actions = {a.dest: a for a in ref._actions}
    my_actions = {a.dest: a for a in me._actions}
    for dest, a in ref_actions.items():
        assert dest in my_actions, dest
        assert my_actions[dest].default == a.default, dest
        assert my_actions[dest].option_strings == a.option_strings


def test_fresh_case_against_reference():
    """A case that is NOT in the golden set: different size, seed and instance count."""
    from oracle import gen_golden as G, omniparser_ref as O
    from advancedliteratemachinery_amd.utils import synthetic as weights
    from advancedliteratemachinery_amd.utils.parser import make_args
    torch.set_num_threads(8)
    args = make_args(tfm_pre_norm=True, use_fpn=True, use_char_window_prompt=True,
                     pt_seq_length=4)
    sd = weights.make_state_dict(args, seed=3, depths=(2, 2, 2, 2))
    model = ref_import.build_reference_model(args, sd, depths=(2, 2, 2, 2))
    img = torch.randn(1, 3, 97, 141, generator=torch.Generator().manual_seed(7))
    mask = torch.zeros(1, 97, 141, dtype=torch.bool)
    seqs = O.default_prompts(args)
    with torch.no_grad():
        ref = model(ref_import.nested(img, mask), seqs)
        out = O.forward(sd, args, img, mask, seqs, depths=(2, 2, 2, 2))
    for a, b in zip(ref[0], out[0]):
        assert torch.equal(a, b)
    assert (ref[1][0] - out[1][0]).abs().max() < 1e-5
