"""TEST INFRASTRUCTURE -- generate tests/golden/mgp_str_base.pt from the REAL reference MGP-STR code
(oracle/ref_import_mgp.py: the reference's MGPSTR.forward_features / TokenLearner on the authors' `transformers`
ViT blocks).  Run in the build container (needs /root/reference):   python -m oracle.gen_golden_mgp

The fixture stores the seeds (weights are procedural: oracle.mgp_str_ref.make_state_dict), the input images and
REDUCED reference outputs -- enough to pin every stage without shipping 10 MB of logits:
  enc_proj   encoder output projected on 16 seeded directions        [B, 257, 16]
  attens     the three A^3 attention maps                              3 x [B, 27, 257]
  char       character logits, complete                                [B, 27, 38]
  bpe_cols / wp_cols + bpe_sub / wp_sub   logits at 64 seeded vocabulary columns
  *_ids / *_prob   greedy ids and max-softmax probabilities of the three heads (what result decoding consumes)
"""
import os

import torch
import torch.nn.functional as F

from . import mgp_str_ref as R
from . import ref_import_mgp as I

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'mgp_str_base.pt')
SEED_W, SEED_X, SEED_P = 11, 12, 13


def reduce_outputs(x, attens, ch, bp, wp):
    g = torch.Generator().manual_seed(SEED_P)
    proj = torch.randn(x.shape[-1], 16, generator=g) / x.shape[-1] ** 0.5
    bcols = torch.randperm(bp.shape[-1], generator=g)[:64]
    wcols = torch.randperm(wp.shape[-1], generator=g)[:64]
    out = dict(enc_proj=x @ proj, attens=[a.clone() for a in attens], char=ch.clone(), bpe_cols=bcols, wp_cols=wcols,
               bpe_sub=bp[..., bcols].clone(), wp_sub=wp[..., wcols].clone())
    for name, lg in (('char', ch), ('bpe', bp), ('wp', wp)):
        out[name + '_ids'] = lg.argmax(-1)
        out[name + '_prob'] = F.softmax(lg, dim=2).max(dim=2)[0]
    return out


def main():
    c = R.cfg()
    sd = R.make_state_dict(c, seed=SEED_W)
    model = I.build_reference_model(c, sd)
    img = torch.rand(2, 3, 32, 128, generator=torch.Generator().manual_seed(SEED_X)) * 2 - 1   # test_final normalises to [-1, 1]
    with torch.no_grad():
        # the reference's own forward, plus its encoder output via the same code path (forward_features up to the blocks)
        attens, ch, bp, wp = model(img, is_eval=True)
        x = model.patch_embed(img)
        x = torch.cat((model.cls_token.expand(2, -1, -1), x), dim=1) + model.pos_embed
        for blk in model.blocks:
            x = blk(x)
    fix = dict(cfg=c, seed_w=SEED_W, img=img, ref=reduce_outputs(x, attens, ch, bp, wp))
    torch.save(fix, OUT)
    print('wrote', OUT, os.path.getsize(OUT) // 1024, 'KiB')


if __name__ == '__main__':
    main()
