"""CPU simulation (oracle, fp32): which cross-attention operands of the parity engine may be stored / used as ONE bf16 plane instead
of a split pair [hi | lo]?  Teacher-forced logits of the three decoders on a fixture's own sequences with K, V (memory projections),
q or P (softmax probabilities) rounded to bf16 inside the cross-attention only; everything else fp32.  `sharp` multiplies the
cross-attention scores (peaky attention, as a trained checkpoint has; random-init attention is diffuse and averages V errors away).
    python tools/sim_cross_planes.py [fixture] [sharp]  ->  max |delta logit| per variant (north_star gate: 1e-3)"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import omniparser_ref as O  # noqa: E402
from oracle.gen_golden import CASES, BIG_CASES, case_inputs  # noqa: E402

ROUND = dict(k=False, v=False, q=False, p=False)
SHARP = [1.0]
_bf = lambda t: t.to(torch.bfloat16).float()   # noqa: E731
_mha0 = O.mha


def mha(sd, p, q_in, k_in, v_in, nH, attn_mask=None, key_padding_mask=None):
    if 'multihead_attn' not in p:
        return _mha0(sd, p, q_in, k_in, v_in, nH, attn_mask, key_padding_mask)
    E = q_in.shape[-1]
    hd = E // nH
    W, bvec = sd[p + 'in_proj_weight'], sd[p + 'in_proj_bias']
    q = F.linear(q_in, W[:E], bvec[:E])
    k = F.linear(k_in, W[E:2 * E], bvec[E:2 * E])
    v = F.linear(v_in, W[2 * E:], bvec[2 * E:])
    L, N, _ = q.shape
    S = k.shape[0]
    q = q.reshape(L, N, nH, hd).permute(1, 2, 0, 3) * (SHARP[0] / math.sqrt(hd))
    k = k.reshape(S, N, nH, hd).permute(1, 2, 0, 3)
    v = v.reshape(S, N, nH, hd).permute(1, 2, 0, 3)
    if ROUND['q']:
        q = _bf(q)
    if ROUND['k']:
        k = _bf(k)
    if ROUND['v']:
        v = _bf(v)
    att = q @ k.transpose(-2, -1)
    if key_padding_mask is not None:
        att = att.masked_fill(key_padding_mask[:, None, None, :], float('-inf'))
    att = att.softmax(-1)
    if ROUND['p']:
        att = _bf(att)
    out = (att @ v).permute(2, 0, 1, 3).reshape(L, N, E)
    return F.linear(out, sd[p + 'out_proj.weight'], sd[p + 'out_proj.bias'])


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'spot_224'
    SHARP[0] = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    torch.set_num_threads(8)
    case = CASES[name] if name in CASES else BIG_CASES[name]
    args, sd, img, mask, seqs = case_inputs(case)
    gold = torch.load(os.path.join(ROOT, 'tests', 'golden', name + '.pt'), weights_only=False)
    with torch.no_grad():
        enc = O.encode(sd, args, img, mask, depths=case['depths'])
        memory, pos, kpm = enc['memory'], enc['pos'], enc['mask']
        O.mha = mha
        tf = gold['tf']
        base = {}
        rows = []
        for tag, rnd in (('fp32', ''), ('V', 'v'), ('P', 'p'), ('V+P', 'vp'), ('K', 'k'), ('q', 'q'), ('K+q', 'kq'), ('all four', 'kvqp')):
            for key in ROUND:
                ROUND[key] = key in rnd
            errs, entropy = [], []
            for kind in ('pt', 'poly', 'rec'):
                lg = O.decode(sd, args, tf[kind + '_in'], memory, kpm, pos, kind)
                if tag == 'fp32':
                    base[kind] = lg
                errs.append((lg - base[kind]).abs().max().item())
            rows.append((tag, errs))
        print('%s  M=%d  sharp=%.1f  max|logit|=%.1f' % (name, memory.shape[0], SHARP[0], max(b.abs().max().item() for b in base.values())))
        for tag, errs in rows:
            print('  bf16 %-9s: max |dlogit| pt %.2e  poly %.2e  rec %.2e' % (tag, *errs))


if __name__ == '__main__':
    main()
