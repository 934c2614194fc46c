"""OmniParser: drop-in for the reference model class on the inference path.

Same constructor-by-args, same state-dict layout, same `forward(samples, sequence)` contract as
OCR/OmniParser/model/omniparser.py:7-32 (eval mode), with the computation executed by
libomp355 (hand-written gfx950 kernels).  Additions over the reference:
  * batches of B > 1 images (the reference asserts B == 1, engine/val.py:22): every image is
    decoded exactly as if it had been submitted alone; `forward` then returns a list of B results;
  * `forced_instances`: fixed-length decoding for throughput measurement with random weights.

Return value per image (reference transformer.py:240-246,286):
  text spotting : ([pt (1,2N), poly (1,32N), rec (1,N,rec_length)] int64, [probs (N,rec_length)])
  KIE           : list of (text, class_name, prob, [rects])
  no points     : None
"""
import os

import torch
import torch.nn as nn

from .. import ops
from ..utils.env import env_flag, env_int
from ..utils.nested_tensor import NestedTensor
from . import params
from .backbone import Encoder
from .transformer import Decoder, index2class

# 'bf16x3' = the parity engine: fp32 storage / fp32 non-GEMM kernels, large products as three bf16 products of split
# operands on the bf16 matrix cores (model/backbone.py, include/omp355.h omp_gemm_args.a_wrap)
_DTYPES = {'bf16': torch.bfloat16, 'fp32': torch.float32, 'bf16x3': 'bf16x3', torch.bfloat16: torch.bfloat16,
           torch.float32: torch.float32}


def _image_sizes(sizes, B):
    """orig_size of seqs[3] (engine/val.py:33): a (2,) tensor, a (B,2) tensor or a list of B of them."""
    if isinstance(sizes, (list, tuple)) and len(sizes) == B and not isinstance(sizes[0], (int, float)):
        t = torch.stack([torch.as_tensor(s).reshape(2) for s in sizes])
    else:
        t = torch.as_tensor(sizes).reshape(-1, 2)
    if t.shape[0] == 1 and B > 1:
        t = t.expand(B, 2)
    return t.cpu()


class OmniParser(nn.Module):
    def __init__(self, args, swin_cfg=None, engine_dtype=None):
        super().__init__()
        self.args = args
        self.swin_cfg = dict(params.SWIN_B)
        self.swin_cfg.update(swin_cfg or {})
        self.use_fpn = bool(args.use_fpn)
        self.engine_dtype = _DTYPES[engine_dtype or getattr(args, 'engine_dtype', 'bf16')]
        spec = params.expected_state_dict(args, self.swin_cfg)
        shared = [['transformer.%s_decoder.norm.%s' % (k, leaf) for k in ('pt', 'poly', 'rec')]
                  for leaf in ('weight', 'bias')]
        params.attach_parameters(self, spec, shared)
        self._engine = None
        self._engine_key = None
        self.use_graph = True          # decoder steps replay as hipGraphs when run on a non-default stream
        self.overlap_decoders = True   # polygon || recognition decoders on two streams
        # images per encoder pass inside one engine call (see _encode_chunked); None = by engine (enc_chunk_default), OMP355_ENC_CHUNK is
        # the A/B knob of the sweeps
        self.enc_chunk = env_int('OMP355_ENC_CHUNK', 0, 0, 4096) or None
        self._streams = None
        self.phase_events = None       # set to [] to collect (name, torch.cuda.Event) marks per infer()
        self.eval()

    # -- engine lifecycle -------------------------------------------------------------------------
    def _key(self):
        ps = list(self.parameters())
        # OMP355_KV_SPLIT changes the layout of the K / V^T slabs the decoder is built around: part of the key
        return (ps[0].device, self.engine_dtype, sum(p._version for p in ps), id(ps[0]), env_flag('OMP355_KV_SPLIT', True))

    def engine(self):
        """(Encoder, Decoder) packed from the CURRENT parameters; rebuilt after load_state_dict/.to()."""
        key = self._key()
        if self._engine is None or key != self._engine_key:
            dev = key[0]
            if dev.type != 'cuda':
                raise RuntimeError('OmniParser runs on MI355X only: move the model to a cuda (HIP) device; '
                                   'there is no CPU fallback')
            sd = {k: v for k, v in self.state_dict().items()}
            x3 = self.engine_dtype == 'bf16x3'
            tdt = torch.float32 if x3 else self.engine_dtype
            with torch.cuda.device(dev):
                enc = Encoder(sd, self.args, self.swin_cfg, tdt, x3=x3)
                dec = Decoder(sd, self.args, tdt, dev, x3=x3)
            self._engine, self._engine_key = (enc, dec), key
        self._engine[1].use_graph = self.use_graph
        return self._engine

    def _mark(self, name):
        if self.phase_events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.phase_events.append((name, ev))

    def _side_streams(self, dev):
        """Two extra HIP streams so the polygon and recognition decoders overlap (None -> sequential)."""
        if not self.overlap_decoders:
            return None
        if self._streams is None or self._streams[0].device != dev:
            # OMP355_SIDE_PRIO=1 (A/B of the paired schedule, Decoder.decode_poly_and_rec / omp_decoder_run_pair): the side streams at HIGH priority, so that
            # the row-owner chains they carry are placed before the cross-attention workgroups of the caller's stream when a launch drains.  Default 0:
            # high-priority streams take hardware queues of their own, and the four single-stream lanes of a pipelined run created afterwards then
            # share what is left -- bench.py's batch8 leg fell from 228 to 129 img/s with them (profiles/r06g_bench.json vs r06h)
            pr = -1 if env_flag('OMP355_SIDE_PRIO', False) else 0
            self._streams = (torch.cuda.Stream(device=dev, priority=pr), torch.cuda.Stream(device=dev, priority=pr))
        return self._streams

    def set_engine_dtype(self, dtype):
        self.engine_dtype = _DTYPES[dtype]
        self._engine = None

    # -- reference-compatible forward -------------------------------------------------------------
    @torch.no_grad()
    def forward(self, samples, sequence, forced_instances=None):
        if self.training:
            raise NotImplementedError('training is out of scope for the MI355X inference engine')
        if not isinstance(samples, NestedTensor) and hasattr(samples, 'tensors'):
            samples = NestedTensor(samples.tensors, samples.mask)
        img, mask = samples.tensors, samples.mask
        if mask is None:
            mask = torch.zeros(img.shape[0], img.shape[2], img.shape[3], dtype=torch.bool, device=img.device)
        results = self.infer(img, mask, sequence, forced_instances=forced_instances)
        return results[0] if img.shape[0] == 1 else results

    # -- batched inference ------------------------------------------------------------------------
    @torch.no_grad()
    def infer(self, img, mask, sequence, forced_instances=None, has_padding=None, lane=None, packed=None):
        """lane: a pipeline Lane (engine/pipeline.py) -- private decoder state + side streams, so several
        batches can be in flight on different HIP streams; None = the model's own state.
        packed = N (text spotting only): return (ids int32 [B, N, 34 + rec_length], probs [B, N, rec_length], n_inst [B]) device
        tensors -- the all-gather payload of utils/dist.py -- packed by ONE kernel from the decoders' buffers instead of the
        per-image result lists (no per-image host work or device copies)."""
        enc, dec = self.engine()
        side = None
        if lane is not None:
            dec, side = lane.decoder(dec), (lane.side if lane.side is not None else False)   # False: no side streams at all
        a = self.args
        dev = img.device
        B = img.shape[0]
        with torch.cuda.device(dev):
            img = img.float().contiguous()
            if has_padding is None:
                has_padding = bool(mask.any())
            self._mark('start')
            e = self._encode_chunked(enc, img, mask, no_padding=not has_padding)
            self._mark('encode')
            kv = dec.project_memory(e['memory'], e['mem_pos'], B, e['M'], e['key_mask'] if has_padding else None)
            prompt = [int(t) for t in sequence[0].reshape(-1).tolist()]
            poly_sos = int(sequence[1].reshape(-1)[0])
            rec_sos = int(sequence[2].reshape(-1)[0])
            self._mark('kv_project')
            dstream = getattr(lane, 'dec_stream', None) if lane is not None else None
            if dstream is not None:
                # decoder phases on the lane's high-priority stream; the lane stream rejoins below
                outer = torch.cuda.current_stream()
                dstream.wait_stream(outer)
                with torch.cuda.stream(dstream):
                    out = self._decode(dec, kv, prompt, poly_sos, rec_sos, sequence, forced_instances, B, dev, side, packed)
                outer.wait_stream(dstream)
                return out
            return self._decode(dec, kv, prompt, poly_sos, rec_sos, sequence, forced_instances, B, dev, side, packed)

    def enc_chunk_default(self):
        """Images per encoder pass.  What matters at 1024 x 1024 is the last, partial round of the stage-2 chains over the 256 compute
        units (18 of the 24 Swin blocks): a chunk of n images is n x 4096 / 80 workgroups of 80 rows (bf16) = n / 5 rounds, n x 4096 / 48 of
        48 rows (parity engine) = n / 3 rounds.  bf16: 32 -> 40 images (6.4 -> 8.0 rounds) took the encoder from 169.4 to 164.1 ms per 160
        images (profiles/r05n_enc_chunk_32_vs_40.txt), 80 (16.0 rounds, half the launches) to 160.9 (r05q); parity engine: 40 -> 54
        (13.33 -> 18.0 rounds: 160 images are then 54 rounds instead of 56) 375.5 -> 372.0 ms (profiles/r05q_enc_chunk.txt)."""
        return 54 if self.engine_dtype == 'bf16x3' else 80

    def _encode_chunked(self, enc, img, mask, no_padding=False):
        """The encoder gains nothing from more than a few images per launch (its kernels already fill the chip) while
        its activations grow with the batch; the decoders do gain (their steps are latency-bound).  So a large engine
        call is encoded `enc_chunk` images at a time; every chunk's input_proj writes its rows of the call's memory
        tensors directly (the first chunk's are copied once its shape is known)."""
        B, ch = img.shape[0], max(1, int(self.enc_chunk or self.enc_chunk_default()))
        kw = dict(no_padding=True) if no_padding else {}
        if B <= ch:
            return enc.encode(img, mask, **kw)
        first = enc.encode(img[:ch], mask[:ch], **kw)
        M = first['M']
        out = dict(first)
        full = {k: torch.empty((B * M,) + tuple(first[k].shape[1:]), dtype=first[k].dtype, device=first[k].device)
                for k in ('memory', 'mem_pos')}
        for k in full:
            full[k][:ch * M].copy_(first[k])
        pos, km = [first['pos']], [first['key_mask']]
        for i in range(ch, B, ch):
            n = min(ch, B - i)
            p = enc.encode(img[i:i + n], mask[i:i + n], out=(full['memory'][i * M:(i + n) * M], full['mem_pos'][i * M:(i + n) * M]), **kw)
            pos.append(p['pos'])
            km.append(p['key_mask'])
        out.update(full)
        out['pos'] = torch.cat(pos, 0)   # small (diagnostics / parity tests only: the decoders read mem_pos)
        out['key_mask'] = torch.cat(km, 0)
        return out

    def _decode(self, dec, kv, prompt, poly_sos, rec_sos, sequence, forced_instances, B, dev, side, packed=None):
        """point decoder -> polygon || recognition decoders (or the KIE walk) on the current stream"""
        a = self.args
        if packed is not None and a.infer_vie:   # before any work: the KIE branch below returns early
            raise ValueError('packed results are the text-spotting payload; KIE returns entity lists')
        pts = dec.decode_points(kv, prompt, forced_instances=forced_instances)
        self._mark('pt_decode')
        if a.infer_vie:
            sizes = sequence[3]
            return self._kie(dec, kv, pts, poly_sos, rec_sos, sizes, B, side)
        counts = [int(ids.numel()) // 2 for ids, _ in pts]
        R = sum(counts)
        if R == 0:
            if packed is not None:
                return (torch.zeros(B, packed, 34 + a.rec_length, dtype=torch.int32, device=dev),
                        torch.zeros(B, packed, a.rec_length, dtype=torch.float32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
            return [None] * B
        points = torch.cat([ids.reshape(-1, 2) for ids, _ in pts], 0).to(dev, torch.int32)
        (poly, _), (rec, rprob) = dec.decode_poly_and_rec(kv, points, counts, poly_sos, rec_sos, a.rec_length,
                                                          streams=(side or None) if side is not None else self._side_streams(dev))
        if packed is not None:   # one launch from the decoders' own buffers (views into the phase tensors)
            self._mark('poly_rec_decode')
            return ops.pack_spotting(points, poly, rec, rprob, counts, int(packed), a.rec_length)
        poly, rec, rprob = poly.long(), rec.long(), rprob.clone()
        self._mark('poly_rec_decode')
        out, r0 = [], 0
        for b in range(B):
            n = counts[b]
            if n == 0:
                out.append(None)
                continue
            sl = slice(r0, r0 + n)
            out.append(([points[sl].long().reshape(1, -1), poly[sl].reshape(1, -1), rec[sl].unsqueeze(0)],
                        [rprob[sl]]))
            r0 += n
        return out

    # -- KIE assembly (reference transformer.py:143-217) --------------------------------------------------
    def _kie(self, dec, kv, pts, poly_sos, rec_sos, sizes, B, side=None):
        a = self.args
        nb = a.num_bins
        events, words, counts = [], [], []
        for b in range(B):
            ids = pts[b][0].tolist()
            ev, n, i, cnt = [], len(ids), 0, 0
            while i < n:
                if ids[i] < nb:
                    if i + 1 <= n - 1 and ids[i + 1] < nb:
                        ev.append(('word', len(words)))
                        words.append((ids[i], ids[i + 1]))
                        cnt += 1
                        i += 2
                    else:
                        i += 1
                else:
                    ev.append(('class', i))
                    i += 1
            events.append(ev)
            counts.append(cnt)
        poly = rec = None
        if words:
            points = torch.tensor(words, dtype=torch.int32, device=kv['K'].device)
            (poly, _), (rec, _) = dec.decode_poly_and_rec(kv, points, counts, poly_sos, rec_sos, a.rec_length,
                                                          infer_vie=True,
                                                          streams=(side or None) if side is not None else self._side_streams(kv['K'].device))
            poly, rec = poly.cpu(), rec.cpu()
        i2c = index2class(a)
        sizes = _image_sizes(sizes, B)
        out = []
        for b in range(B):
            ids, probs = pts[b][0].tolist(), pts[b][1].tolist()
            if len(ids) == 0:
                out.append(None)
                continue
            ih, iw = sizes[b][0].item(), sizes[b][1].item()
            res, cur_words, cur_rects = [], [], []
            for kind, v in events[b]:
                if kind == 'word':
                    pp = poly[v].reshape(-1, 2)
                    cur_rects.append([iw * pp[:, 0].min().item() / nb, ih * pp[:, 1].min().item() / nb,
                                      iw * pp[:, 0].max().item() / nb, ih * pp[:, 1].max().item() / nb])
                    chars = []
                    for t in rec[v].tolist():
                        if t == a.recog_pad_index or t == a.rec_eos_index:
                            break
                        if t == a.recog_pad_index - 1:
                            continue
                        chars.append(a.chars[t - nb])
                    cur_words.append(''.join(chars))
                else:
                    res.append((' '.join(cur_words), i2c[ids[v]], probs[v], cur_rects))
                    cur_words, cur_rects = [], []
            out.append(res)
        return out
