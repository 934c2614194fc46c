"""Point-conditioned autoregressive decoders on libomp355 (the `decode` half of the hot path).

Host-side mirror of the reference's Transformer (OCR/OmniParser/model/transformer.py):
  * `decode_points`     <-> decode_pt_seq            (:102-141)
  * `decode_instances`  <-> the poly / rec loops     (:252-284) and the per-word loops of
                            decode_vie_pt_poly_rec_seq (:150-183)
  * `assemble_kie`      <-> the walk over (x, y, class) triplets (:143-217)
with the per-step math executed by omp_decoder_run (csrc/decoder.hip): KV-cached, memory K/V
computed once per image and shared by all of its text instances, all images of a batch and all
instances decoded in lock-step.  The host touches the device once per `poll` steps (EOS flags of
the point decoder) and once per phase (results).
"""
import copy
import ctypes
import os
import threading
from collections import OrderedDict

import torch

from .. import _lib, ops
from . import packing
from ..utils.env import env_flag, env_int

KINDS = ('pt', 'poly', 'rec')
KIND_ID = {'pt': _lib.DEC_PT, 'poly': _lib.DEC_POLY, 'rec': _lib.DEC_REC}
LN_EPS = 1e-5

CORD_CLASSES = ['menu.cnt', 'menu.discountprice', 'menu.etc', 'menu.itemsubtotal', 'menu.nm',
                'menu.num', 'menu.price', 'menu.sub.cnt', 'menu.sub.nm', 'menu.sub.price',
                'menu.sub.unitprice', 'menu.unitprice', 'menu.vatyn', 'sub_total.discount_price',
                'sub_total.etc', 'sub_total.othersvc_price', 'sub_total.service_price',
                'sub_total.subtotal_price', 'sub_total.tax_price', 'total.cashprice',
                'total.changeprice', 'total.creditcardprice', 'total.emoneyprice',
                'total.menuqty_cnt', 'total.menutype_cnt', 'total.total_etc', 'total.total_price',
                'void_menu.nm', 'void_menu.price']
SROIE_CLASSES = ['company', 'address', 'date', 'total']


def index2class(args):
    """Class-token -> name table of the reference (transformer.py:49-67, utils/misc.py:6-43)."""
    names = None
    if args.val_dataset:
        if 'cord' in args.val_dataset[0]:
            names = CORD_CLASSES
        elif 'sroie' in args.val_dataset[0]:
            names = SROIE_CLASSES
    return {} if names is None else {args.padding_index + 1 + i: n for i, n in enumerate(names)}


def _round_up(a, b):
    return (a + b - 1) // b * b


class _GraphSlots(object):
    """Process-wide allocator of the library's hipGraph slots (csrc/decoder.hip holds a fixed table): ids are handed out
    from a free list and RECYCLED -- releasing a slot destroys its graph -- so a long evaluation with many distinct
    (decoder, shape) plans neither leaks graph executables nor silently falls back to eager launches."""
    _lock = threading.Lock()
    _free = []
    _next = 0

    @classmethod
    def acquire(cls):
        with cls._lock:
            if cls._free:
                return cls._free.pop()
            if cls._next < _lib.MAX_GRAPH_SLOTS:
                cls._next += 1
                return cls._next - 1
        return -1

    @classmethod
    def release(cls, slot, ctx=None):
        """ctx: the omp_ctx handle the graph was captured in (the slot tables are per context): it is made current for the
        reset and the caller's context restored; None = the calling thread's current context."""
        if slot is None or slot < 0:
            return
        try:
            h = _lib.lib()
            here = h.omp_ctx_current()
            if ctx is not None and ctx != here:
                if ctx not in cls._dead_ctx:     # a destroyed context took its graphs with it
                    h.omp_ctx_make_current(ctx)
                    h.omp_decoder_graph_reset(slot)
                    h.omp_ctx_make_current(here)
            else:
                h.omp_decoder_graph_reset(slot)
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            return
        with cls._lock:
            cls._free.append(slot)

    _dead_ctx = set()   # handles of destroyed contexts (ops.Context.destroy registers them)


# bounds of the per-decoder caches: a real evaluation has variable image sizes and instance counts, so
# K/V slabs (100-200 MB per image size), phase buffers (KV caches, scratch) and captured graphs are LRU-evicted
MAX_KV_ENTRIES = 4
MAX_PHASES = 12
MAX_SLOTS_PER_PHASE = 2


class _Phase(object):
    """Device state + ctypes plan of one decoder kind for R rows (buffers are reused across calls
    so that pointers stay stable and captured graphs remain valid)."""

    def __init__(self, dec, kind, R, Lmax, seq_ld, n_split):
        dev, T, d, ff, V, nH, L = dec.device, dec.dtype, dec.d, dec.ff, dec.V, dec.nH, dec.L
        self.kind, self.R, self.Lmax, self.seq_ld, self.n_split = kind, R, Lmax, seq_ld, n_split
        z = lambda *s, dtype=torch.int32: torch.zeros(s, dtype=dtype, device=dev)  # noqa: E731
        e = lambda *s, dtype=T: torch.empty(s, dtype=dtype, device=dev)             # noqa: E731
        self.seq, self.probs = z(R, seq_ld), z(R, seq_ld, dtype=torch.float32)
        self.finished, self.lengths, self.d_pos = z(R), z(R), z(2)   # d_pos: {position, ticket word of the sampling kernel}
        self.tiles = z(R + 64, 3)
        self.x, self.x2 = e(R, d, dtype=torch.float32), e(R, d, dtype=torch.float32)
        self.y, self.att, self.q, self.hh0, self.hh1 = e(R, d), e(R, d), e(R, d), e(R, d), e(R, d)
        self.qkv, self.ffh = e(R, 3 * d), e(R, ff)
        self.logits = e(R, V, dtype=torch.float32)
        self.partial = e(R, nH, max(1, n_split), 68, dtype=torch.float32)   # workgroup-level partials
        self.kc = [e(R, Lmax, d) for _ in range(L)]
        self.vc = [e(R, Lmax, d) for _ in range(L)]
        self.plan = _lib.DecoderPlan()
        self.n_tiles = 0
        self.slots = OrderedDict()   # plan bytes -> graph slot (a phase re-bound to other K/V slabs is another graph)

    def release_graphs(self):
        for slot, ctx in self.slots.values():
            _GraphSlots.release(slot, ctx)
        self.slots.clear()


class Decoder(object):
    def __init__(self, sd, args, dtype, device, x3=False):
        self.args, self.dtype, self.device = args, dtype, device
        self.x3 = bool(x3)   # bf16x3 engine: the K / V^T projection of the memory runs as split-bf16 products (decoder steps stay fp32)
        # ... and (round 4) writes SPLIT-PLANE slabs: 32-key blocks of [hi plane | lo plane] bf16 -- the bytes of the fp32 slabs, streamed
        # by cross-attention kernels that run three bf16 matrix-core products per block instead of fp32 ones (csrc/decoder.hip, bf16s_t)
        self.kv_split = self.x3 and env_flag('OMP355_KV_SPLIT', True)
        self.d, self.nH, self.L = args.tfm_hidden_dim, args.tfm_nheads, args.tfm_dec_layers
        self.ff, self.V = args.tfm_dim_feedforward, args.num_classes
        if self.d != self.nH * 64:
            raise ValueError('libomp355 decoder kernels are built for head_dim 64')
        if self.L > _lib.MAX_DEC_LAYERS:
            raise ValueError('too many decoder layers')
        self.use_graph = False
        self.n_split_override = env_int('OMP355_CROSS_SPLIT', 0, 0, 16, allowed=(0, 1, 2, 4, 8, 16))
        self._phases = OrderedDict()
        self._kv = OrderedDict()
        d = self.d
        f32 = lambda k: sd[k].detach().float().contiguous()      # noqa: E731
        mat = lambda t: t.detach().to(dtype).contiguous()        # noqa: E731
        tr = 'transformer.'
        self.word = f32(tr + 'embedding.word_embeddings.weight')
        self.emb_g, self.emb_b = f32(tr + 'embedding.LayerNorm.weight'), f32(tr + 'embedding.LayerNorm.bias')
        self.pos_tab, self.layers, self.fn, self.head = {}, {}, {}, {}
        wk, bk, wv, bv = [], [], [], []
        for kind in KINDS:
            pos = f32('%sembedding.%s_position_embeddings.weight' % (tr, kind))
            self.pos_tab[kind] = pos
            layers = []
            for l in range(self.L):
                p = '%s%s_decoder.layers.%d.' % (tr, kind, l)
                w = {}
                in_w, in_b = f32(p + 'self_attn.in_proj_weight'), f32(p + 'self_attn.in_proj_bias')
                w['sa_in_w'] = mat(in_w)
                # (y + qpos) Wq^T + bq == y Wq^T + (qpos Wq^T + bq): fold the position term into a
                # per-position bias table (fp32 GEMM on the fp32 master weights); v takes no qpos.
                qk = ops.gemm(pos, in_w[:2 * d].contiguous(), in_b[:2 * d].contiguous())
                w['sa_bias_tab'] = torch.cat([qk, in_b[2 * d:].expand(pos.shape[0], d)], dim=1).contiguous()
                w['sa_out_w'], w['sa_out_b'] = mat(f32(p + 'self_attn.out_proj.weight')), f32(p + 'self_attn.out_proj.bias')
                cw, cb = f32(p + 'multihead_attn.in_proj_weight'), f32(p + 'multihead_attn.in_proj_bias')
                w['ca_q_w'] = mat(cw[:d])
                w['ca_qbias_tab'] = ops.gemm(pos, cw[:d].contiguous(), cb[:d].contiguous())
                wk.append(cw[d:2 * d]); bk.append(cb[d:2 * d]); wv.append(cw[2 * d:]); bv.append(cb[2 * d:])
                w['ca_out_w'], w['ca_out_b'] = mat(f32(p + 'multihead_attn.out_proj.weight')), f32(p + 'multihead_attn.out_proj.bias')
                w['ff1_w'], w['ff1_b'] = mat(f32(p + 'linear1.weight')), f32(p + 'linear1.bias')
                w['ff2_w'], w['ff2_b'] = mat(f32(p + 'linear2.weight')), f32(p + 'linear2.bias')
                for n in ('1', '2', '3'):
                    w['n%s_g' % n], w['n%s_b' % n] = f32(p + 'norm%s.weight' % n), f32(p + 'norm%s.bias' % n)
                layers.append(w)
            self.layers[kind] = layers
            self.fn[kind] = (f32('%s%s_decoder.norm.weight' % (tr, kind)), f32('%s%s_decoder.norm.bias' % (tr, kind)))
            hp = '%s%s_pred_layer.layers.' % (tr, kind)
            self.head[kind] = [(mat(f32(hp + '%d.weight' % i)), f32(hp + '%d.bias' % i)) for i in range(3)]
        # one stacked projection for the memory K and V of all (decoder, layer) pairs
        if self.x3:
            # K: tokens are the A operand (split pairs, a_wrap), the weight its [hi | hi | lo] image; V^T: operands swapped,
            # the weight is the [hi | lo] A side and the tokens arrive as [hi | hi | lo] (ops.split_bf16(triple=True))
            self.Wk_all, self.Wv_all = ops.split_weight3(torch.cat(wk, 0)), ops.split_weight2(torch.cat(wv, 0))
            self.bk_all, self.bv_all = torch.cat(bk, 0).contiguous(), torch.cat(bv, 0).contiguous()
        else:
            self.Wk_all, self.bk_all = mat(torch.cat(wk, 0)), torch.cat(bk, 0).contiguous()
            self.Wv_all, self.bv_all = mat(torch.cat(wv, 0)), torch.cat(bv, 0).contiguous()
        self.NL = len(KINDS) * self.L
        self._rows_cache = {}   # kind -> packed weight streams of the row-owner chains (bf16 engine, many-row phases)
        self._kv_streams = None  # packed Wk_all / Wv_all of the row-owner memory projection (bf16 engine, built on first use)
        self.kv_rows = env_flag('OMP355_KV_ROWS', True)   # False: the two tiled GEMMs with slab epilogues (A/B)
        self.rows_min = self.ROWS_MIN_ROWS
        # polygon || recognition phases on the chains: their steps interleaved with serialised cross-attention launches (omp_decoder_run_pair);
        # OMP355_PAIR=0: two free-running streams of step graphs (round 5; A/B)
        self.pair_stagger = env_flag('OMP355_PAIR', False)
        # XCD placement of the many-row chains per decoder kind (omp_decoder_plan.rows_xcd_mask): each decoder's weight set then fills four private L2s
        # instead of eight -- HBM traffic of the chains 1.30x -> 1.14x algorithmic (profiles/r06d_pmc_dec_rows_xcd_split.json) -- but the OTHER
        # decoder's cross-attention is then confined to four XCDs' share of the fabric: polygon + recognition 98 -> 109 ms per 160 images
        # (profiles/r06d_ab_xcd_split_*).  Off by default; OMP355_XCD_SPLIT=1 is the A/B knob
        self.xcd_masks = {'poly': 0x0F, 'rec': 0xF0} if env_flag('OMP355_XCD_SPLIT', False) else {}
        self._x3_cache = {}   # kind -> (layers, head) with every matrix as its [w_hi | w_hi | w_lo] image (bf16x3 engine, R > 64)

    GRAPH_RUN = 8       # positions per omp_decoder_run call of the two-stream polygon / recognition schedule (csrc/common.h OMP_GRAPH_RUN)
    X3_MIN_ROWS = env_int('OMP355_X3_MIN', 65, 65, 1 << 30)    # phases with more rows run their products as split-bf16 products (csrc/decoder.hip: step_launch_x3)
    # bf16 engine: phases with at least this many rows run their Linear layers as row-owner chains (csrc/dec_rows.hip: 80 rows per
    # workgroup, weights streamed from L2): from ~50 workgroups on they beat the launch-per-Linear path (profiles/r05*_kbench_dec_rows*)
    ROWS_MIN_ROWS = env_int('OMP355_ROWS_MIN', 4096, 1, 1 << 30)
    # from this many rows (the fused few-row kernels end at 63) the launch-per-Linear path runs its three launches between self- and
    # cross-attention as the mid chain; OMP355_MID_MIN=1073741824 switches it off (A/B)
    MID_MIN_ROWS = env_int('OMP355_MID_MIN', 64, 16, 1 << 30)

    def _rows_streams(self, kind):
        """Packed weight streams of decoder `kind` for the row-owner chains, built on first use (model/packing.py::pack_rows_*):
        (embed, [(mid, ffn) per layer]); the last layer's ffn stream ends in the prediction head."""
        if kind not in self._rows_cache:
            Ls, hd = self.layers[kind], self.head[kind]
            if tuple(hd[0][0].shape) != (self.d, self.d) or tuple(hd[1][0].shape) != (self.d, self.d) or hd[2][0].shape[1] != self.d:
                raise ValueError('row-owner chains: the prediction head must be %d -> %d -> %d -> vocab' % (self.d, self.d, self.d))
            embed = packing.pack_rows_embed_qkv(Ls[0]['sa_in_w'])   # every entry: (stream buffer, bytes per wave) as the packer returns them
            per = []
            for l, w in enumerate(Ls):
                mid = packing.pack_rows_mid(w['sa_out_w'], w['ca_q_w'])
                if l + 1 < len(Ls):
                    ffn = packing.pack_rows_ffn_qkv(w['ca_out_w'], w['ff1_w'], w['ff2_w'], Ls[l + 1]['sa_in_w'])
                else:
                    ffn = packing.pack_rows_ffn_head(w['ca_out_w'], w['ff1_w'], w['ff2_w'], hd[0][0], hd[1][0], hd[2][0])
                per.append((mid, ffn))
            self._rows_cache[kind] = (embed, per)
        return self._rows_cache[kind]

    def _rows_mid_streams(self, kind):
        """Only the mid streams (sa_out_w, ca_q_w: 1 MB per layer) of decoder `kind`: the phases between the fused few-row kernels and the full
        chains run the three launches between self- and cross-attention as the mid chain (csrc/decoder.hip step_launch)."""
        key = ('mid', kind)
        if key not in self._rows_cache:
            if kind in self._rows_cache:
                self._rows_cache[key] = [m for m, _ in self._rows_cache[kind][1]]
            else:
                self._rows_cache[key] = [packing.pack_rows_mid(w['sa_out_w'], w['ca_q_w']) for w in self.layers[kind]]
        return self._rows_cache[key]

    def _x3_weights(self, kind):
        """[out, 3 in] bf16 images of decoder `kind`'s matrices, built on first use (the fp32 masters stay bound for the
        few-row phases, whose weight-streaming kernels are fp32)."""
        if kind not in self._x3_cache:
            mats = ('sa_in_w', 'sa_out_w', 'ca_q_w', 'ca_out_w', 'ff1_w', 'ff2_w')
            layers = [{n: ops.split_weight3(w[n]) for n in mats} for w in self.layers[kind]]
            head = [ops.split_weight3(w) for w, _ in self.head[kind]]
            self._x3_cache[kind] = (layers, head)
        return self._x3_cache[kind]

    # -- memory K/V: once per batch ---------------------------------------------------------------
    def project_memory(self, memory, mem_pos, B, M, key_mask):
        """Cross-attention memory of all (decoder, layer) pairs, written by the GEMM epilogues straight into
        the head-blocked slabs the step kernel streams (DESIGN.md "cross-attention memory layout"):
          K   = (memory+pos) Wk^T + bk  ->  [NL][B][nH][Mpad][64]
          V^T = (memory Wv^T + bv)^T    ->  [NL][B][nH][Mpad/KB][64][KB]
        The padded tail (keys >= M) is zero from allocation and never written."""
        KB = 16 if (self.dtype == torch.float32 and not self.kv_split) else 32
        Mpad = _round_up(M, KB)
        key = (B, M)
        if key not in self._kv:
            while len(self._kv) >= MAX_KV_ENTRIES:
                self._kv.popitem(last=False)   # plans bound to these slabs keep them alive until their phase goes too
            if self.kv_split:   # [block][plane][32 keys x 64 dims] and [block][plane][64 dims x 32 key slots]
                slabs = (torch.zeros(self.NL, B, self.nH, Mpad // 32, 2, 32, 64, dtype=torch.bfloat16, device=self.device),
                         torch.zeros(self.NL, B, self.nH, Mpad // 32, 2, 64, 32, dtype=torch.bfloat16, device=self.device))
            else:
                slabs = (torch.zeros(self.NL, B, self.nH, Mpad, 64, dtype=self.dtype, device=self.device),
                         torch.zeros(self.NL, B, self.nH, Mpad // KB, 64, KB, dtype=self.dtype, device=self.device))
            self._kv[key] = slabs + (torch.zeros(B, M, dtype=torch.uint8, device=self.device),)
        self._kv.move_to_end(key)
        K_all, Vt_all, mask_buf = self._kv[key]
        if key_mask is not None:
            # a STABLE buffer: the plan (and the graph captured from it) holds this pointer, not the caller's tensor
            mask_buf.copy_(key_mask.reshape(B, M))
            key_mask = mask_buf
        geom = (B, M, Mpad, self.nH, KB)
        if self.x3:
            d = self.d
            od = ops.SPLIT if self.kv_split else None
            ops.gemm(ops.split_bf16(mem_pos), self.Wk_all, self.bk_all, out=K_all, out_dtype=od, store_mode=_lib.STORE_KBLK, kv=geom, a_wrap=2 * d,
                     M=B * M, N=self.Wk_all.shape[0], K=3 * d)
            ops.gemm(self.Wv_all, ops.split_bf16(memory, triple=True), self.bv_all, out=Vt_all, out_dtype=od, store_mode=_lib.STORE_VBLK, kv=geom,
                     bias_along_m=True, a_wrap=2 * d, M=self.Wv_all.shape[0], N=B * M, K=3 * d)
            return dict(K=K_all, Vt=Vt_all, B=B, M=M, Mpad=Mpad, KB=KB, key_mask=key_mask)
        if self.kv_rows and self.dtype == torch.bfloat16 and self.d == 512 and self.nH == 8 and M % 64 == 0:
            # one row-owner launch per tensor (csrc/kv_rows.hip): 64 memory rows per workgroup, the weights of all slabs streamed; bit-identical slabs
            if self._kv_streams is None:
                self._kv_streams = (packing.pack_kv_rows_k(self.Wk_all), packing.pack_kv_rows_v(self.Wv_all))
            (sk, nk), (sv, nv) = self._kv_streams
            ops.kv_project_rows(mem_pos, sk, nk, self.bk_all, K_all, B, M, Mpad, self.NL, False)
            ops.kv_project_rows(memory, sv, nv, self.bv_all, Vt_all, B, M, Mpad, self.NL, True)
            return dict(K=K_all, Vt=Vt_all, B=B, M=M, Mpad=Mpad, KB=KB, key_mask=key_mask)
        ops.gemm(mem_pos, self.Wk_all, self.bk_all, out=K_all, store_mode=_lib.STORE_KBLK, kv=geom)
        # swapped operands: rows = value features, columns = memory tokens, so a lane owns 4 consecutive keys
        ops.gemm(self.Wv_all, memory, self.bv_all, out=Vt_all, store_mode=_lib.STORE_VBLK, kv=geom, bias_along_m=True,
                 M=self.Wv_all.shape[0], N=B * M, K=self.d)
        return dict(K=K_all, Vt=Vt_all, B=B, M=M, Mpad=Mpad, KB=KB, key_mask=key_mask)

    # -- plans --------------------------------------------------------------------------------------
    def _phase(self, kind, R, Lmax, seq_ld, n_split):
        key = (kind, R, Lmax, seq_ld, n_split)
        if key not in self._phases:
            while len(self._phases) >= MAX_PHASES:
                _, old = self._phases.popitem(last=False)
                old.release_graphs()
            self._phases[key] = _Phase(self, kind, R, Lmax, seq_ld, n_split)
        self._phases.move_to_end(key)
        return self._phases[key]

    @staticmethod
    def make_tiles(counts):
        """rows sorted by image -> (groups, q_tiles): groups of <= 16*q_tiles consecutive rows of ONE image,
        (row0, nrows, image); one cross-attention workgroup streams an (image, head) key range once per group."""
        mx = max(counts) if counts else 1
        qt = 1 if mx <= 16 else (2 if mx <= 32 else 4)
        groups, r0 = [], 0
        for img, n in enumerate(counts):
            for o in range(0, n, 16 * qt):
                groups.append((r0 + o, min(16 * qt, n - o), img))
            r0 += n
        return groups, qt

    def _bind(self, ph, kv, tiles, n_prompt, suppress_eos, infer_vie):
        tiles, qt = tiles
        P, a, d = ph.plan, self.args, self.d
        esz = 4 if self.dtype == torch.float32 else 2   # bytes per (key, dim): a split-plane pair is 2 x 2
        P.dtype, P.n_layers, P.d_model, P.n_heads, P.d_ff, P.vocab = ops.dt(self.dtype), self.L, d, self.nH, self.ff, self.V
        P.pre_norm = 1 if a.tfm_pre_norm else 0
        # row-owner chains: the bf16 engine's (csrc/dec_rows.hip) or, on a gemm_x3 plan, the parity engine's (csrc/dec_rows_x3.hip).  A phase that
        # takes the chains is a gemm_x3 plan whatever its row count (the tiled x3 GEMMs want more than 64 rows, the chains any number: the parity
        # tests lower rows_min to put the fixtures' 1 .. 64-row phases on the benchmark's kernels)
        rows_ok = bool(a.tfm_pre_norm and ph.R >= self.rows_min and d == 512 and self.ff == 2048 and self.nH == 8 and self.V % 4 == 0)
        use_x3 = bool(self.x3 and a.tfm_pre_norm and (ph.R >= self.X3_MIN_ROWS or rows_ok) and d % 64 == 0 and self.ff % 64 == 0)
        P.gemm_x3 = 1 if use_x3 else 0
        P.kv_split = 1 if self.kv_split else 0
        use_rows = bool((self.dtype == torch.bfloat16 or use_x3) and rows_ok)
        P.rows_fused = 1 if use_rows else 0
        if use_rows:
            r_embed, r_layers = self._rows_streams(ph.kind)
            P.rows_embed, P.rows_embed_stride = r_embed[0].data_ptr(), r_embed[1]
        else:
            P.rows_embed, P.rows_embed_stride = None, 0
        # (A/B, off by default) each decoder's chains on its own four XCDs (include/omp355.h rows_xcd_mask; launches of more than 128 workgroups ignore it)
        P.rows_xcd_mask = self.xcd_masks.get(ph.kind, 0) if use_rows else 0
        # in between (more rows than the fused few-row kernels take, fewer than the chains want): the mid chain alone
        use_mid = bool(not use_rows and self.dtype == torch.bfloat16 and not self.kv_split and a.tfm_pre_norm and self.MID_MIN_ROWS <= ph.R
                       and d == 512 and self.nH == 8)
        r_mid = self._rows_mid_streams(ph.kind) if use_mid else None
        x3_layers, x3_head = self._x3_weights(ph.kind) if use_x3 else (None, None)
        P.R, P.Lmax, P.M, P.Mpad, P.n_tiles, P.q_tiles, P.n_split, P.n_prompt = (ph.R, ph.Lmax, kv['M'], kv['Mpad'], len(tiles), qt,
                                                                                 ph.n_split, n_prompt)
        P.eps = LN_EPS
        t = torch.tensor(tiles, dtype=torch.int32).reshape(-1, 3)
        ph.tiles[:t.shape[0]].copy_(t.to(self.device, non_blocking=False))
        ph.n_tiles = t.shape[0]
        kidx = KINDS.index(ph.kind)
        img_stride = self.nH * kv['Mpad'] * 64      # (key, dim) pairs per image
        slab = kv['B'] * img_stride   # one (decoder, layer) slab, K and V^T alike
        for l, w in enumerate(self.layers[ph.kind]):
            Lc = P.layers[l]
            for name in ('sa_in_w', 'sa_bias_tab', 'sa_out_w', 'sa_out_b', 'ca_q_w', 'ca_qbias_tab', 'ca_out_w',
                         'ca_out_b', 'ff1_w', 'ff1_b', 'ff2_w', 'ff2_b', 'n1_g', 'n1_b', 'n2_g', 'n2_b', 'n3_g', 'n3_b'):
                setattr(Lc, name, (x3_layers[l][name] if use_x3 and name in x3_layers[l] else w[name]).data_ptr())
            Lc.kcache, Lc.vcache = ph.kc[l].data_ptr(), ph.vc[l].data_ptr()
            if use_rows:
                (m_, ms_), (f_, fs_) = r_layers[l]
                Lc.rows_mid, Lc.rows_mid_stride, Lc.rows_ffn, Lc.rows_ffn_stride = m_.data_ptr(), ms_, f_.data_ptr(), fs_
            elif use_mid:
                Lc.rows_mid, Lc.rows_mid_stride, Lc.rows_ffn, Lc.rows_ffn_stride = r_mid[l][0].data_ptr(), r_mid[l][1], None, 0
            else:
                Lc.rows_mid, Lc.rows_mid_stride, Lc.rows_ffn, Lc.rows_ffn_stride = None, 0, None, 0
            off = (kidx * self.L + l) * slab * esz
            Lc.crossK = kv['K'].data_ptr() + off
            Lc.crossVt = kv['Vt'].data_ptr() + off
        P.word_emb, P.pos_tab = self.word.data_ptr(), self.pos_tab[ph.kind].data_ptr()
        P.emb_g, P.emb_b = self.emb_g.data_ptr(), self.emb_b.data_ptr()
        P.fn_g, P.fn_b = self.fn[ph.kind][0].data_ptr(), self.fn[ph.kind][1].data_ptr()
        (P.h0_w, P.h0_b), (P.h1_w, P.h1_b), (P.h2_w, P.h2_b) = [((x3_head[i] if use_x3 else w).data_ptr(), b.data_ptr())
                                                                 for i, (w, b) in enumerate(self.head[ph.kind])]
        P.kv_img_stride = img_stride * (2 if self.kv_split else 1)   # in slab elements: two bf16 planes per (key, dim)
        P.key_mask = kv['key_mask'].data_ptr() if kv['key_mask'] is not None else None
        P.tiles = ph.tiles.data_ptr()
        P.seq, P.seq_ld, P.d_pos, P.probs = ph.seq.data_ptr(), ph.seq_ld, ph.d_pos.data_ptr(), ph.probs.data_ptr()
        P.finished, P.lengths = ph.finished.data_ptr(), ph.lengths.data_ptr()
        for n in ('x', 'x2', 'y', 'qkv', 'att', 'q', 'ffh', 'hh0', 'hh1', 'partial', 'logits'):
            setattr(P, n, getattr(ph, n).data_ptr())
        s = P.sample
        s.kind, s.num_bins, s.pt_eos, s.poly_eos, s.rec_eos = KIND_ID[ph.kind], a.num_bins, a.pt_eos_index, a.poly_eos_index, a.rec_eos_index
        s.vocab, s.vie_categories, s.infer_vie = self.V, a.vie_categories, 1 if infer_vie else 0
        s.suppress_eos, s.step0 = 1 if suppress_eos else 0, n_prompt
        ph._keepalive = (kv['K'], kv['Vt'], kv['key_mask'])
        return P

    def fork(self):
        """A decoder that shares the packed weights but owns its phase buffers, K/V slabs and graph slots:
        one per pipeline lane (engine/pipeline.py), so lanes can be in different phases at the same time."""
        other = copy.copy(self)
        other._phases, other._kv = OrderedDict(), OrderedDict()
        return other

    def release(self):
        for ph in self._phases.values():
            ph.release_graphs()
        self._phases.clear()
        self._kv.clear()

    def __del__(self):
        try:
            self.release()
        except Exception:   # noqa: BLE001
            pass

    def _slot(self, ph):
        # stream capture is illegal on the legacy null stream: graphs only when the caller runs us on a
        # real stream (bench / predict do), eager launches otherwise
        if not self.use_graph or torch.cuda.current_stream().cuda_stream == 0:
            return -1
        here = _lib.lib().omp_ctx_current()
        key = bytes(ph.plan) + here.to_bytes(8, 'little')   # a graph lives in the omp_ctx it was captured in
        if key in ph.slots:
            ph.slots.move_to_end(key)
            return ph.slots[key][0]
        while len(ph.slots) >= MAX_SLOTS_PER_PHASE:
            _, (old, octx) = ph.slots.popitem(last=False)
            _GraphSlots.release(old, octx)
        slot = _GraphSlots.acquire()   # -1 (eager launches) only when every slot of the library's table is live
        if slot >= 0:
            ph.slots[key] = (slot, here)
        return slot

    def _run(self, ph, first_pos, n_steps):
        if n_steps <= 0:
            return
        rc = _lib.lib().omp_decoder_run(ctypes.byref(ph.plan), first_pos, n_steps, self._slot(ph), ops.stream())
        _lib.check(rc, 'omp_decoder_run')

    def _n_split(self, tiles, M):
        """Workgroup-level key splits S of the cross-attention kernel (each workgroup's 4 waves split again).
        Per-wave slice ~4 key blocks (one memory round trip with 4 blocks in flight) but never fewer than
        ~512 workgroups' worth of parallelism when the memory is large; power of two <= 16."""
        groups, qt = tiles
        kb = 16 if (self.dtype == torch.float32 and not self.kv_split) else 32
        if self.n_split_override:
            return self.n_split_override
        if qt == 4:   # bf16 and fp32 slabs alike (csrc/decoder.hip: dec_cross_attn_q4_kernel)
            # LDS-ring kernel (waves own query tiles, the workgroup streams one key range): enough workgroups
            # to put two on every CU, but at least 4 key blocks each
            S = 1
            while S < 16 and len(groups) * self.nH * S < 512 and M >= 8 * kb * S:
                S *= 2
            return S
        per_wave = 4 * kb * (2 if qt > 1 else 1)
        S = 1
        while S < 16 and 4 * S * per_wave < M:
            S *= 2
        # ~4 workgroups per CU are enough to saturate HBM (profiles/r02j_kbench_cross128.txt: 128 images x 8 heads at
        # S = 1 stream at 6.1-6.3 TB/s); beyond that a split only adds the partial buffer and the merge launch
        while S > 1 and len(groups) * self.nH * S > 1024:
            S //= 2
        return S

    # -- greedy drivers ---------------------------------------------------------------------------
    def decode_points(self, kv, prompt, max_new=None, forced_instances=None, poll=16):
        """Point decoder for B images in lock-step.  Returns per image (ids list, probs list) after
        prompt strip and the reference's tail trim (transformer.py:131-141)."""
        a, B = self.args, kv['B']
        n_prompt = len(prompt)
        limit = 1024 - n_prompt + 1  # beyond this the reference indexes past its position table
        S = min(a.pt_seq_length if max_new is None else max_new, limit)
        suppress = forced_instances is not None
        if suppress:
            S = min((3 if a.infer_vie else 2) * forced_instances, limit)
        tiles = self.make_tiles([1] * B)
        ph = self._phase('pt', B, n_prompt - 1 + S, n_prompt + S + 1, self._n_split(tiles, kv['M']))
        self._bind(ph, kv, tiles, n_prompt, suppress, a.infer_vie)
        ph.seq.zero_(); ph.probs.zero_(); ph.finished.zero_(); ph.lengths.zero_(); ph.d_pos.zero_()
        ph.seq[:, :n_prompt] = torch.tensor(prompt, dtype=torch.int32, device=self.device)
        total = n_prompt - 1 + S
        done, fin = 0, [0] * B
        if suppress:
            self._run(ph, 0, total)
            done = total
        else:
            while done < total:
                n = min(total - done, (n_prompt - 1 if done == 0 else 0) + poll)
                self._run(ph, done, n)
                done += n
                fin = ph.finished.tolist()   # the only host sync of the point phase
                if all(fin):
                    break
        seq, probs, lens = ph.seq.cpu(), ph.probs.cpu(), ph.lengths.tolist()
        sampled = done - (n_prompt - 1)      # sampling steps executed
        out = []
        for b in range(B):
            # finished rows: tokens before the EOS (reference breaks before appending it, :126-127)
            end = lens[b] if fin[b] else n_prompt + sampled
            ids, pr = seq[b, n_prompt:end], probs[b, n_prompt:end]
            if ids.numel() % 2 != 0:          # reference :138-139
                ids = ids[:-1]
            out.append((ids, pr))
        return out

    def begin_instances(self, kind, kv, points, counts, sos, n_new, infer_vie=False):
        """Set up poly / rec decoding of R = sum(counts) instances (rows sorted by image).  points: int32
        [R,2] on the device.  Returns the phase; drive it with `_run(ph, pos, n)` for 2 + n_new positions."""
        R = int(points.shape[0])
        tiles = self.make_tiles(counts)
        ph = self._phase(kind, R, 2 + n_new, 3 + n_new + 1, self._n_split(tiles, kv['M']))
        self._bind(ph, kv, tiles, 3, False, infer_vie)
        ph.d_pos.zero_()
        ph.seq[:, 0:2] = points
        ph.seq[:, 2] = sos
        ph.n_new = n_new
        return ph

    @staticmethod
    def instances_result(ph):
        return ph.seq[:, 3:3 + ph.n_new], ph.probs[:, 3:3 + ph.n_new]

    def decode_instances(self, kind, kv, points, counts, sos, n_new, infer_vie=False):
        """-> (ids [R,n_new] int32, probs [R,n_new] fp32) views into the phase buffers (device)."""
        ph = self.begin_instances(kind, kv, points, counts, sos, n_new, infer_vie)
        self._run(ph, 0, 2 + n_new)
        return self.instances_result(ph)

    def decode_poly_and_rec(self, kv, points, counts, poly_sos, rec_sos, rec_length, infer_vie=False, streams=None):
        """The polygon and recognition decoders only depend on the points: run them CONCURRENTLY on two
        streams, enqueueing step by step so that both queues stay fed (small-kernel phases do not fill
        256 CUs on their own)."""
        if streams is None or torch.cuda.current_stream().cuda_stream == 0:
            poly = self.decode_instances('poly', kv, points, counts, poly_sos, 32, infer_vie)
            rec = self.decode_instances('rec', kv, points, counts, rec_sos, rec_length, infer_vie)
            return poly, rec
        cur = torch.cuda.current_stream()
        sp, sr = streams
        sp.wait_stream(cur)
        sr.wait_stream(cur)
        with torch.cuda.stream(sp):
            php = self.begin_instances('poly', kv, points, counts, poly_sos, 32, infer_vie)
        with torch.cuda.stream(sr):
            phr = self.begin_instances('rec', kv, points, counts, rec_sos, rec_length, infer_vie)
        np_, nr = 2 + 32, 2 + rec_length
        if self.pair_stagger and php.plan.rows_fused and phr.plan.rows_fused:
            # both phases on the row-owner chains: ONE interleaved schedule whose cross-attention launches are serialised by events, so that one
            # decoder's chains (half the compute units) run beside the other's HBM-bound cross-attention (csrc/decoder.hip: omp_decoder_run_pair)
            rc = _lib.lib().omp_decoder_run_pair(ctypes.byref(php.plan), ctypes.byref(phr.plan), 0, np_, nr, sp.cuda_stream, sr.cuda_stream, cur.cuda_stream)
            _lib.check(rc, 'omp_decoder_run_pair')
            cur.wait_stream(sp)
            cur.wait_stream(sr)
            return self.instances_result(php), self.instances_result(phr)
        # enqueue in runs of GRAPH_RUN positions, alternating between the two streams (both queues stay fed; omp_decoder_run replays a run of sampling
        # steps as ONE graph: consecutive graph launches are ~8.5 us apart on the GPU)
        # (few-row phases keep alternating step by step: runs of 8 bought them nothing -- the 8-image phase reads 19.7-21.1 ms with or without,
        # box to box -- and a launch-bound stream that waits eight steps for its turn can only lose; 160-image call 100.7 -> 99.3 ms, KIE's
        # 2048-row phases +1 %: profiles/r06l_*)
        run = self.GRAPH_RUN if php.R >= 2048 else 1
        for pos in range(0, max(np_, nr), run):
            if pos < np_:
                with torch.cuda.stream(sp):
                    self._run(php, pos, min(run, np_ - pos))
            if pos < nr:
                with torch.cuda.stream(sr):
                    self._run(phr, pos, min(run, nr - pos))
        cur.wait_stream(sp)
        cur.wait_stream(sr)
        return self.instances_result(php), self.instances_result(phr)

    def teacher_forced_logits(self, kind, kv, seqs, counts, n_prompt, infer_vie=False):
        """Parity helper: logits [R, L, V] of decoder `kind` fed the given token sequences."""
        seqs = seqs.to(self.device, torch.int32)
        R, Ls = seqs.shape
        tiles = self.make_tiles(counts)
        ph = self._phase(kind, R, Ls, Ls + 1, self._n_split(tiles, kv['M']))
        self._bind(ph, kv, tiles, n_prompt, False, infer_vie)
        ph.d_pos.zero_()
        ph.seq[:, :Ls] = seqs
        outs = []
        for p in range(Ls):
            rc = _lib.lib().omp_decoder_step_logits(ctypes.byref(ph.plan), p, ops.stream())
            _lib.check(rc, 'omp_decoder_step_logits')
            outs.append(ph.logits.clone())
        return torch.stack(outs, dim=1)
