"""Swin-B backbone + FPN + input projection on libomp355 (the `encode` half of the hot path).

Host-side driver only: every tensor op is a libomp355 kernel (ops.py); the launches are large
(token-level GEMMs, fused window attention, LayerNorm) so Python-side sequencing stays far ahead
of the GPU.  Mirrors, stage by stage, SwinTransformer.forward (reference
model/backbone/swin_transformer.py:597-625), Joiner/PositionEmbeddingSine
(backbone/joiner.py:10-18, position_embedding.py:24-44), FPN.forward (model/fpn.py:21-45) and the
input_proj call of OmniParser.forward (model/omniparser.py:19-31).

Layout: activations stay token-major [B*H*W, C] in the engine dtype for the whole backbone -- the
reference's NCHW permutes, window partition/reverse copies, rolls and pads never materialise.

Engine precisions (model/omniparser.py `engine_dtype`):
  * 'bf16'    bf16 activations and weights, fp32 accumulation (the throughput configuration);
  * 'fp32'    everything in fp32 on the fp32 matrix-core path (the reference's own precision);
  * 'bf16x3'  the PARITY engine: fp32 storage and fp32 non-GEMM kernels exactly as 'fp32', but every large product runs on
              the bf16 matrix cores as three bf16 products of split operands (x = hi + lo: hi.w_hi + lo.w_hi + hi.w_lo;
              include/omp355.h, omp_gemm_args.a_wrap).  GEMM inputs travel as split pair rows [hi | lo] (same bytes as
              fp32) written by their producers (LayerNorm, window attention, the fc1 epilogue); weights are split once.
"""
import torch

from .. import ops
from ..utils.env import env_int
from .packing import pack_attn_block, pack_mlp, pack_rows_embed_qkv, pack_rows_ffn, pack_rows_ffn_qkv

LN_EPS = 1e-5
# stages whose MLP runs as ONE fused launch in the bf16 engine (csrc/mlp.hip); at C >= 512 the row-stationary kernel is
# bound by its weight stream and the two GEMMs win (profiles/r02c_kbench_mlp.txt)
FUSED_MLP_WIDTHS = (128, 256)
# C = 512 (stage 2 of Swin-B: 18 of the 24 blocks, 60 % of the encoder): everything of a block except the window attention core as ONE row-owner
# chain (csrc/dec_rows.hip, omp_swin_rows_block, round 5) -- proj + residual, norm2, fc1 + GELU, fc2 + residual and the NEXT block's norm1 + qkv
# -- once a launch has at least this many tokens (a workgroup streams the chain's 6 MB of weights for its 80 tokens: it needs several
# workgroups per compute unit to pay; below, the launch-per-Linear path)
ROWS_BLOCK_MIN_TOKENS = env_int('OMP355_MLP_ROWS_MIN', 32768, 1, 1 << 30)


class _Block(object):
    __slots__ = ('n1g', 'n1b', 'qkv_w', 'qkv_b', 'table', 'proj_w', 'proj_b', 'n2g', 'n2b', 'fc1_w',
                 'fc1_b', 'fc2_w', 'fc2_b', 'shift', 'mlp_pack', 'bias_exp', 'attn_fused', 'attn_pack', 'rows')


class _Stage(object):
    __slots__ = ('C', 'nH', 'blocks', 'down_g', 'down_b', 'down_w', 'out_g', 'out_b', 'rows_qkv0')


class Encoder(object):
    """Packs the backbone/FPN/input_proj weights once (matrices -> engine dtype, vectors stay fp32)
    and runs images -> (memory, memory+pos, key padding mask)."""

    def __init__(self, sd, args, swin_cfg, dtype, x3=False):
        self.dtype = dtype
        self.x3 = bool(x3)
        if self.x3 and dtype != torch.float32:
            raise ValueError('bf16x3 products run inside the fp32 engine')
        self.args = args
        self.window = swin_cfg['window']
        self.embed_dim = swin_cfg['embed_dim']
        f32 = lambda k: sd[k].detach().float().contiguous()          # noqa: E731
        # The bf16 matrix-core GEMMs step K in 64-element tiles (csrc/gemm.hip).  A width that does not divide (BASELINE config 1's
        # Swin-T: stage 0 is 96 wide) gets its WEIGHT image zero-padded to the next multiple of 64 (round 4; it was refused before):
        # zero columns cost no accuracy; the activation is zero-padded to the same pitch (see _gemm).
        def pad64(w):
            k = w.shape[1]
            return w if k % 64 == 0 else torch.nn.functional.pad(w, (0, 64 - k % 64))
        if self.x3:   # [w_hi | w_hi | w_lo] images of the fp32 matrices (ops.split_weight3), the IMAGE padded to K' % 64 == 0
            def mat(k):
                w = sd[k].detach().float()
                return pad64(ops.split_weight3(w.reshape(w.shape[0], -1))).contiguous()
        elif dtype == torch.bfloat16:
            def mat(k):   # every matrix this encoder multiplies with is [out, in] after flattening (1x1 convolutions included)
                w = sd[k].detach().to(dtype)
                return pad64(w.reshape(w.shape[0], -1)).contiguous()
        else:
            mat = lambda k: sd[k].detach().to(dtype).contiguous()    # noqa: E731
        bb = 'backbone.0.'
        self.pe_w = f32(bb + 'patch_embed.proj.weight').reshape(self.embed_dim, 48).contiguous()
        self.pe_b = f32(bb + 'patch_embed.proj.bias')
        self.pe_g = f32(bb + 'patch_embed.norm.weight')
        self.pe_bt = f32(bb + 'patch_embed.norm.bias')
        self.stages = []
        depths, heads = swin_cfg['depths'], swin_cfg['num_heads']
        for s, (dep, nh) in enumerate(zip(depths, heads)):
            st = _Stage()
            st.C, st.nH, st.blocks = self.embed_dim << s, nh, []
            if st.C != nh * 32:
                raise ValueError('libomp355 window attention is built for head_dim 32 (C=%d, heads=%d)' % (st.C, nh))
            for b in range(dep):
                p = '%slayers.%d.blocks.%d.' % (bb, s, b)
                blk = _Block()
                blk.n1g, blk.n1b = f32(p + 'norm1.weight'), f32(p + 'norm1.bias')
                blk.qkv_w, blk.qkv_b = mat(p + 'attn.qkv.weight'), f32(p + 'attn.qkv.bias')
                blk.table = f32(p + 'attn.relative_position_bias_table')
                blk.bias_exp = ops.swin_expand_bias(blk.table)   # gathered once (the reference re-gathers table[index] per call)
                blk.proj_w, blk.proj_b = mat(p + 'attn.proj.weight'), f32(p + 'attn.proj.bias')
                blk.n2g, blk.n2b = f32(p + 'norm2.weight'), f32(p + 'norm2.bias')
                blk.fc1_w, blk.fc1_b = mat(p + 'mlp.fc1.weight'), f32(p + 'mlp.fc1.bias')
                blk.fc2_w, blk.fc2_b = mat(p + 'mlp.fc2.weight'), f32(p + 'mlp.fc2.bias')
                blk.shift = 0 if b % 2 == 0 else self.window // 2
                fuse = (dtype == torch.bfloat16 and st.C in FUSED_MLP_WIDTHS and blk.fc1_w.shape[0] % 32 == 0
                        and getattr(args, 'fused_mlp', True))
                blk.mlp_pack = pack_mlp(blk.fc1_w, blk.fc1_b, blk.fc2_w) if fuse else None
                blk.rows = None
                # norm1 + qkv + (S)W-MSA + proj + residual in one launch where the kernel is built (C = 128 with 4 heads: Swin-B stage 0)
                blk.attn_fused = (dtype == torch.bfloat16 and not self.x3 and st.C == 128 and nh == 4 and self.window == 7
                                  and getattr(args, 'fused_attn', True))
                # C = 256 with 8 heads (stage 1): the same block streaming a fragment-major weight image (model/packing.py::pack_attn_block)
                blk.attn_pack = (pack_attn_block(blk.qkv_w, blk.proj_w, nh) if (dtype == torch.bfloat16 and not self.x3 and st.C == 256 and nh == 8
                                                                                and self.window == 7 and getattr(args, 'fused_attn', True)) else None)
                st.blocks.append(blk)
            if s + 1 < len(depths):
                p = '%slayers.%d.downsample.' % (bb, s)
                st.down_g, st.down_b = f32(p + 'norm.weight'), f32(p + 'norm.bias')
                st.down_w = mat(p + 'reduction.weight')
            else:
                st.down_g = st.down_b = st.down_w = None
            st.out_g, st.out_b = f32('%snorm%d.weight' % (bb, s)), f32('%snorm%d.bias' % (bb, s))
            # row-owner chains of a C = 512 stage: packed weight streams per block (its proj + MLP + the next block's qkv) and the first block's
            # qkv; the bf16 engine packs its bf16 matrices (csrc/dec_rows.hip), the parity engine the fp32 masters as (hi, lo) fragment pairs
            # (csrc/dec_rows_x3.hip)
            st.rows_qkv0 = None
            if ((dtype == torch.bfloat16 or self.x3) and st.C == 512 and all(tuple(sd['%slayers.%d.blocks.%d.mlp.fc1.weight' % (bb, s, i)].shape) == (2048, 512) for i in range(dep))
                    and getattr(args, 'fused_mlp', True) and getattr(args, 'fused_attn', True)):
                if self.x3:
                    rw = lambda i, leaf: f32('%slayers.%d.blocks.%d.%s' % (bb, s, i, leaf))                 # noqa: E731
                else:
                    rw = lambda i, leaf: sd['%slayers.%d.blocks.%d.%s' % (bb, s, i, leaf)].detach().to(dtype).contiguous()   # noqa: E731
                st.rows_qkv0 = pack_rows_embed_qkv(rw(0, 'attn.qkv.weight'))
                for i, b_ in enumerate(st.blocks):
                    args_ = (rw(i, 'attn.proj.weight'), rw(i, 'mlp.fc1.weight'), rw(i, 'mlp.fc2.weight'))
                    b_.rows = pack_rows_ffn_qkv(*args_, rw(i + 1, 'attn.qkv.weight')) if i + 1 < dep else pack_rows_ffn(*args_)
            self.stages.append(st)
        self.use_fpn = bool(args.use_fpn)
        if self.use_fpn:
            if len(self.stages) != 4:
                raise ValueError('FPN needs 4 backbone stages')
            # fpn_in[0] consumes c5 ... fpn_in[3] consumes c2 (reference fpn.py:18-19)
            self.fpn_w = [mat('fpn.fpn_in.%d.weight' % i).reshape(256, -1).contiguous() for i in range(4)]
        self._pos_cache = {}   # (B, h, w, device) -> sine embedding of an all-False mask (encode(no_padding=True))
        self.proj_w = mat('input_proj.weight').reshape(args.tfm_hidden_dim, -1).contiguous()
        self.proj_b = f32('input_proj.bias')

    @staticmethod
    def _gemm(A, W, bias=None, **kw):
        """ops.gemm for a weight whose K was zero-padded to a multiple of 64 (bf16 engine, widths like Swin-T's 96).  The activation
        gets the same treatment: it is copied into a zero-filled operand of the padded pitch, so a row's product reads its OWN
        columns and zeros -- never a neighbour's values (round 4 read the next row against the zero weights: one Inf / NaN there
        poisoned this row, 0 x Inf = NaN).  One extra pass over a 96-wide tensor, on the Swin-T path only."""
        if W.shape[1] != A.shape[-1] and not kw.get('a_wrap'):
            Ap = torch.zeros((A.shape[0], W.shape[1]), dtype=A.dtype, device=A.device)
            Ap[:, :A.shape[-1]].copy_(A)
            A = Ap
        return ops.gemm(A, W, bias, **kw)

    @staticmethod
    def _rows(rows, C, dtype, device):
        """GEMM operand buffer [rows, C]."""
        return torch.empty((rows, C), dtype=dtype, device=device)

    # -- Swin ---------------------------------------------------------------------------------
    def _backbone_x3(self, img, want_f32):
        """The bf16x3 engine's backbone: fp32 residual stream x, split-pair GEMM operands, fp32 window attention.
        -> list of (normed map as split pairs [B*h*w, 2C], h, w, fp32 copy or None) per stage."""
        B = img.shape[0]
        S = ops.SPLIT
        x, H, W = ops.patch_embed_ln(img, self.pe_w, self.pe_b, self.pe_g, self.pe_bt, torch.float32, LN_EPS)
        x = x.view(B * H * W, self.embed_dim)
        outs = []
        for st in self.stages:
            C = st.C
            blks = st.blocks
            if st.rows_qkv0 is not None and x.shape[0] >= ROWS_BLOCK_MIN_TOKENS:
                # stage 2 as row-owner chains over split operands (csrc/dec_rows_x3.hip): per block the split-product window attention on fp32
                # q | k | v, then ONE launch for proj, norm2, fc1 + GELU, fc2 and the next block's norm1 + qkv
                qkv = ops.swin_rows_qkv(x, (blks[0].n1g, blks[0].n1b), blks[0].qkv_b, st.rows_qkv0[0], st.rows_qkv0[1], eps=LN_EPS, x3=True)
                att = torch.empty((x.shape[0], 2 * C), dtype=torch.bfloat16, device=x.device)
                for i, blk in enumerate(blks):
                    ops.swin_window_attn(qkv, blk.qkv_b, blk.table, B, H, W, C, st.nH, blk.shift, out=att, window=self.window,
                                         bias_expanded=blk.bias_exp, out_split=True)
                    nxt = blks[i + 1] if i + 1 < len(blks) else None
                    ops.swin_rows_block(x, att, blk.rows[0], blk.rows[1], blk.proj_b, (blk.n2g, blk.n2b), blk.fc1_b, blk.fc2_b,
                                        next_n1=(nxt.n1g, nxt.n1b) if nxt is not None else None, next_qkv_b=nxt.qkv_b if nxt is not None else None,
                                        qkv=qkv, eps=LN_EPS, x3=True)
                blks = ()
            for blk in blks:
                y = ops.layernorm(x, blk.n1g, blk.n1b, out_dtype=S, eps=LN_EPS)
                qkv = ops.gemm(y, blk.qkv_w, blk.qkv_b, out_dtype=torch.float32, a_wrap=2 * C)
                att = ops.swin_window_attn(qkv, blk.qkv_b, blk.table, B, H, W, C, st.nH, blk.shift, out=y, window=self.window,
                                           bias_expanded=blk.bias_exp, out_split=True)
                ops.gemm(att, blk.proj_w, blk.proj_b, residual=x, out=x, a_wrap=2 * C)
                y = ops.layernorm(x, blk.n2g, blk.n2b, out=y, out_dtype=S, eps=LN_EPS)
                h = ops.gemm(y, blk.fc1_w, blk.fc1_b, act=ops.ACT_GELU, out_dtype=S, a_wrap=2 * C)
                ops.gemm(h, blk.fc2_w, blk.fc2_b, residual=x, out=x, a_wrap=8 * C)
            f32c = torch.empty_like(x) if want_f32 else None
            outs.append((ops.layernorm(x, st.out_g, st.out_b, out_dtype=S, out_f32=f32c, eps=LN_EPS), H, W, f32c))
            if st.down_w is not None:
                y, H2, W2 = ops.patch_merge_gather_ln(x, st.down_g, st.down_b, B, H, W, C, LN_EPS, out_dtype=S)
                x = ops.gemm(y, st.down_w, out_dtype=torch.float32, a_wrap=8 * C)
                H, W = H2, W2
        return outs

    def backbone(self, img):
        """img [B,3,H,W] fp32 -> list of (normed map [B*h*w, C], h, w) per stage."""
        if self.x3:
            return [(f, h, w) for _, h, w, f in self._backbone_x3(img, True)]
        B = img.shape[0]
        # The residual stream x is fp32 in EVERY engine (round 3): in the bf16 engine only GEMM / attention operands are bf16
        # (LayerNorm outputs, qkv, attention output, the MLP hidden), so 48 residual adds no longer round to 8 mantissa bits
        # each -- measured: stage-map error vs the reference halves (profiles/r03*_parity_report*.json).
        T = self.dtype
        x, H, W = ops.patch_embed_ln(img, self.pe_w, self.pe_b, self.pe_g, self.pe_bt, torch.float32, LN_EPS)
        x = x.view(B * H * W, self.embed_dim)
        outs = []
        for si, st in enumerate(self.stages):
            C = st.C
            y = None   # LayerNorm operand buffer of the unfused halves (a block whose attention half is fused never made one)
            if st.rows_qkv0 is not None and x.shape[0] >= ROWS_BLOCK_MIN_TOKENS:
                # two launches per block: window attention on q | k | v, then everything else of the block (and the next block's qkv) as a chain
                blks = st.blocks
                qkv = ops.swin_rows_qkv(x, (blks[0].n1g, blks[0].n1b), blks[0].qkv_b, st.rows_qkv0[0], st.rows_qkv0[1], eps=LN_EPS)
                att = torch.empty((x.shape[0], C), dtype=T, device=x.device)
                for i, blk in enumerate(blks):
                    ops.swin_window_attn(qkv, blk.qkv_b, blk.table, B, H, W, C, st.nH, blk.shift, out=att, window=self.window, bias_expanded=blk.bias_exp)
                    nxt = blks[i + 1] if i + 1 < len(blks) else None
                    ops.swin_rows_block(x, att, blk.rows[0], blk.rows[1], blk.proj_b, (blk.n2g, blk.n2b), blk.fc1_b, blk.fc2_b,
                                        next_n1=(nxt.n1g, nxt.n1b) if nxt is not None else None, next_qkv_b=nxt.qkv_b if nxt is not None else None,
                                        qkv=qkv, eps=LN_EPS)
                blks = ()
            else:
                blks = st.blocks
            for blk in blks:
                if blk.attn_fused:
                    ops.swin_attn_block(x, blk.n1g, blk.n1b, blk.qkv_w, blk.qkv_b, blk.bias_exp, blk.proj_w, blk.proj_b, B, H, W, C, st.nH,
                                        blk.shift, window=self.window, eps=LN_EPS)
                elif blk.attn_pack is not None:
                    ops.swin_attn_block_packed(x, blk.n1g, blk.n1b, blk.attn_pack, blk.qkv_b, blk.bias_exp, blk.proj_b, B, H, W, C, st.nH,
                                               blk.shift, window=self.window, eps=LN_EPS)
                else:
                    if y is None:
                        y = self._rows(x.shape[0], C, T, x.device)
                    y = ops.layernorm(x, blk.n1g, blk.n1b, out=y, out_dtype=T, eps=LN_EPS)
                    qkv = self._gemm(y, blk.qkv_w, blk.qkv_b)
                    att = ops.swin_window_attn(qkv, blk.qkv_b, blk.table, B, H, W, C, st.nH, blk.shift, out=y,
                                               window=self.window, bias_expanded=blk.bias_exp)
                    self._gemm(att, blk.proj_w, blk.proj_b, residual=x, out=x)
                if blk.mlp_pack is not None:   # norm2 + fc1 + GELU + fc2 + residual in one launch, in place
                    ops.swin_mlp_fused(x, blk.n2g, blk.n2b, blk.mlp_pack, blk.fc2_b, out=x, eps=LN_EPS)
                else:
                    if y is None:
                        y = self._rows(x.shape[0], C, T, x.device)
                    y = ops.layernorm(x, blk.n2g, blk.n2b, out=y, out_dtype=T, eps=LN_EPS)
                    h = self._gemm(y, blk.fc1_w, blk.fc1_b, act=ops.ACT_GELU)
                    self._gemm(h, blk.fc2_w, blk.fc2_b, residual=x, out=x)
            outs.append((ops.layernorm(x, st.out_g, st.out_b, out=self._rows(x.shape[0], C, T, x.device), out_dtype=T, eps=LN_EPS), H, W))
            if st.down_w is not None:
                ym, H2, W2 = ops.patch_merge_gather_ln(x, st.down_g, st.down_b, B, H, W, C, LN_EPS, out_dtype=T)
                x = self._gemm(ym, st.down_w, out_dtype=torch.float32)
                H, W = H2, W2
        return outs

    @staticmethod
    def level_mask(mask, h, w):
        """Padding mask (uint8, 1 = padding) at a feature level: nearest resize exactly as swin_transformer.py:621."""
        m8 = mask if mask.dtype == torch.uint8 else mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        return ops.mask_nearest(m8.contiguous(), h, w)

    # -- full encode ----------------------------------------------------------------------------
    def encode(self, img, mask, want_intermediates=False, no_padding=False, out=None):
        """-> dict(memory [B*M,d], mem_pos [B*M,d], key_mask uint8 [B,M] or None, M, hw).
        no_padding: the caller knows `mask` is all False -> the sine embedding (a function of the mask only) is taken from a
        per-shape cache.  out: optional (memory, mem_pos) destination views of a larger engine call (_encode_chunked)."""
        B = img.shape[0]
        x3 = self.x3
        if x3:
            sfeats = self._backbone_x3(img, want_intermediates)
            feats = [(f, h, w) for _, h, w, f in sfeats]
        else:
            feats = self.backbone(img)
        if self.use_fpn:
            (c2, h2, w2), (c3, h3, w3), (c4, h4, w4), (c5, h5, w5) = feats
            if x3:
                lat = lambda i, lvl: ops.gemm(sfeats[lvl][0], self.fpn_w[i], out_dtype=torch.float32, a_wrap=sfeats[lvl][0].shape[1])  # noqa: E731
                l5, l4, l3, l2 = lat(0, 3), lat(1, 2), lat(2, 1), lat(3, 0)
            else:
                # through _gemm: a stage map whose width is not a multiple of 64 (Swin-T's c2: 96) meets a zero-padded weight image
                l5 = self._gemm(c5, self.fpn_w[0])
                l4 = self._gemm(c4, self.fpn_w[1])
                l3 = self._gemm(c3, self.fpn_w[2])
                l2 = self._gemm(c2, self.fpn_w[3])
            sizes = ((h2, w2), (h3, w3), (h4, w4), (h5, w5))
            src, ho, wo = ops.fpn_fuse(l2, l3, l4, l5, B, sizes, 2)
            lvl = (h4, w4)
            if (ho, wo) != lvl:
                raise RuntimeError('stride-2 projection grid %s != stage-2 grid %s' % ((ho, wo), lvl))
        else:
            src, ho, wo = feats[-1]
            if x3:
                src = sfeats[-1][0]   # already split pairs
            lvl = (ho, wo)
        lm = self.level_mask(mask, *lvl)
        M = ho * wo
        pkey = (B, ho, wo, src.device)
        hit = self._pos_cache.get(pkey) if no_padding else None
        if hit is not None:
            # produced on another stream (pipeline lanes share the encoder): order this stream after its kernel
            pos, ready = hit
            torch.cuda.current_stream().wait_event(ready)
        else:
            pos = ops.sine_posembed(lm, self.args.tfm_hidden_dim // 2, self.dtype).view(B * M, -1)
            if no_padding:
                ready = torch.cuda.Event()
                ready.record()
                if len(self._pos_cache) >= 8:
                    self._pos_cache.clear()
                self._pos_cache[pkey] = (pos, ready)
        # ONE product, two destinations: memory = src W^T + b and memory + pos (the key input of every decoder layer)
        if out is not None:
            memory, mem_pos_out = out
        else:
            memory, mem_pos_out = torch.empty((B * M, self.proj_w.shape[0]), dtype=self.dtype, device=src.device), None
        gk = {}
        if x3:
            ssrc = src if src.dtype == torch.bfloat16 else ops.split_bf16(src)   # FPN output: fp32 -> split pairs
            gk = dict(a_wrap=ssrc.shape[1], out_dtype=torch.float32)
        else:
            ssrc = src
        if B * M > 64:
            mem_pos = ops.gemm(ssrc, self.proj_w, self.proj_b, residual=pos, out_noresidual=memory, out=mem_pos_out, **gk)
        else:   # tiny inputs run on the small-M kernels, which have no second destination
            ops.gemm(ssrc, self.proj_w, self.proj_b, out=memory, **gk)
            mem_pos = ops.gemm(ssrc, self.proj_w, self.proj_b, residual=pos, out=mem_pos_out, **gk)
        out = dict(memory=memory, mem_pos=mem_pos, M=M, hw=(ho, wo), pos=pos, key_mask=lm.reshape(B, M))
        if want_intermediates:
            out['feats'] = feats
            out['src'] = src if not (x3 and not self.use_fpn) else feats[-1][0]
            if self.use_fpn:
                out['src_full'] = ops.fpn_fuse(l2, l3, l4, l5, B, sizes, 1)[0]
        return out
