"""ctypes binding of libomp355.so (the C ABI declared in include/omp355.h).

There is NO fallback: if the shared library is missing the import fails loudly, and every op
raises RuntimeError(omp_last_error()) on a non-zero return code.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libomp355.so')

OMP_F32, OMP_BF16, OMP_BF16X2 = 0, 1, 2   # BF16X2: split-bf16 pair rows [hi | lo] (include/omp355.h)
ABI_VERSION = 21
STORE_PLAIN, STORE_KBLK, STORE_VBLK, STORE_ROWSTAT = 0, 2, 3, 4
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
DEC_PT, DEC_POLY, DEC_REC = 0, 1, 2
MAX_DEC_LAYERS = 8
MAX_GRAPH_SLOTS = 4096   # csrc/decoder.hip: fixed table of cached hipGraphs

c_void_p, c_int, c_int32, c_int64, c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int32,
                                              ctypes.c_int64, ctypes.c_float)


class GemmArgs(ctypes.Structure):
    _fields_ = [('A', c_void_p), ('lda', c_int64), ('W', c_void_p), ('ldw', c_int64),
                ('bias', c_void_p), ('bias_row', c_void_p), ('bias_row_stride', c_int64),
                ('residual', c_void_p), ('ldr', c_int64), ('C', c_void_p), ('ldc', c_int64),
                ('M', c_int64), ('N', c_int32), ('K', c_int32), ('dtype', c_int32),
                ('out_dtype', c_int32), ('act', c_int32), ('trans_out', c_int32),
                ('trans_rows', c_int64), ('trans_ld', c_int64), ('ln_gamma', c_void_p), ('ln_beta', c_void_p),
                ('ln_eps', c_float), ('small_m_splitk', c_int32), ('store_mode', c_int32), ('bias_along_m', c_int32),
                ('kv_images', c_int32), ('kv_tokens', c_int32), ('kv_mpad', c_int32), ('kv_heads', c_int32),
                ('kv_key_block', c_int32), ('C2', c_void_p), ('ldc2', c_int64), ('a_wrap', c_int32)]


class SampleCfg(ctypes.Structure):
    _fields_ = [('kind', c_int32), ('num_bins', c_int32), ('pt_eos', c_int32), ('poly_eos', c_int32),
                ('rec_eos', c_int32), ('vocab', c_int32), ('vie_categories', c_int32),
                ('infer_vie', c_int32), ('suppress_eos', c_int32), ('step0', c_int32)]


class DecLayer(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        'sa_in_w', 'sa_bias_tab', 'sa_out_w', 'sa_out_b', 'ca_q_w', 'ca_qbias_tab', 'ca_out_w',
        'ca_out_b', 'ff1_w', 'ff1_b', 'ff2_w', 'ff2_b', 'n1_g', 'n1_b', 'n2_g', 'n2_b', 'n3_g', 'n3_b',
        'kcache', 'vcache', 'crossK', 'crossVt', 'rows_mid', 'rows_ffn')] + [('rows_mid_stride', c_int64), ('rows_ffn_stride', c_int64)]


class DecRowsArgs(ctypes.Structure):
    """omp_dec_rows_args (include/omp355.h): the row-owner chains of the decoders' many-row phases"""
    _fields_ = [('R', c_int32), ('eps', c_float), ('d_pos', c_void_p), ('x', c_void_p), ('att', c_void_p), ('wstream', c_void_p),
                ('wave_stride', c_int64), ('out_b', c_void_p), ('ln_g', c_void_p), ('ln_b', c_void_p), ('qbias_tab', c_void_p), ('q', c_void_p),
                ('prologue', c_int32), ('tail', c_int32), ('ff1_b', c_void_p), ('ff2_b', c_void_p), ('seq', c_void_p), ('seq_ld', c_int32),
                ('word_emb', c_void_p), ('pos_tab', c_void_p), ('emb_g', c_void_p), ('emb_b', c_void_p), ('lnt_g', c_void_p), ('lnt_b', c_void_p),
                ('bias_tab', c_void_p), ('qkv', c_void_p), ('h0_b', c_void_p), ('h1_b', c_void_p), ('h2_b', c_void_p), ('logits', c_void_p),
                ('vocab', c_int32), ('x3', c_int32), ('xcd_mask', c_int32)]


class SwinRowsArgs(ctypes.Structure):
    """omp_swin_rows_args (include/omp355.h): a Swin stage-2 block minus its window attention core as one row-owner chain"""
    _fields_ = [('M', c_int64), ('eps', c_float), ('mode', c_int32), ('x', c_void_p), ('att', c_void_p), ('qkv', c_void_p), ('wstream', c_void_p),
                ('wave_stride', c_int64), ('proj_b', c_void_p), ('n2_g', c_void_p), ('n2_b', c_void_p), ('fc1_b', c_void_p), ('fc2_b', c_void_p),
                ('n1_g', c_void_p), ('n1_b', c_void_p), ('qkv_b', c_void_p), ('x3', c_int32)]


class DecoderPlan(ctypes.Structure):
    _fields_ = ([(n, c_int32) for n in ('dtype', 'n_layers', 'd_model', 'n_heads', 'd_ff', 'vocab',
                                        'pre_norm', 'R', 'Lmax', 'M', 'Mpad', 'n_tiles', 'q_tiles', 'n_split', 'n_prompt', 'gemm_x3', 'kv_split', 'rows_fused', 'rows_xcd_mask')]
                + [('eps', c_float), ('rows_embed', c_void_p), ('rows_embed_stride', c_int64), ('layers', DecLayer * MAX_DEC_LAYERS)]
                + [(n, c_void_p) for n in ('word_emb', 'pos_tab', 'emb_g', 'emb_b', 'fn_g', 'fn_b',
                                           'h0_w', 'h1_w', 'h2_w', 'h0_b', 'h1_b', 'h2_b')]
                + [('kv_img_stride', c_int64)]
                + [('key_mask', c_void_p), ('tiles', c_void_p), ('seq', c_void_p), ('seq_ld', c_int32),
                   ('d_pos', c_void_p), ('probs', c_void_p), ('finished', c_void_p), ('lengths', c_void_p)]
                + [(n, c_void_p) for n in ('x', 'x2', 'y', 'qkv', 'att', 'q', 'ffh', 'hh0', 'hh1',
                                           'partial', 'logits')]
                + [('sample', SampleCfg)])


_SIGS = {
    'omp_abi_version': (c_int, []),
    'omp_layernorm': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64,
                              c_int, c_float, c_void_p]),
    'omp_gemm_bias_act': (c_int, [ctypes.POINTER(GemmArgs), c_void_p]),
    'omp_swin_mlp_fused': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                   c_int, c_int, c_void_p]),
    'omp_swin_mlp_fused2': (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                    c_int, c_int, c_void_p]),
    'omp_debug_swin_mlp_variant': (c_int, [c_int]),
    'omp_debug_swin_mlp_trace': (c_int, [c_void_p]),
    'omp_patch_embed_ln': (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_float, c_void_p]),
    'omp_swin_window_attn': (c_int, [c_void_p] * 4 + [c_int] * 8 + [c_void_p]),
    'omp_swin_window_attn2': (c_int, [c_void_p] * 5 + [c_int] * 9 + [c_void_p]),
    'omp_swin_expand_bias': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'omp_swin_attn_block': (c_int, [c_void_p] * 4 + [c_float] + [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    'omp_swin_attn_block_packed': (c_int, [c_void_p] * 4 + [c_float] + [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    'omp_patch_merge_gather_ln': (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_float, c_void_p]),
    'omp_patch_merge_gather_ln2': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_float, c_void_p]),
    'omp_split_bf16': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    'omp_fpn_fuse': (c_int, [c_void_p] * 5 + [c_int] * 11 + [c_void_p]),
    'omp_mask_nearest': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'omp_sine_posembed': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    'omp_dec_embed_ln': (c_int, [c_void_p, c_int] + [c_void_p] * 7 + [c_int, c_int, c_int, c_float, c_void_p]),
    'omp_dec_self_attn_step': (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    'omp_dec_cross_attn_step': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                        c_int, c_void_p]),
    'omp_head_softmax_mask_argmax': (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(SampleCfg), c_void_p,
                                             c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'omp_pack_spotting': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p]),
    'omp_dec_rows_mid': (c_int, [ctypes.POINTER(DecRowsArgs), c_void_p]),
    'omp_dec_rows_ffn': (c_int, [ctypes.POINTER(DecRowsArgs), c_void_p]),
    'omp_dec_rows_tile': (c_int, []),
    'omp_kv_project_rows': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'omp_swin_rows_block': (c_int, [ctypes.POINTER(SwinRowsArgs), c_void_p]),
    'omp_decoder_run': (c_int, [ctypes.POINTER(DecoderPlan), c_int, c_int, c_int, c_void_p]),
    'omp_decoder_run_pair': (c_int, [ctypes.POINTER(DecoderPlan), ctypes.POINTER(DecoderPlan), c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'omp_decoder_graph_reset': (c_int, [c_int]),
    'omp_capture_gate_enter': (c_int, []),
    'omp_capture_gate_leave': (c_int, []),
    'omp_decoder_step_logits': (c_int, [ctypes.POINTER(DecoderPlan), c_int, c_void_p]),
    'omp_debug_force_gemm_kernel': (c_int, [c_int]),
    'omp_debug_gemm_choice': (c_int, [c_void_p]),
    'omp_debug_set_gemm_trace': (c_int, [c_void_p, c_int64]),
    'omp_debug_swin_attn_impl': (c_int, [c_int]),
    'omp_debug_rows_tile': (c_int, [c_int]),
    'omp_debug_rows_tile_choice': (c_int, [c_int, c_int]),
    'omp_debug_cross_q4': (c_int, [c_int]),
    'omp_debug_cross_nt': (c_int, [c_int]),
    'omp_debug_self_attn_impl': (c_int, [c_int]),
    'omp_debug_dec_fused': (c_int, [c_int]),
    'omp_vit_patch_embed': (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    'omp_vit_attn_qkv': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    'omp_vit_attn': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    'omp_a3_pool': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'omp_row_stat_merge': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'omp_row_argmax_prob': (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'omp_resize_normalize_pad': (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                        c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'omp_ctx_create': (c_int, [ctypes.POINTER(c_void_p)]),
    'omp_ctx_destroy': (c_int, [c_void_p]),
    'omp_ctx_make_current': (c_int, [c_void_p]),
    'omp_ctx_current': (c_void_p, []),
    'omp_stream_create_cu_mask': (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p)]),
    'omp_stream_destroy': (c_int, [c_void_p]),
    'omp_debug_where': (c_int, [c_void_p, c_int, c_void_p]),
    'omp_prof_enable': (c_int, [c_int]),
    'omp_prof_read': (c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64)]),
    'omp_prof_read_roofline': (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    'omp_prof_read_class': (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64), ctypes.POINTER(ctypes.c_double)]),
}
EXPORTS = sorted(list(_SIGS) + ['omp_last_error'])

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the library was not built."""
    global _lib
    if _lib is None:
        # PyTorch ships its own HIP runtime and loads it into the global symbol scope.  libomp355.so must be
        # dlopen'ed AFTER it: its kernels register themselves (at load time) with whichever runtime the loader
        # binds first, and every stream / pointer it is handed comes from torch's.  Loaded the other way round the
        # kernels sit in /opt/rocm's runtime while later calls bind to torch's -> "invalid device function".
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libomp355.so not found at %s -- build it with `python -m advancedliteratemachinery_amd.build` '
                '(there is no CPU/PyTorch fallback for the OmniParser hot path)' % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        h.omp_last_error.restype = ctypes.c_char_p
        h.omp_last_error.argtypes = []
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.omp_abi_version() != ABI_VERSION:
            raise RuntimeError('libomp355.so ABI version mismatch')
        _lib = h
    return _lib


def check(rc, what=''):
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (what or 'libomp355', rc,
                                                   lib().omp_last_error().decode('utf-8', 'replace')))
