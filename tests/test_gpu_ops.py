"""GPU parity tests of the individual libomp355 entry points (through the C ABI) vs the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _assert_all(records):
    bad = [r for r in records if not r['ok']]
    assert not bad, '\n'.join('%s: err=%.3e tol=%.1e %s' % (r['name'], r['err'], r['tol'], r['note']) for r in bad)


@pytest.fixture(scope='module')
def C():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from tests import gpu_checks
    return gpu_checks


@pytest.mark.parametrize('name', ['check_layernorm', 'check_gemm', 'check_mlp_fused', 'check_self_attn', 'check_gemm_small', 'check_patch_embed', 'check_window_attn',
                                  'check_patch_merge', 'check_fpn', 'check_posembed', 'check_sampling', 'check_cross_attn', 'check_split_ops', 'check_gemm_x3',
                                  'check_window_attn_split', 'check_swin_block', 'check_cross_attn_split', 'check_gemm_4w', 'check_dec_rows', 'check_swin_rows_block', 'check_dec_rows_x3', 'check_swin_rows_block_x3', 'check_dec_rows_tiles', 'check_kv_rows', 'check_dec_rows_xcd'])
def test_op(C, name):
    _assert_all(getattr(C, name)())


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
@pytest.mark.parametrize('pre_norm', [True, False])
def test_decoder_teacher_forced(C, dtype, pre_norm):
    _assert_all(C.check_decoder(dtype, pre_norm, with_mask=True))


@pytest.mark.parametrize('with_mask', [True, False])
def test_decoder_fused_few_row_kernels(C, with_mask):
    _assert_all(C.check_decoder_fused(with_mask))


@pytest.mark.parametrize('with_mask', [True, False])
def test_decoder_x3_many_row_phases(C, with_mask):
    _assert_all(C.check_decoder_x3(with_mask))


@pytest.mark.parametrize('with_mask', [True, False])
def test_decoder_row_owner_chains(C, with_mask):
    """round 5: the many-row phases' Linear layers as two launches per layer (csrc/dec_rows.hip)"""
    _assert_all(C.check_decoder_rows(with_mask))


def test_sampling_block_kernel(C):
    _assert_all(C.check_sampling_block())


def test_decoder_no_mask(C):
    _assert_all(C.check_decoder('fp32', True, with_mask=False))


def test_decoder_longest_sequence(C):
    """config 4 (long structured-sequence decode) up to the reference's 1024-entry position tables"""
    _assert_all(C.check_decoder_long('fp32'))


def test_decoder_longest_sequence_bf16_fused(C):
    """the same 1023-position sequence through the fused few-row kernels (16 key chunks, the prefetch ring wrapping 5 times)"""
    _assert_all(C.check_decoder_long('bf16'))


def test_contexts_isolate_state(C):
    """omp_ctx: private selectors and graph tables per context"""
    _assert_all(C.check_contexts('bf16'))


def test_cu_masked_stream(C):
    _assert_all(C.check_masked_stream())
