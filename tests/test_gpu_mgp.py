"""MGP-STR (BASELINE config 5) GPU parity: kernels of csrc/vit.hip, the ViT block on the shared kernels, end to end
against the oracle and against the golden fixture written by the reference's own code."""
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
    from tests import gpu_checks_mgp
    return gpu_checks_mgp


@pytest.mark.parametrize('name', ['check_vit_patch_embed', 'check_a3_pool', 'check_row_argmax_prob', 'check_vit_attn'])
def test_mgp_op(C, name):
    _assert_all(getattr(C, name)())


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'bf16x3'])
def test_vit_block(C, dtype):
    _assert_all(C.check_vit_block(dtype))


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'bf16x3'])
def test_mgp_end_to_end(C, dtype):
    _assert_all(C.check_mgp_e2e(dtype))


def test_mgp_golden_fp32(C):
    _assert_all(C.check_mgp_golden())


def test_mgp_golden_parity_engine(C):
    """round 5: MGP-STR's bf16x3 engine (split-bf16 products, split-plane attention) under the fp32 gates on the reference's fixture"""
    _assert_all(C.check_mgp_golden('bf16x3'))


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'bf16x3'])
def test_mgp_batch512_config5(C, dtype):
    """BASELINE config 5 at its stated shape (ViT-B, batch 512) against the oracle on probe rows."""
    _assert_all(C.check_mgp_b512(dtype))


def test_crop_resizer_bit_exact_with_pillow(C):
    _assert_all(C.check_crop_resizer())


def test_two_stage_pipeline_matches_oracle_chain(C):
    """OmniParser detections -> device crops -> MGP-STR -> fused decoding."""
    _assert_all(C.check_two_stage())


def test_gemm_row_statistics_equal_logits_then_argmax(C):
    """round 6: OMP_STORE_ROWSTAT + omp_row_stat_merge == fp32 logits + omp_row_argmax_prob (ids identical, probabilities to rounding)"""
    _assert_all(C.check_gemm_row_stats())


@pytest.mark.parametrize('dtype', ['bf16', 'bf16x3', 'fp32'])
def test_mgp_recognize_without_logits_tensor(C, dtype):
    """MGPSTR.recognize: the BPE / WordPiece heads decode from the head product's row statistics -- same ids, choice, text, confidences as
    logits + arg-max (test_final.py:145-240)"""
    _assert_all(C.check_mgp_greedy_fused(dtype))


def test_vit_attention_on_the_fused_qkv_projection(C):
    """round 6: one token-major q | k | v product + omp_vit_attn_qkv == three projections into blocked slabs + omp_vit_attn, bit for bit"""
    _assert_all(C.check_vit_attn_qkv())
