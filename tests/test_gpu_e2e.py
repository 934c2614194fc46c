"""End-to-end GPU parity: HIP engine vs fixtures produced by the REAL reference (tests/golden)."""
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


@pytest.mark.parametrize('name', ['spot_odd', 'spot_224', 'kie_sroie', 'postnorm_nofpn'])
def test_golden_fp32(C, name):
    """fp32 engine: logits within 1e-3 of the reference, decoded token ids identical."""
    _assert_all(C.check_e2e(name, 'fp32'))


@pytest.mark.parametrize('name', ['spot_odd', 'spot_224'])
def test_golden_bf16(C, name):
    """bf16 engine (the benchmarked precision): loose bounds; token agreement is reported, not gated."""
    _assert_all(C.check_e2e(name, 'bf16'))


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_batch_equals_single(C, dtype):
    _assert_all(C.check_batch_equivalence(dtype))


@pytest.mark.parametrize('name', ['spot_odd', 'kie_sroie'])
def test_golden_fp32_graph_and_overlap(C, name):
    """Same gate with decoder steps replayed as hipGraphs and poly || rec on two streams."""
    _assert_all(C.check_e2e(name, 'fp32', graph=True))


def test_graph_replay_matches_eager(C):
    _assert_all(C.check_graph_matches_eager('fp32'))


def test_pipelined_lanes_match_direct(C):
    """Batches in flight on several HIP streams (engine/pipeline.py) == the synchronous path."""
    _assert_all(C.check_lanes('fp32'))
