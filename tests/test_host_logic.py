"""CPU tests of the host-side logic around the hot path (no GPU): C-ABI export list, state-dict layout,
result formatting vs the reference's decode_seq, image sharding + all-gather under gloo (world 2)."""
import os
import sys
import types

import pytest
import torch
import torch.multiprocessing as mp

from advancedliteratemachinery_amd.utils import dist as udist
from advancedliteratemachinery_amd.utils.parser import make_args


def test_library_loads_and_exports_every_declared_symbol():
    import re
    from advancedliteratemachinery_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the product ABI and the development hooks (kept out of the public header) must both resolve
    declared = set()
    for rel in (('include', 'omp355.h'), ('advancedliteratemachinery_amd', 'csrc', 'omp355_debug.h')):
        found = set(re.findall(r'\b(omp_[a-z0-9_]+)\s*\(', open(os.path.join(root, *rel)).read()))
        assert found, 'no declarations parsed in %s' % (rel,)
        declared |= found
    public = set(re.findall(r'\b(omp_[a-z0-9_]+)\s*\(', open(os.path.join(root, 'include', 'omp355.h')).read()))
    assert not [n for n in public if n.startswith(('omp_debug_', 'omp_prof_'))], 'development hooks leaked into the public header'
    if not os.path.exists(_lib.LIB_PATH):
        from advancedliteratemachinery_amd import build
        build.build(verbose=False)
    h = _lib.lib()
    for name in declared:
        assert hasattr(h, name), 'libomp355.so does not export %s' % name
    assert h.omp_abi_version() == _lib.ABI_VERSION


def test_state_dict_layout_matches_reference_keys():
    from advancedliteratemachinery_amd.model import OmniParser, expected_state_dict
    from advancedliteratemachinery_amd.utils import synthetic as weights
    for kw in (dict(use_fpn=True), dict(use_fpn=False), dict(use_fpn=True, vie_categories=29)):
        args = make_args(tfm_pre_norm=True, **kw)
        spec = expected_state_dict(args)
        ora = weights.state_dict_spec(args)
        assert list(spec) == list(ora)
        assert all(tuple(spec[k][0]) == tuple(ora[k]) for k in ora)
    args = make_args(tfm_pre_norm=True, use_fpn=True)
    m = OmniParser(args, dict(depths=(2, 2, 2, 2)))
    sd = weights.make_state_dict(args, depths=(2, 2, 2, 2))
    m.load_state_dict(sd, strict=True)
    # the three decoder-norm key pairs alias ONE tensor, like the reference's shared nn.LayerNorm
    s = m.state_dict()
    assert s['transformer.pt_decoder.norm.weight'].data_ptr() == s['transformer.rec_decoder.norm.weight'].data_ptr()
    with pytest.raises(RuntimeError):
        m.engine()  # CPU model: must fail loudly, never fall back


def test_ops_refuse_cpu_tensors():
    from advancedliteratemachinery_amd import ops
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.zeros(4, 128), torch.ones(128), torch.zeros(128))


def _reference_decode_seq():
    from oracle import ref_import
    if not ref_import.available():
        return None
    for name in ('bezier',):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_misc', os.path.join(ref_import.REF_ROOT, 'utils', 'misc.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.decode_seq


def test_decode_seq_matches_reference():
    from advancedliteratemachinery_amd.utils.misc import decode_seq
    args = make_args()
    g = torch.Generator().manual_seed(0)
    pt = torch.randint(0, 1000, (10,), generator=g)
    poly = torch.randint(0, 1000, (5 * 32,), generator=g)
    rec = torch.randint(1000, 1100, (5, 25), generator=g)
    rec[0, 3] = args.rec_eos_index
    rec[1, 0] = args.recog_pad_index
    rec[2, 5] = args.recog_pad_index - 1
    probs = torch.rand(5, 25, generator=g)
    mine = (decode_seq(pt, args, 'pt'), decode_seq(poly, args, 'poly'), decode_seq(rec, args, 'rec', probs))
    # known answers
    assert mine[0][0]['point'] == ((pt[0] / 1000).item(), (pt[1] / 1000).item())
    assert mine[2][0][0]['rec'] == ''.join(args.chars[t - 1000] for t in rec[0, :3].tolist())
    assert mine[2][0][1]['rec'] == ''
    ref = _reference_decode_seq()
    if ref is None:
        return
    r = (ref(pt, args, 'pt', 'none'), ref(poly, args, 'poly', 'none'), ref(rec, args, 'rec', probs))
    assert mine[0] == r[0]
    assert all(torch.equal(a['polygon'], b['polygon']) for a, b in zip(mine[1], r[1]))
    assert mine[2][0] == r[2][0]
    assert all(abs(a - b) < 1e-6 for a, b in zip(mine[2][1], r[2][1]))


def test_shard_range_partitions_exactly():
    for n in (1, 7, 8, 32, 33):
        for ws in (1, 2, 3, 8):
            spans = [udist.shard_range(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_result(seed, n, rec_len):
    g = torch.Generator().manual_seed(seed)
    if n == 0:
        return None
    return ([torch.randint(0, 1000, (1, 2 * n), generator=g), torch.randint(0, 1000, (1, 32 * n), generator=g),
             torch.randint(1000, 1096, (1, n, rec_len), generator=g)], [torch.rand(n, rec_len, generator=g)])


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    assert udist.init_distributed_mode(backend='gloo')
    rec_len, nmax, total = 25, 6, 6
    lo, hi = udist.shard_range(total, rank, world)
    local = [_fake_result(100 + i, [3, 0, 6, 1, 2, 5][i], rec_len) for i in range(lo, hi)]
    ids, probs, n = udist.pack_results(local, nmax, rec_len, 'cpu')
    ids, probs, n = udist.all_gather_results(ids, probs, n)
    res = udist.unpack_results(ids, probs, n)
    ok = len(res) == total
    for i, r in enumerate(res):
        exp = _fake_result(100 + i, [3, 0, 6, 1, 2, 5][i], rec_len)
        if exp is None:
            ok &= r is None
            continue
        ok &= all(torch.equal(a, b) for a, b in zip(r[0], exp[0])) and torch.equal(r[1][0], exp[1][0])
    q.put((rank, bool(ok)))
    torch.distributed.destroy_process_group()


def test_image_sharded_gather_world2_gloo():
    """N > 1 path: every rank ends up with every image's result, in global image order."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(got) == [(0, True), (1, True)]


def test_decode_pred_seq_record_format():
    from advancedliteratemachinery_amd.engine.inference import build_prompts, decode_pred_seq
    args = make_args(use_char_window_prompt=True)
    assert build_prompts(args)[0].tolist() == [[0, 0, 999, 999, 1000, 1095, 1100]]
    r = _fake_result(1, 2, 25)
    recs = decode_pred_seq([t[0] for t in r[0]], r[1][0], {'file_name': 'a.jpg', 'orig_size': (480, 640)}, args)
    assert len(recs) == 2 and set(recs[0]) == {'image_id', 'pts', 'score', 'polys', 'rec'}
    assert len(recs[0]['polys']) == 16 and abs(recs[0]['pts'][0][0] - r[0][0][0, 0].item() / 1000 * 640) < 1e-3


def test_erf_fast_twin_matches_scipy():
    """numpy twin of csrc/common.h erf_fast (the GELU epilogue's branch-free erf): same constants, fp32
    arithmetic, checked against scipy.special.erf -- pins the coefficients the kernel is built with."""
    import re
    import numpy as np
    from scipy.special import erf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'advancedliteratemachinery_amd', 'csrc', 'common.h')).read()
    body = src[src.index('float erf_fast(float a)'):src.index('// nn.GELU()')]
    consts = [float(c) for c in re.findall(r'(-?\d\.\d+e-?\d+)f', body)]
    assert len(consts) == 13, consts
    f = np.float32

    def fma(a, b, c):
        return (np.float64(a) * np.float64(b) + np.float64(c)).astype(f)
    a = np.linspace(-6, 6, 400001).astype(f)
    t, s_ = np.abs(a), (a * a).astype(f)
    r = fma(f(consts[0]), t, f(consts[1]))
    u = fma(f(consts[2]), t, f(consts[3]))
    r = fma(r, s_, u)
    for c in consts[4:7]:
        r = fma(r, t, f(c))
    r = fma(r, t, -t)
    big = np.copysign((1.0 - np.exp(r.astype(np.float64))).astype(f), a)
    q = np.full_like(a, consts[7])
    for c in consts[8:13]:
        q = fma(q, s_, f(c))
    small = fma(q, a, a)
    y = np.where(t > 0.927734375, big, small)
    assert np.abs(y.astype(np.float64) - erf(a.astype(np.float64))).max() < 1.5e-7


def test_gelu_fast_twin_matches_exact_gelu():
    """numpy twin of csrc/common.h gelu_fast2 (the bf16 engine's GELU: max(x,0) - 0.5|x| 2^(t q(t)), t = min(|x|, 6)):
    same constants, fp32 arithmetic, against 0.5 x (1 + erf(x / sqrt 2)) in float64 -- pins the coefficients."""
    import re
    import numpy as np
    from scipy.special import erf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'advancedliteratemachinery_amd', 'csrc', 'common.h')).read()
    body = src[src.index('f32x2 gelu_fast2(f32x2 x)'):src.index('// n (even) values in place')]
    consts = [float(c) for c in re.findall(r'\{(-?\d\.\d+e[-+]?\d+)f,', body)]
    assert len(consts) == 6, consts
    f = np.float32

    def fma(a, b, c):
        return (np.float64(a) * np.float64(b) + np.float64(c)).astype(f)
    x = np.linspace(-9, 9, 600001).astype(f)
    a = np.abs(x)
    t = np.minimum(a, f(6.0))
    q = np.full_like(x, consts[0])
    for c in consts[1:]:
        q = fma(q, t, f(c))
    p = (q * t).astype(f)
    e = np.exp2(p.astype(np.float64)).astype(f)
    y = fma((a * f(-0.5)).astype(f), e, np.maximum(x, f(0)))
    ref = 0.5 * x.astype(np.float64) * (1.0 + erf(x.astype(np.float64) / np.sqrt(2.0)))
    assert np.abs(y.astype(np.float64) - ref).max() < 5e-7


def test_encoder_chunking_concatenates_memory_in_image_order():
    """OmniParser._encode_chunked: a large engine call is encoded enc_chunk images at a time; memory rows / masks of
    the chunks are concatenated in image order and the per-batch scalars are kept"""
    from advancedliteratemachinery_amd.model.omniparser import OmniParser

    class FakeEnc(object):
        calls = []

        def encode(self, img, mask, out=None):
            b = img.shape[0]
            FakeEnc.calls.append(b)
            ids = img[:, 0, 0, 0].repeat_interleave(4)          # M = 4 memory rows per image, tagged with the image id
            memory, mem_pos = ids[:, None].float(), ids[:, None].float() + 0.5
            if out is not None:                                  # later chunks write straight into the call's tensors
                out[0].copy_(memory)
                out[1].copy_(mem_pos)
                memory, mem_pos = out
            return dict(memory=memory, mem_pos=mem_pos, pos=ids[:, None].float() * 0,
                        key_mask=mask[:, 0, :4].to(torch.uint8), M=4, hw=(2, 2))

    m = OmniParser.__new__(OmniParser)
    m.enc_chunk = 3
    img = torch.arange(8, dtype=torch.float32).reshape(8, 1, 1, 1).expand(8, 3, 2, 8).contiguous()
    mask = torch.zeros(8, 2, 8, dtype=torch.bool)
    mask[5] = True
    out = OmniParser._encode_chunked(m, FakeEnc(), img, mask)
    assert FakeEnc.calls == [3, 3, 2]
    assert out['memory'][:, 0].tolist() == [float(i) for i in range(8) for _ in range(4)]
    assert out['mem_pos'][:, 0].tolist() == [float(i) + 0.5 for i in range(8) for _ in range(4)]
    assert out['key_mask'].shape == (8, 4) and out['key_mask'][5].all() and not out['key_mask'][4].any()
    assert out['M'] == 4 and out['hw'] == (2, 2)


def test_graph_slots_are_recycled():
    """graph slot ids come from a free list and are reused after release (the library's table is finite)."""
    from advancedliteratemachinery_amd import _lib
    from advancedliteratemachinery_amd.model.transformer import _GraphSlots
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libomp355.so not built')
    a, b = _GraphSlots.acquire(), _GraphSlots.acquire()
    assert a >= 0 and b >= 0 and a != b
    _GraphSlots.release(a)
    c = _GraphSlots.acquire()
    assert c == a
    _GraphSlots.release(b)
    _GraphSlots.release(c)
    # exhausting the table yields -1 (eager launches), never an out-of-range id
    got = []
    while True:
        s = _GraphSlots.acquire()
        if s < 0:
            break
        got.append(s)
    assert len(set(got)) == len(got) and max(got) < _lib.MAX_GRAPH_SLOTS
    for s in got:
        _GraphSlots.release(s)
    assert _GraphSlots.acquire() >= 0


def test_omp_ctx_is_per_thread_and_default_is_protected():
    """include/omp355.h context API: a thread works on the process default context until it makes another one current;
    other threads are unaffected; the default context cannot be destroyed."""
    import ctypes
    import threading
    from advancedliteratemachinery_amd import _lib, ops
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libomp355.so not built')
    h = _lib.lib()
    default = ops.current_context_handle()
    assert default
    ctx = ops.Context()
    seen = {}
    with ctx:
        assert ops.current_context_handle() == ctx.handle != default
        t = threading.Thread(target=lambda: seen.setdefault('other', ops.current_context_handle()))
        t.start()
        t.join()
        inner = ops.Context()
        with inner:
            assert ops.current_context_handle() == inner.handle
        assert ops.current_context_handle() == ctx.handle        # restored on exit
        inner.destroy()
    assert seen['other'] == default                                # a new thread starts on the default context
    assert ops.current_context_handle() == default
    assert h.omp_ctx_destroy(ctypes.c_void_p(default)) != 0        # refused
    # a context destroyed on THIS thread while ANOTHER thread still has it current: that thread must fall back to the default
    # context on its next call instead of dereferencing freed memory, and the stale handle must be refused
    go, done = threading.Event(), threading.Event()

    def worker():
        ops.make_context_current(ctx.handle)
        seen['worker_before'] = ops.current_context_handle()
        done.set()
        go.wait(10)
        seen['worker_after'] = ops.current_context_handle()
    t = threading.Thread(target=worker)
    t.start()
    assert done.wait(10)
    stale = ctx.handle
    ctx.destroy()
    go.set()
    t.join()
    assert seen['worker_before'] == stale and seen['worker_after'] == default
    assert h.omp_ctx_make_current(ctypes.c_void_p(stale)) != 0     # a destroyed handle cannot become current
    assert h.omp_ctx_destroy(ctypes.c_void_p(stale)) != 0          # nor be destroyed twice
    assert ops.current_context_handle() == default


def test_capture_gate_is_a_process_wide_mutex():
    """include/omp355.h omp_capture_gate_enter / _leave (ABI 21): the bracket a host thread puts around event calls on a stream another thread
    drives through omp_decoder_run -- the mutex a graph capture holds.  No GPU needed: while one thread is inside the gate, another thread's
    enter blocks, and proceeds on leave; engine/pipeline.py::LaneEvent runs every event call inside it and polls instead of blocking there."""
    import threading
    import time
    from advancedliteratemachinery_amd import _lib, ops
    from advancedliteratemachinery_amd.engine.pipeline import LaneEvent
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libomp355.so not built')
    order = []
    inside, release = threading.Event(), threading.Event()

    def holder():                      # stands for a lane thread in the middle of a capture
        with ops.capture_gate():
            inside.set()
            release.wait(10)
            order.append('capture done')
    t = threading.Thread(target=holder)
    t.start()
    assert inside.wait(10)

    class FakeEvent(object):           # a torch.cuda.Event's interface; what matters is WHEN it is touched
        def __init__(self):
            self.polls = 0

        def query(self):
            order.append('query')
            self.polls += 1
            return self.polls >= 3
    ev = FakeEvent()
    waiter = threading.Thread(target=lambda: LaneEvent(ev).synchronize(poll_s=0.001))
    waiter.start()
    time.sleep(0.05)
    assert order == []                 # the waiter is parked at the gate: the event has not been touched during the "capture"
    release.set()
    t.join(10)
    waiter.join(10)
    assert not waiter.is_alive()
    assert order == ['capture done', 'query', 'query', 'query']    # polled (never blocked inside the gate) until complete
    with ops.capture_gate():           # left unlocked
        pass


def test_cu_mask_words_are_balanced_over_xcds():
    """ops.cu_mask_words: n CUs on EVERY XCD under the mask numbering measured on the MI355X (bit i -> XCD i % 8, profiles/
    r02u_cu_mask_probe.txt) and under an XCD-major numbering; a mask and its complement partition the 256 CUs."""
    from advancedliteratemachinery_amd import ops
    for n in (8, 16, 24):
        w, c = ops.cu_mask_words(n), ops.cu_mask_words(n, complement=True)
        bits = [i for i in range(256) if (w[i // 32] >> (i % 32)) & 1]
        cbits = [i for i in range(256) if (c[i // 32] >> (i % 32)) & 1]
        assert len(bits) == 8 * n and sorted(bits + cbits) == list(range(256))
        for xcd in range(8):
            assert sum(1 for i in bits if i % 8 == xcd) == n            # interleaved numbering (measured)
            assert sum(1 for i in bits if i // 32 == xcd) == n          # XCD-major numbering


def test_make_tiles_partitions_rows_by_image():
    """Decoder.make_tiles: every row belongs to exactly one group, groups never straddle images, a group holds at most
    16 * q_tiles rows, q_tiles follows the largest per-image count (1 / 2 / 4 query tiles)."""
    import random
    from advancedliteratemachinery_amd.model.transformer import Decoder
    rng = random.Random(0)
    for _ in range(200):
        counts = [rng.choice([0, 1, 1, 3, 16, 17, 32, 33, 64]) for _ in range(rng.randint(1, 9))]
        if not any(counts):
            counts[0] = 1
        groups, qt = Decoder.make_tiles(counts)
        mx = max(counts)
        assert qt == (1 if mx <= 16 else (2 if mx <= 32 else 4))
        starts = [sum(counts[:i]) for i in range(len(counts))]
        covered = []
        for (r0, n, img) in groups:
            assert 1 <= n <= 16 * qt
            assert starts[img] <= r0 and r0 + n <= starts[img] + counts[img]      # inside ONE image's rows
            covered += list(range(r0, r0 + n))
        assert covered == list(range(sum(counts)))                                 # each row once, in order


def test_n_split_is_a_power_of_two_that_fills_the_chip_without_oversplitting():
    """Decoder._n_split: power of two <= 16; never more than ~4 workgroups per CU worth of (group, head, split) triples once
    a split exists; the LDS-ring kernel (64 rows per image) splits only while that still leaves >= 8 key blocks per split."""
    import types
    from advancedliteratemachinery_amd.model.transformer import Decoder
    d = types.SimpleNamespace(dtype=torch.bfloat16, nH=8, n_split_override=0)
    for n_img in (1, 2, 8, 40, 80, 128, 160, 256, 512):
        for M in (400, 1024, 4096, 4800):
            tiles1 = ([(r, 1, r) for r in range(n_img)], 1)
            s1 = Decoder._n_split(d, tiles1, M)
            assert s1 in (1, 2, 4, 8, 16)
            assert s1 == 1 or n_img * 8 * s1 <= 1024
            tiles4 = ([(r * 64, 64, r) for r in range(n_img)], 4)
            s4 = Decoder._n_split(d, tiles4, M)
            assert s4 in (1, 2, 4, 8, 16)
            assert s4 == 1 or M >= 8 * 32 * (s4 // 2)
            if n_img * 8 >= 512:
                assert s4 == 1


def test_split_bf16_products_are_one_gemm_over_3k():
    """The algebra of the bf16x3 engine (include/omp355.h omp_gemm_args.a_wrap), on the CPU: x = hi + lo keeps 16
    mantissa bits; [a_hi | a_lo | a_hi] . [w_hi | w_hi | w_lo]^T is hi.hi + lo.hi + hi.lo -- ONE bf16 GEMM with K' = 3K whose
    A side is the [hi | lo] pair row read with a wrap -- and is within 2^-15 of the fp64 product where plain bf16 is 2^-8."""
    from advancedliteratemachinery_amd import ops
    g = torch.Generator().manual_seed(0)
    A, W = torch.randn(64, 256, generator=g), torch.randn(48, 256, generator=g) / 16.0
    W3 = ops.split_weight3(W).float()                     # [N, 3K] = [hi | hi | lo]
    W2 = ops.split_weight2(W).float()                     # [N, 2K] = [hi | lo]
    a_hi = A.to(torch.bfloat16).float()
    a_lo = (A - a_hi).to(torch.bfloat16).float()
    assert float((A - a_hi - a_lo).abs().max() / A.abs().max()) < 2.0 ** -15          # 16 mantissa bits in 2 x 16 bits
    assert torch.equal(W3[:, :256], W2[:, :256]) and torch.equal(W3[:, 256:512], W2[:, :256]) and torch.equal(W3[:, 512:], W2[:, 256:])
    pair = torch.cat([a_hi, a_lo], 1)                     # what the producers store (OMP_BF16X2)
    wrapped = torch.cat([pair, pair[:, :256]], 1)         # what the kernels read: columns >= a_wrap = 2K come from column k - a_wrap
    y3 = wrapped.double() @ W3.double().t()
    ref = A.double() @ W.double().t()
    assert float((y3 - ref).abs().max() / ref.abs().max()) < 2.0 ** -15
    y1 = a_hi.double() @ W2[:, :256].double().t()         # plain bf16 operands
    assert float((y1 - ref).abs().max() / ref.abs().max()) > 2.0 ** -10


def test_swin_block_tile_layout_identities():
    """csrc/swin_block.hip addresses its bf16 tiles ([rows][128] at 256 B per row, 16-byte chunk c of row r at slot c ^ (r & 15))
    through three shortcuts; each is an identity over the index ranges the kernel uses."""
    def sw_off(row, chunk):
        return row * 256 + ((chunk ^ (row & 15)) << 4)

    # (1) fragment reads: rows R + li with R a multiple of 16 -> R * 256 + cx[ks], cx[ks] = li * 256 + (((ks * 4 + g) ^ li) << 4);
    #     the 16 lanes of one k-group hit 16 different chunk slots (conflict-free ds_read_b128)
    for R in (0, 16, 48, 128, 256 + 96):
        for ks in range(4):
            for g in range(4):
                slots = set()
                for li in range(16):
                    cx = li * 256 + (((ks * 4 + g) ^ li) << 4)
                    assert R * 256 + cx == sw_off(R + li, ks * 4 + g)
                    slots.add((cx >> 4) & 15)
                assert len(slots) == 16
    # (2) LayerNorm writes: wave `head`, lane (tk, lj), pass `it`, piece k -> token row it * 32 + head * 8 + tk, fp32 piece c = k * 8 + lj,
    #     i.e. bf16 chunk c >> 1 = k * 4 + (lj >> 1), half (lj & 1); the kernel keeps ONE offset and folds k in as ^ (k << 6)
    for head in range(4):
        for tk in range(8):
            for lj in range(8):
                ln_row = head * 8 + tk
                ln_off = ln_row * 256 + (((lj >> 1) ^ (ln_row & 15)) << 4) + (lj & 1) * 8
                for it in range(2):
                    for k in range(4):
                        row, c = it * 32 + ln_row, k * 8 + lj
                        assert it * 32 * 256 + (ln_off ^ (k << 6)) == sw_off(row, c >> 1) + (c & 1) * 8
    # every (row, 8-byte half chunk) of the 64 x 128 tile is written exactly once per window
    seen = set()
    for head in range(4):
        for lane in range(64):
            tk, lj = lane >> 3, lane & 7
            for it in range(2):
                for k in range(4):
                    seen.add((it * 32 + head * 8 + tk, k * 8 + lj))
    assert len(seen) == 64 * 32
    # (3) O^T accumulator pieces: head h, dim tile dt, lane (li, g) holds dims dt * 16 + 4 g .. + 4 of query t4 * 16 + li
    for h in range(4):
        for li in range(16):
            for g in range(4):
                off = [li * 256 + (((h * 4 + dt * 2 + (g >> 1)) ^ li) << 4) + (g & 1) * 8 for dt in range(2)]
                for t4 in range(4):
                    for dt in range(2):
                        q, ch = t4 * 16 + li, h * 32 + dt * 16 + 4 * g      # channel of the first of the 4 values
                        assert t4 * 4096 + off[dt] == sw_off(q, ch >> 3) + ((ch >> 2) & 1) * 8
    # (4) window-local token t -> (t // 7, t % 7) as (t * 37) >> 8 for every t the kernel decodes
    for t in range(64):
        assert (t * 37) >> 8 == t // 7


def test_swin_block_accumulators_are_operands():
    """The fused block feeds MFMA accumulators straight back as operands.  A 16x16x32 operand lane (li, g) holds k = 8 g + e, e < 8;
    an accumulator lane (li, g) holds rows 4 g + r, r < 4, of column li.  Packing {acc[tile 0][0..3], acc[tile 1][0..3]} therefore
    presents row (e >> 2) * 16 + 4 g + (e & 3) at hardware index 8 g + e: a product over k is unchanged as long as BOTH operands
    use the same map and the map is a bijection of the 32 indices."""
    import numpy as np
    perm = np.array([[(e >> 2) * 16 + 4 * g + (e & 3) for e in range(8)] for g in range(4)]).reshape(-1)   # hardware k -> head dim
    assert sorted(perm.tolist()) == list(range(32))
    rng = np.random.default_rng(0)
    q, k = rng.standard_normal((64, 32)), rng.standard_normal((64, 32))
    # S^T = K Q^T with both operands permuted along k == the unpermuted product
    assert np.allclose(k[:, perm] @ q[:, perm].T, k @ q.T)
    # O^T = V^T P per 32-key step ps: P comes from the S^T accumulators (keys kt * 16 + 4 g + r of tile kt), V from the v accumulators
    # (tokens tt * 16 + 4 g + r of tile tt): the same map, offset by the step
    v, p = rng.standard_normal((64, 32)), rng.random((64, 64))
    o = np.zeros((32, 64))
    for ps in range(2):
        keys = 32 * ps + perm
        assert sorted(keys.tolist()) == list(range(32 * ps, 32 * ps + 32))
        o += v[keys].T @ p[:, keys].T          # [dim][query]
    assert np.allclose(o, (p @ v).T)


def test_swin_attn_block_wrapper_refuses_bad_operands():
    """ops.swin_attn_block validates layouts before the C ABI sees a pointer (no device needed: nothing reaches the library)."""
    import pytest
    import torch
    from advancedliteratemachinery_amd import ops
    x = torch.zeros(2 * 7 * 7, 128)
    w, pw, be = torch.zeros(384, 128, dtype=torch.bfloat16), torch.zeros(128, 128, dtype=torch.bfloat16), torch.zeros(4, 64, 64)
    call = lambda **kw: ops.swin_attn_block(kw.get('x', x), None, None, kw.get('w', w), None, kw.get('be', be), kw.get('pw', pw), None,  # noqa: E731
                                            2, 7, 7, 128, 4, 0, out=kw.get('out'))
    with pytest.raises(ValueError):
        call(w=w[:, :64])                       # a strided view
    with pytest.raises(ValueError):
        call(x=torch.zeros(10, 128))            # not B*H*W rows
    with pytest.raises(ValueError):
        call(be=torch.zeros(8, 64, 64))         # bias expanded for another head count
    with pytest.raises(ValueError):
        call(out=torch.zeros(3, 128))
    with pytest.raises(TypeError):
        call(w=w.float())                       # the kernel takes bf16 weights

def test_pack_attn_block_fragment_order():
    """model/packing.py::pack_attn_block: every 64-lane x 8-element fragment is the slice of the weight the kernel's indexing names."""
    import torch
    from advancedliteratemachinery_amd.model.packing import pack_attn_block
    C, nH = 256, 8
    KS = C // 32
    qkv = torch.arange(3 * C * C, dtype=torch.float32).reshape(3 * C, C).to(torch.bfloat16)
    qkv = (torch.randn(3 * C, C, generator=torch.Generator().manual_seed(0))).to(torch.bfloat16)
    proj = (torch.randn(C, C, generator=torch.Generator().manual_seed(1))).to(torch.bfloat16)
    img = pack_attn_block(qkv, proj, nH)
    assert img.numel() == 4 * C * C
    for (h, ks, sel, dt) in ((0, 0, 0, 0), (3, 5, 1, 1), (7, 7, 2, 0), (2, 1, 2, 1)):
        frag = img[((h * KS + ks) * 6 + sel * 2 + dt) * 512:][:512].reshape(64, 8)
        for lane in (0, 5, 17, 40, 63):
            g, li = lane >> 4, lane & 15
            assert torch.equal(frag[lane], qkv[sel * C + 32 * h + 16 * dt + li, 32 * ks + 8 * g:32 * ks + 8 * g + 8])
    base = 3 * C * C
    for (w, nt, ks) in ((0, 0, 0), (5, 1, 3), (7, 0, 7)):
        frag = img[base + ((w * 2 + nt) * KS + ks) * 512:][:512].reshape(64, 8)
        for lane in (0, 9, 33, 63):
            g, li = lane >> 4, lane & 15
            assert torch.equal(frag[lane], proj[32 * w + 16 * nt + li, 32 * ks + 8 * g:32 * ks + 8 * g + 8])


def test_swin_block256_tile_layout_identities():
    """The C = 256 variant (rows of 512 B = 32 chunks, slot = chunk ^ (row & 15)) folds the k-step / piece index into its offsets by XOR."""
    def sw_off(row, chunk):
        return row * 512 + ((chunk ^ (row & 15)) << 4)

    for li in range(16):
        for g in range(4):
            cx0 = li * 512 + ((g ^ li) << 4)
            for ks in range(8):
                for R in (0, 16, 32, 48):
                    assert R * 512 + (cx0 ^ (ks << 6)) == sw_off(R + li, ks * 4 + g)
    seen = set()
    for head in range(8):
        for lane in range(64):
            tk, lj = lane >> 4, lane & 15
            ln_row = head * 4 + tk
            ln_off = ln_row * 512 + (((lj >> 1) ^ (ln_row & 15)) << 4) + (lj & 1) * 8
            for it in range(2):
                for k in range(4):
                    row, c = it * 32 + ln_row, k * 16 + lj            # fp32 piece c of the 1 KB row -> bf16 chunk c >> 1, half c & 1
                    assert it * 32 * 512 + (ln_off ^ (k << 7)) == sw_off(row, c >> 1) + (c & 1) * 8
                    seen.add((row, c))
    assert len(seen) == 64 * 64
    for h in range(8):
        for li in range(16):
            for g in range(4):
                off = [li * 512 + (((h * 4 + dt * 2 + (g >> 1)) ^ li) << 4) + (g & 1) * 8 for dt in range(2)]
                for t4 in range(4):
                    for dt in range(2):
                        q, ch = t4 * 16 + li, h * 32 + dt * 16 + 4 * g
                        assert t4 * 8192 + off[dt] == sw_off(q, ch >> 3) + ((ch >> 2) & 1) * 8


def test_split_plane_slab_indexing_is_a_bijection():
    """Round 4, the parity engine's K / V^T slabs: every 32-key block of an (image, head) slab is
    [hi plane | lo plane]; the element of key k, dim d, plane p sits at ((k // 32) * 2 + p) * 2048 + (k % 32) * 64 + d (K) and at
    ((k // 32) * 2 + p) * 2048 + d * 32 + slot(k % 32) (V^T, slot = the bf16 slab's matrix-core order) -- the formulas the GEMM
    epilogues (csrc/gemm.hip::store4, gemm256.inc, gemm4w.inc) and the cross-attention kernels (csrc/decoder.hip, bf16s_t) share.
    Both maps are bijections onto the slab and a value survives hi + lo to 2^-16 relative."""
    import numpy as np
    Mpad = 96
    k, d, p = np.meshgrid(np.arange(Mpad), np.arange(64), np.arange(2), indexing='ij')
    ik = ((k // 32) * 2 + p) * 2048 + (k % 32) * 64 + d
    kl = k % 32
    slot = ((kl & 15) >> 2) * 8 + (kl >> 4) * 4 + (kl & 3)
    iv = ((k // 32) * 2 + p) * 2048 + d * 32 + slot
    for idx in (ik, iv):
        assert sorted(idx.reshape(-1).tolist()) == list(range(Mpad * 128))
    import torch
    x = torch.randn(4096) * 3
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    rel = ((hi.float() + lo.float() - x).abs() / x.abs().clamp_min(1e-30)).max().item()
    assert rel <= 2.0 ** -16


def test_gemm_dispatch_table():
    """The kernel selector of omp_gemm_bias_act is host logic (csrc/gemm.hip: launch_gemm): omp_debug_gemm_choice returns it without a
    launch or a device.  Pins the table DESIGN.md sections 5 and 10 describe -- which product of which engine runs on which kernel --
    so that a change of a rule shows up here and not as an unexplained move of a bench leg."""
    from advancedliteratemachinery_amd import _lib, ops
    G = ops.gemm_choice
    F32, BF, SPLIT, GELU = ops.OMP_F32, ops.OMP_BF16, ops.OMP_BF16X2, ops.ACT_GELU
    # bf16 engine, 32-image chunk of 1024x1024 images
    assert G(131072, 1536, 512) == 9                                        # stage-2 qkv: gemm_256
    assert G(131072, 2048, 512, act=GELU) == 20                             # stage-2 fc1 + GELU: persistent four-wave kernel
    assert G(32768, 4096, 1024, act=GELU) == 20                             # stage-3 fc1
    assert G(131072, 512, 512, out_dtype=F32, residual=True) == 9           # proj into the fp32 residual stream
    assert G(131072, 512, 2048, out_dtype=F32, residual=True) == 9          # fc2
    assert G(524288, 768, 256) == 10                                        # half-million-row stage-1 product without GELU: gemm_4w
    assert G(131584, 2304, 768) == 20 and G(131584, 3072, 768, act=GELU) == 20   # MGP-STR's ViT-B qkv / fc1 (K = 768)
    assert G(131000, 1536, 512) == 9                                        # ragged M: not the persistent kernel
    assert G(655360, 6144, 512, store_mode=_lib.STORE_KBLK, kv=(160, 4096, 4096, 8, 32)) == 9   # K slabs of 160 images
    # parity engine (bf16x3 operands): the fused three-product kernel wherever a 256x256-tile kernel would run on an even shape
    assert G(131072, 1536, 1536, out_dtype=F32, a_wrap=1024) == 22
    assert G(131072, 2048, 1536, out_dtype=SPLIT, act=GELU, a_wrap=1024) == 22
    assert G(524288, 768, 768, out_dtype=F32, a_wrap=512) == 22
    assert G(32768, 1024, 12288, out_dtype=F32, residual=True, a_wrap=8192) == 22
    assert G(2097152, 512, 384, out_dtype=SPLIT, act=GELU, a_wrap=256) == 9    # stage 0: K0 = 128 is below the fused kernel's granularity
    # decoder phases
    assert G(10240, 512, 512, out_dtype=F32, residual=True) == 5            # 160 images x 64 rows: 320 tiles of 128x128
    assert G(10240, 1536, 512) == 9                                         # ... its qkv fills the chip with 256x256 tiles (240 of them)
    assert G(160, 1536, 512) == 6                                           # point decoder of a 160-image call: 64x64 ring
    assert G(8, 512, 512, small_m=True) == 4                                # few rows: split-K over the waves
    assert G(32, 1536, 512, ln=True, small_m=True, lda=512) == 4            # ... with the LayerNorm prologue
    assert G(300, 384, 128) == 6
    # fp32 engine: no 256x256 kernels
    assert G(131072, 1536, 512, dtype=F32) == 5
    # argument errors surface as errors, not as a selector
    with pytest.raises(RuntimeError):
        G(128, 512, 100)


def test_gemm_choice_refuses_what_the_launch_path_refuses():
    """omp_debug_gemm_choice used to return BEFORE the selector / argument compatibility checks of launch_gemm, so the host-logic
    test of the dispatch table could report a selector for arguments the real call rejects.  A forced selector that does not take the
    product is now OMP_ERR_UNSUPPORTED in the host-logic call as well."""
    import pytest
    from advancedliteratemachinery_amd import ops
    try:
        ops.force_gemm_kernel(3)      # the row-streaming kernel has no second destination
        with pytest.raises(RuntimeError):
            ops.gemm_choice(131072, 512, 1024, out_dtype=ops.OMP_BF16, residual=True, two_destinations=True)
        assert ops.gemm_choice(16, 512, 512) == 3
        ops.force_gemm_kernel(20)     # the persistent four-wave kernel needs M, N, K multiples of 256
        with pytest.raises(RuntimeError):
            ops.gemm_choice(10240, 1104, 512)
        assert ops.gemm_choice(131072, 2048, 512, act=ops.ACT_GELU) == 20
        ops.force_gemm_kernel(22)     # the fused bf16x3 kernel takes bf16x3 operands only
        with pytest.raises(RuntimeError):
            ops.gemm_choice(131072, 1536, 512)
    finally:
        ops.force_gemm_kernel(0)


def test_env_knobs_are_validated(monkeypatch):
    """OMP355_* knobs were read with bare int() / string compares"""
    import pytest
    from advancedliteratemachinery_amd.utils.env import env_flag, env_int
    monkeypatch.setenv('OMP355_ENC_CHUNK', '64')
    assert env_int('OMP355_ENC_CHUNK', 32, 1, 4096) == 64
    for bad in ('0', 'x', '99999'):
        monkeypatch.setenv('OMP355_ENC_CHUNK', bad)
        with pytest.raises(ValueError):
            env_int('OMP355_ENC_CHUNK', 32, 1, 4096)
    monkeypatch.setenv('OMP355_CROSS_SPLIT', '3')
    with pytest.raises(ValueError):
        env_int('OMP355_CROSS_SPLIT', 0, 0, 16, allowed=(0, 1, 2, 4, 8, 16))
    monkeypatch.setenv('OMP355_KV_SPLIT', 'yes')
    with pytest.raises(ValueError):
        env_flag('OMP355_KV_SPLIT', True)
    monkeypatch.delenv('OMP355_KV_SPLIT')
    assert env_flag('OMP355_KV_SPLIT', True) is True


def test_pack_kv_rows_streams_name_the_rows_the_kernel_expects():
    """model/packing.py::pack_kv_rows_k / _v (csrc/kv_rows.hip): wave w's stream holds, per slab, 16 k-steps x 4 feature tiles of 1 KB fragments
    [64 lanes][8]; lane (li, g) of tile ft carries 8 consecutive k of ONE weight row -- for V^T the row of dim 16 ft + li of head w, for K the row
    of dim 32 (ft / 2) + 8 (li / 4) + 4 (ft % 2) + li % 4 (so that the lane owning matrix-core rows 4 g .. 4 g + 3 stores dims 8 g .. 8 g + 7 and 32 + 8 g ..)."""
    import torch
    from advancedliteratemachinery_amd.model import packing
    n_slabs = 2
    rows = torch.arange(n_slabs * 512, dtype=torch.float32)[:, None]
    cols = torch.arange(512, dtype=torch.float32)[None, :]
    wr = (rows + 0 * cols).to(torch.bfloat16)                     # bf16 cannot carry row and column in one value: two matrices
    wc = (0 * rows + cols).to(torch.bfloat16)
    for name, perm in (('k', True), ('v', False)):
        pack = packing.pack_kv_rows_k if perm else packing.pack_kv_rows_v
        (br, stride), (bc, _) = pack(wr), pack(wc)
        assert stride == n_slabs * 64 * 1024
        fr = br[:8 * stride].view(torch.bfloat16).reshape(8, n_slabs, 16, 4, 64, 8).float()   # [wave][slab][k-step][tile][lane][8]
        fc = bc[:8 * stride].view(torch.bfloat16).reshape(8, n_slabs, 16, 4, 64, 8).float()
        for (wv, p, ks, ft, lane) in ((0, 0, 0, 0, 0), (2, 1, 5, 3, 37), (7, 1, 15, 1, 63), (4, 0, 9, 2, 18)):
            li, g = lane & 15, lane >> 4
            dim = (ft // 2) * 32 + (li // 4) * 8 + (ft % 2) * 4 + li % 4 if perm else ft * 16 + li
            assert fr[wv, p, ks, ft, lane].tolist() == [float(torch.tensor(p * 512 + wv * 64 + dim, dtype=torch.float32).to(torch.bfloat16))] * 8, name
            assert fc[wv, p, ks, ft, lane].tolist() == [float(torch.tensor(ks * 32 + g * 8 + e, dtype=torch.float32).to(torch.bfloat16)) for e in range(8)], name


def test_rows_tile_rule():
    """csrc/dec_rows.hip rows_rtt as host logic (omp_debug_rows_tile_choice): a decoder chain launch takes the smallest tile of 32 / 48 / 64 / 80
    rows (the mid chain: from 16) that keeps it on half the chip (128 workgroups) -- the rule DESIGN.md section 11 derives from the two-stream
    measurement -- and a forced tile (omp_debug_rows_tile) overrides it."""
    from advancedliteratemachinery_amd import ops
    T = ops.rows_tile_choice
    assert T(10240) == 80            # 160 images x 64 instances: 128 workgroups
    assert T(5120) == 48             # 107 workgroups (32 rows would be 160)
    assert T(4096) == 32             # exactly 128
    assert T(4097) == 48
    assert T(200) == 32
    assert T(1 << 20) == 80          # beyond the chip: the largest tile
    assert T(160, mid=True) == 16    # the point decoder of a 160-image call: 10 workgroups
    assert T(2048, mid=True) == 16 and T(2049, mid=True) == 32
    assert T(10240, mid=True) == 80
    try:
        ops.rows_tile(3)
        assert T(10240) == 48 and T(160, mid=True) == 48
    finally:
        ops.rows_tile(0)
    assert T(10240) == 80


def test_qkv_tail_rows_match_the_16_byte_store_mapping():
    """model/packing.py::_qkv_tail_rows (bf16 chains' q | k | v tail) against csrc/dec_rows.hip::store_bias_perm: the accumulator quad of feature
    tile ft that lane g holds (matrix-core rows 4 g + r) is stored at feature offset 32 (ft / 2) + 8 g + 4 (ft % 2) + r of the wave's 64 -- so the
    weight row packed at stream position (tile ft, row 4 g + r) must be exactly that feature; fp32 masters (parity engine) keep their order."""
    import torch
    from advancedliteratemachinery_amd.model import packing
    w = torch.arange(1536, dtype=torch.float32)[:, None].repeat(1, 512)
    assert torch.equal(packing._qkv_tail_rows(w), w)                       # fp32: untouched
    p = packing._qkv_tail_rows(w.to(torch.bfloat16)).float()
    for group in (0, 5, 23):                                               # a wave's 64 features of some pass
        for ft in range(4):
            for g in range(4):
                for r in range(4):
                    stored_at = 32 * (ft // 2) + 8 * g + 4 * (ft % 2) + r
                    assert p[group * 64 + ft * 16 + 4 * g + r, 0].item() == float(torch.tensor(group * 64 + stored_at, dtype=torch.float32).to(torch.bfloat16))


def test_build_plan_is_by_content_and_audited_objects_need_proof():
    """advancedliteratemachinery_amd/build.py: an unchanged tree compiles nothing; an AUDITED source (asm-addressed kernels) whose object has
    neither an audit stamp of the current digest nor its device assembly on disk is STALE -- the audit is never skipped silently."""
    import shutil
    from advancedliteratemachinery_amd import build as B
    B.build(verbose=False)                       # the driver has built the tree already: verifies, links nothing new
    jobs, objs, stamps, audited = B.plan()
    assert jobs == [] and len(objs) == len(B.SOURCES) and set(audited) == set(B.AUDITED)
    obj = os.path.join(B.OBJ, 'kv_rows.o')
    asm, ast = B._asm_of('kv_rows.hip'), obj + '.audit.stamp'
    assert os.path.exists(ast)
    moved = []
    try:
        for f in (asm, ast):
            if os.path.exists(f):
                shutil.move(f, f + '.away')
                moved.append(f)
        stale = [os.path.basename(j[1]) for j in B.plan()[0]]
        assert stale == ['kv_rows.o']            # no proof of an audit left: recompile (and re-audit), do not skip
        if asm in moved:                         # the assembly alone is proof enough to run the audit again without recompiling
            shutil.move(asm + '.away', asm)
            moved.remove(asm)
            assert B.plan()[0] == []
    finally:
        for f in moved:
            shutil.move(f + '.away', f)
    assert B.plan()[0] == []
    # a changed flag set changes every digest
    keep = list(B.FLAGS)
    try:
        B.FLAGS.append('-DOMP355_TEST_FLAG')
        assert len(B.plan()[0]) == len(B.SOURCES)
    finally:
        B.FLAGS[:] = keep
