"""Batch pipelining over HIP streams: several image batches in flight on one GPU.

One batch through the hot path is a chain of very different phases -- the Swin encoder (matrix-core /
HBM bound, fills the chip), the point decoder (~130 strictly sequential steps of small kernels: latency
bound, a fraction of the CUs busy), the polygon / recognition decoders (medium kernels).  Run back to back
they leave most of the MI355X idle most of the time.  A `LanePool` keeps `n_lanes` batches in flight, each
lane on its own HIP stream (+ two side streams for polygon || recognition) with its own decoder state
(K/V slabs, KV caches, captured hipGraphs -- `Decoder.fork()`), so the latency-bound phases of one batch run
under the throughput-bound phases of the others.  Weights are shared; nothing is replicated but buffers.

Each lane is driven by its own host thread (the per-batch host work is ctypes / HIP calls that release the
GIL, and the EOS polls of the point decoder only block their own lane).  The reference has no counterpart:
engine/val.py:19-35 processes one image at a time, synchronously.
"""
import os
import queue
import threading
import time
from concurrent.futures import Future

import torch

from ..utils.env import env_flag


class Lane(object):
    def __init__(self, device, index, dec_priority=False, side_streams=True):
        self.index = index
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        if not side_streams:   # one HIP stream per lane: polygon then recognition on the lane stream (see LanePool)
            self.side, self.dec_stream, self._dec, self._dec_src = None, None, None, None
            return
        # decoder phases are chains of small latency-bound kernels: with dec_priority they run on high-priority
        # streams, so the hardware dispatcher places their workgroups ahead of the thousands queued by another
        # lane's encoder GEMMs whenever CUs free up
        pr = -1 if dec_priority else 0
        self.side = (torch.cuda.Stream(device=device, priority=pr), torch.cuda.Stream(device=device, priority=pr))
        self.dec_stream = torch.cuda.Stream(device=device, priority=pr) if dec_priority else None
        self._dec = None
        self._dec_src = None

    def decoder(self, base):
        """This lane's private fork of the model's decoder (re-forked when the engine was rebuilt)."""
        if self._dec is None or self._dec_src is not base:
            self._dec, self._dec_src = base.fork(), base
        self._dec.use_graph = base.use_graph
        return self._dec


class LaneEvent(object):
    """Completion event of a lane job, safe to wait on from ANOTHER host thread.  The event is recorded on the lane's stream, and the lane's worker
    may already be capturing the decoder graphs of its NEXT job on that stream: while a stream is capturing, HIP refuses hipEventSynchronize /
    hipEventQuery / hipStreamWaitEvent for every event last recorded in it -- earlier, uncaptured work included -- and invalidates the capture
    (hipErrorCapturedEvent / hipErrorStreamCaptureIsolation; observed once in nine full GPU suites, tests/test_gpu_e2e.py::
    test_pipelined_lanes_match_direct).  Every call below therefore runs inside the library's capture gate (ops.capture_gate: the mutex a
    capture holds from begin to instantiate), and only NON-BLOCKING calls do: synchronize() polls.  Graphs are captured once per slot, so in
    steady state the gate is never contended.
    (Measured and not kept: capturing on a private stream and launching the graph on the lane's -- graphs captured off their launch stream
    serialise against each other, polygon || recognition 100 -> 120 ms per 160 images, profiles/r06ze_*, r06zf_*; retrying the refused call
    -- the refusal has already invalidated the lane's capture, profiles/r06zg_*.)"""
    __slots__ = ('event',)

    def __init__(self, event):
        self.event = event

    def query(self):
        from .. import ops
        with ops.capture_gate():
            return self.event.query()

    def synchronize(self, poll_s=0.0001):
        while not self.query():
            time.sleep(poll_s)

    def wait(self, stream=None):
        """make `stream` (default: the current stream) wait for the lane job"""
        from .. import ops
        st = stream if stream is not None else torch.cuda.current_stream()
        with ops.capture_gate():
            st.wait_event(self.event)


class LanePool(object):
    """submit(fn) runs fn(lane) on the next lane (round robin) inside that lane's stream context and
    returns a Future of (result, LaneEvent); the event is recorded on the lane stream after fn's last launch."""

    def __init__(self, device, n_lanes, dec_priority=None, side_streams=True):
        """side_streams=False: every lane is ONE HIP stream.  The runtime multiplexes streams onto a few hardware queues
        (GPU_MAX_HW_QUEUES, default 4) and streams that share a queue serialise: with three streams per lane which lanes collide
        is decided by creation order, and the throughput of small-call pipelines swings 95-208 img/s between pools of the same
        size (profiles/r03j_lane_sweep_*).  One stream per lane keeps up to four lanes on queues of their own."""
        self.device = torch.device(device)
        if self.device.index is None:   # torch.cuda.set_device() in the lane threads needs an explicit index
            self.device = torch.device(self.device.type, torch.cuda.current_device())
        if dec_priority is None:
            dec_priority = env_flag('OMP355_DEC_PRIORITY', False)
        from .. import ops
        self._ctx = ops.current_context_handle()   # lane threads work on the omp_ctx of the thread that built the pool
        self.lanes = [Lane(self.device, i, dec_priority, side_streams) for i in range(max(1, n_lanes))]
        self._queues = [queue.Queue() for _ in self.lanes]
        self._next = 0
        self._threads = []
        for lane, q in zip(self.lanes, self._queues):
            t = threading.Thread(target=self._worker, args=(lane, q), daemon=True, name='omp355-lane%d' % lane.index)
            t.start()
            self._threads.append(t)

    def _worker(self, lane, q):
        boot_error = None
        try:
            torch.cuda.set_device(self.device)
            from .. import ops
            ops.make_context_current(self._ctx)
        except BaseException as e:  # noqa: BLE001 -- a dead lane must fail its jobs, not leave them pending forever
            boot_error = e
        while True:
            item = q.get()
            if item is None:
                return
            fn, fut = item
            if not fut.set_running_or_notify_cancel():
                continue
            if boot_error is not None:
                fut.set_exception(boot_error)
                continue
            try:
                with torch.cuda.stream(lane.stream):
                    res = fn(lane)
                    ev = torch.cuda.Event()
                    ev.record(lane.stream)
                fut.set_result((res, LaneEvent(ev)))
            except BaseException as e:  # noqa: BLE001 -- delivered to the submitter
                fut.set_exception(e)

    def submit(self, fn):
        fut = Future()
        i = self._next
        self._next = (i + 1) % len(self.lanes)
        self._queues[i].put((fn, fut))
        return fut

    def infer(self, model, img, mask, sequence, **kw):
        """model.infer(...) on the next lane; Future of (results, event)."""
        return self.submit(lambda lane: model.infer(img, mask, sequence, lane=lane, **kw))

    def synchronize(self):
        """Wait for everything the lanes have been handed so far.  Call it with no job in flight on the host side (every Future resolved): a lane
        that is still capturing would refuse the stream synchronisation (see LaneEvent)."""
        for lane in self.lanes:
            lane.stream.synchronize()
            for s in (lane.side or ()) + ((lane.dec_stream,) if lane.dec_stream is not None else ()):
                s.synchronize()

    def close(self):
        for q in self._queues:
            q.put(None)
        for t in self._threads:
            t.join(timeout=5)
