"""Weight gradients off the critical path: a second HIP stream for the backward pass.

In the backward of a convolution the gradient of the input feeds the next layer down the chain,
the gradient of the WEIGHT feeds nothing until the optimizer (or the gradient all-reduce) runs
after the whole backward.  The reference leaves both on one stream (autograd's default); here the
weight-gradient launches of the sparse convs (kernels.py) and of the dense projection network
(models/ponder/unet3d.py) go to a per-device side stream:

    main stream   ... BN backward -> grad-input(l) -> BN backward -> grad-input(l-1) -> ...
    side stream          wgrad(l) ------------> wgrad(l-1) ---------------> ...

Most kernels of the path do not fill 256 CUs on their own (a few hundred workgroups, long tails,
atomics-bound scatter phases), so the two chains overlap instead of adding up.  One event makes
the side stream wait for the producer of its operands (``fork``); ONE join at the end of the
backward pass (an autograd-engine final callback) makes the caller's stream wait for the side
stream, so everything that runs after ``loss.backward()`` returns - GradScaler, gradient clipping,
FlatGradSync, the optimizer - sees finished gradients.

What must NOT be combined with it: consumers that read a gradient WHILE the backward is still
running, i.e. ``DistributedDataParallel`` (its per-parameter hooks copy gradients into buckets as
autograd produces them).  ``ponder.engines.defaults.create_ddp_model`` and ``bench.py --grad-sync
ddp`` therefore call ``disable()``; the flat gradient sync (utils/grad_sync.py) reduces after the
backward and keeps it on.  Gradients that autograd would ACCUMULATE on the spot (``param.grad``
already set: gradient accumulation over micro-batches) are computed on the main stream as before
(``safe_leaf``).  ``PV2_WGRAD_STREAM=0`` switches the side stream off altogether.
"""
import os

import torch

# The fork / join below lean on three private torch entry points (the current graph task id, the
# raw stream switch, the engine's final-callback queue).  They exist in the torch 2.x line this
# image ships (2.10); where any of them is missing the side stream switches itself off - weight
# gradients then simply stay on the main stream - instead of failing in the middle of a backward.
_PRIVATE_API_OK = (hasattr(torch._C, "_current_graph_task_id") and hasattr(torch._C, "_cuda_setStream")
                   and hasattr(getattr(torch.autograd.Variable, "_execution_engine", None),
                               "queue_callback"))
ENABLED = os.environ.get("PV2_WGRAD_STREAM", "1") != "0" and _PRIVATE_API_OK
_DISABLED_BECAUSE = None if _PRIVATE_API_OK else "this torch build lacks the private stream / graph-task hooks"
_STREAMS = {}
_JOIN_QUEUED = {}   # device index -> id of the graph task whose final callback joins the stream


def disable(reason="disabled by the caller"):
    """Keep every weight gradient on the main stream from now on (e.g. under DDP)."""
    global ENABLED, _DISABLED_BECAUSE
    ENABLED, _DISABLED_BECAUSE = False, reason


def enable():
    global ENABLED, _DISABLED_BECAUSE
    ENABLED = os.environ.get("PV2_WGRAD_STREAM", "1") != "0" and _PRIVATE_API_OK
    _DISABLED_BECAUSE = None if _PRIVATE_API_OK else "this torch build lacks the private stream / graph-task hooks"


def status():
    return "on" if ENABLED else f"off ({_DISABLED_BECAUSE or 'PV2_WGRAD_STREAM=0'})"


def stream(device):
    s = _STREAMS.get(device.index)
    if s is None:
        # PV2_WGRAD_PRIORITY: 0 = the main stream's priority (default), 1 = lower (the hardware
        # queues then prefer the critical chain whenever both have workgroups ready)
        prio = int(os.environ.get("PV2_WGRAD_PRIORITY", "0"))
        s = _STREAMS[device.index] = torch.cuda.Stream(device=device, priority=prio)
    return s


def _graph_task_id():
    fn = getattr(torch._C, "_current_graph_task_id", None)
    return fn() if fn is not None else -1


def active(t):
    """True inside an autograd backward pass on a device tensor while the side stream is on."""
    return ENABLED and t.is_cuda and _graph_task_id() != -1


_TASK_LEAVES = {}   # device index -> (graph task id, ids of the leaves already forked in it)


def _drop_stale_state(device, task):
    """A backward pass that raised never ran its final callback: its join is still pending and the
    side stream's operands are still held.  Join now, then let go."""
    queued = _JOIN_QUEUED.get(device.index)
    if queued is not None and queued != task:
        torch.cuda.current_stream(device).wait_stream(stream(device))
        _JOIN_QUEUED.pop(device.index, None)
        _KEEP.pop(device.index, None)


def safe_leaf(param_view):
    """The gradient handed to autograd for this tensor will only be STORED as ``.grad`` of a leaf -
    no kernel of the main stream touches it before the join:
      * the tensor is a parameter or a plain view of one (a copy, e.g. ``permute().contiguous()``,
        sends its gradient through further autograd kernels);
      * that parameter has no ``.grad`` yet to be added to, and no tensor hooks / post-accumulate
        hooks that would read the gradient when it arrives;
      * it is the FIRST gradient this backward pass produces for that leaf.  A parameter that feeds
        two nodes (tied weights, a module called twice) gets its two gradients summed by the
        engine on the main stream the moment the second one exists: the second one is computed on
        the main stream, after the main stream has waited for the side stream's first."""
    base = param_view._base if param_view._base is not None else param_view
    if not (base.is_leaf and base.grad is None):
        return False
    if getattr(base, "_backward_hooks", None) or getattr(base, "_post_accumulate_grad_hooks", None):
        return False
    if not base.is_cuda:
        return True
    device, task = base.device, _graph_task_id()
    rec = _TASK_LEAVES.get(device.index)
    if rec is None or rec[0] != task:
        _drop_stale_state(device, task)
        rec = _TASK_LEAVES[device.index] = (task, set())
    if id(base) in rec[1]:
        torch.cuda.current_stream(device).wait_stream(stream(device))
        return False
    rec[1].add(id(base))
    return True


_KEEP = {}          # device index -> tensors the side stream reads, held until the join


def _queue_join(device, side):
    task = _graph_task_id()
    if _JOIN_QUEUED.get(device.index) == task:
        return
    _JOIN_QUEUED[device.index] = task

    def join():
        # final callbacks run under the stream that surrounded the call to backward()
        _JOIN_QUEUED.pop(device.index, None)
        torch.cuda.current_stream(device).wait_stream(side)
        _KEEP.pop(device.index, None)

    torch.autograd.Variable._execution_engine.queue_callback(join)


def native_fork(device, reads):
    """For entry points that launch on the side stream THEMSELVES (pv2_convbn_backward records the
    fork event and makes the side stream wait inside the C call): keep ``reads`` - everything the
    side stream touches - alive until the join, queue the join, and return the side stream."""
    side = stream(device)
    _KEEP.setdefault(device.index, []).append(reads)
    _queue_join(device, side)
    return side


_EVENTS = {}        # device index -> (ring of reusable events, next slot)
_RING = 64


def _fork_event(device):
    ring = _EVENTS.get(device.index)
    if ring is None:
        ring = _EVENTS[device.index] = [[torch.cuda.Event() for _ in range(_RING)], 0]
    ev = ring[0][ring[1]]
    ring[1] = (ring[1] + 1) % _RING
    return ev


def fork(fn, reads):
    """Run ``fn()`` (which launches kernels and returns a tensor or a tuple of tensors) on the
    side stream, after everything the current stream has queued so far; ``reads``: the tensors it
    reads that the current stream produced or may free.  The caller's stream joins at the end of
    the running backward pass.

    This runs ~70 times per training step on a host-bound path, hence the plain calls: a reused
    event (a wait binds to the record that precedes it, so re-recording an event later is safe),
    the raw stream switch instead of ``torch.cuda.stream``, and NO ``Tensor.record_stream`` (each
    recorded block costs the allocator an event at free time and a query per later allocation -
    measured: ~5 ms of host time per step for ~200 blocks).  Instead the operands are simply kept
    alive until the join: memory released after it is reused by work queued after it.  The
    results are allocated from the side stream's pool and consumed after the join; when they are
    released (the next zero_grad) the side stream's next use of that memory is behind its next
    fork event, i.e. behind whatever the main stream had queued - including their consumers."""
    device = reads[0].device
    cur = torch.cuda.current_stream(device)
    side = stream(device)
    ev = _fork_event(device)
    ev.record(cur)
    side.wait_event(ev)
    _KEEP.setdefault(device.index, []).append(reads)
    set_stream = torch._C._cuda_setStream
    set_stream(stream_id=side.stream_id, device_index=side.device_index, device_type=side.device_type)
    try:
        out = fn()
    finally:
        set_stream(stream_id=cur.stream_id, device_index=cur.device_index, device_type=cur.device_type)
    _queue_join(device, side)
    return out

