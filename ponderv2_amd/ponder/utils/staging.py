"""Input staging on a second host thread.

A training step is ~2000 kernel launches issued by one Python thread, and on MI355X that thread -
not the GPU - bounds the reduced-precision step and ties with the GPU in fp32.  Part of what it
issues is not the step at all but the NEXT batch's input work: the host->device copies, the
optional device-side GridSample and the ~250 launches of the sparse-conv geometry (ten rulebooks:
hash tables, neighbour tables, radix sorts, compactions).  The reference gives such work to its
dataloader worker processes; here it runs where the data already lives - on the device, on the
geometry side stream - so what remains is to take its LAUNCH cost off the training thread:
``BackgroundStager`` runs ``stage(batch)`` on one worker thread while the main thread enqueues
the current step.  ctypes foreign calls and torch's operator bindings release the GIL, so the two
threads' launch calls genuinely overlap; only their Python byte code interleaves.

Ordering is by streams, not by threads: the geometry is built on its own stream after an event of
the consumer's stream (kernels.prefetch_unet_geometry) and handed back through
``PendingGeometry`` (event + pinned counts), exactly as in the single-threaded lookahead.
"""
from concurrent.futures import ThreadPoolExecutor

import torch


class _Ready:
    """A finished result with the Future interface (the synchronous fallback)."""

    def __init__(self, value):
        self._value = value

    def result(self):
        return self._value


class BackgroundStager:
    def __init__(self, stage_fn, device, enabled=True):
        self.stage_fn, self.device = stage_fn, torch.device(device)
        self._pool = None
        if enabled and self.device.type == "cuda":
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="pv2-stage")

    @property
    def threaded(self):
        return self._pool is not None

    def _run(self, item):
        torch.cuda.set_device(self.device)     # the current device is per thread
        return self.stage_fn(item)

    def submit(self, item):
        """Start staging ``item``; ``.result()`` of the returned handle is the staged batch (and
        re-raises whatever ``stage_fn`` raised)."""
        if self._pool is None:
            return _Ready(self.stage_fn(item))
        return self._pool.submit(self._run, item)

    def shutdown(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
