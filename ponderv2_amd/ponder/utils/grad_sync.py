"""Gradient averaging for data-parallel training without DistributedDataParallel's per-parameter
machinery.

The reference wraps its model in DDP (ponder/engines/defaults.py:31-56): a hook per parameter, a
traversal of the autograd graph per step (``find_unused_parameters``) and ~7 bucket all-reduces
overlapped with backward.  On MI355X that costs this model 3-8 ms of host and stream overhead per
33 ms step (measured with one rank: bench.py, PV2_BENCH_FORCE_DIST=1) for 160 MB of gradients that
RCCL moves across eight xGMI-connected GPUs in about a millisecond.  ``FlatGradSync`` does the
reduction the direct way instead: after backward, every gradient is copied into ONE flat fp32
buffer (a multi-tensor copy), the buffer is all-reduced in a few large slices - large messages are
what the point-to-point xGMI links like -, and the averages are copied back.  Parameters unused
everywhere keep ``grad is None``, as under DDP (so weight decay does not touch them).
``uniform_usage=True`` (single-dataset models: every rank runs the same code path, so the set of
parameters with gradients is the same everywhere) needs nothing else.  ``uniform_usage=False`` (the
multi-dataset model trains a different condition's norms per rank and step) reduces a usage flag
per parameter along with the data and reads the flags back - one small device read per step, the
counterpart of DDP's ``find_unused_parameters`` bitmap exchange.
"""
import torch
import torch.distributed as dist


class FlatGradSync:
    def __init__(self, params, process_group=None, slice_mb: float = 64.0, uniform_usage: bool = True):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.numel = sum(p.numel() for p in self.params)
        self.slice_elems = max(int(slice_mb * 2 ** 20 // 4), 1)
        self._flat = None
        self._views = None
        self._flag_key, self._flags = None, None
        self.uniform_usage = uniform_usage
        # uniform usage is an ASSUMPTION about the model; it is checked, not trusted (see sync)
        self._last_used, self._steps, self.check_every = None, 0, 64

    def _buffers(self):
        if self._flat is None:
            ref = self.params[0]
            # gradients, then one usage flag per parameter
            self._flat = torch.zeros(self.numel + len(self.params), dtype=torch.float32, device=ref.device)
            views, off = [], 0
            for p in self.params:
                views.append(self._flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self._views = views
        return self._flat, self._views

    @torch.no_grad()
    def sync(self):
        """Average ``.grad`` over the ranks of the group (call between backward and the optimiser
        step).  A no-op on a single process without a process group."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.group)
        flat, views = self._buffers()
        used = [i for i, p in enumerate(self.params) if p.grad is not None]
        if not self.uniform_usage:
            flat.zero_()
        else:
            # the same slices are overwritten every step and the rest stay 0 - as long as the set of
            # parameters with gradients does not change.  When it does change on this rank, the
            # slices that fell out are cleared (a silent rank must contribute zeros, not last
            # step's averages), and a checksum of the set rides along with the data: every
            # ``check_every`` steps (and right after a local change) it is read back and compared -
            # ranks that disagree about the set raise instead of silently diverging, which is what
            # DistributedDataParallel(find_unused_parameters=False) does in this situation.
            key = tuple(used)
            if self._last_used is not None and key != self._last_used:
                for i in set(self._last_used) - set(key):
                    views[i].zero_()
                self._steps = 0   # check at once
            self._last_used = key
            flat[self.numel] = float(sum(i + 1 for i in used))
        if used:
            torch._foreach_copy_([views[i] for i in used], [self.params[i].grad for i in used])
        if not self.uniform_usage:
            key = tuple(used)
            if self._flag_key != key:  # the local set changes rarely: its device copy is cached
                f = torch.zeros(len(self.params), dtype=torch.float32)
                f[used] = 1.0
                self._flag_key, self._flags = key, f.to(flat.device)
            flat[self.numel:].copy_(self._flags)
        # RCCL averages in the reduction itself; other backends (gloo in the CPU tests) sum
        avg = dist.get_backend(self.group) == "nccl" and self.uniform_usage
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        end = self.numel + 1 if self.uniform_usage else flat.numel()
        handles = [dist.all_reduce(flat[a:min(a + self.slice_elems, end)], op=op, group=self.group,
                                   async_op=True) for a in range(0, end, self.slice_elems)]
        for h in handles:
            h.wait()
        if self.uniform_usage:
            if self._steps % self.check_every == 0:
                mine = float(sum(i + 1 for i in used))
                total = float(flat[self.numel]) * (world if avg else 1.0)
                if abs(total - mine * world) > 0.5:
                    raise RuntimeError(
                        "FlatGradSync(uniform_usage=True): the ranks disagree about which parameters "
                        "received a gradient this step (a rank skipped a branch of the model); use "
                        "uniform_usage=False for such models")
            self._steps += 1
        if self.uniform_usage or len(used) == len(self.params):
            anywhere = [p.grad is not None for p in self.params]
        else:  # somebody else's parameters: one small read, only when this rank skipped some
            anywhere = (flat[self.numel:] > 0).tolist()
        if not avg:
            flat[:self.numel].div_(world)
        targets, sources = [], []
        for i, p in enumerate(self.params):
            if not anywhere[i]:
                continue
            if p.grad is None:
                p.grad = torch.empty_like(p)
            targets.append(p.grad)
            sources.append(views[i])
        if targets:
            torch._foreach_copy_(targets, sources)
