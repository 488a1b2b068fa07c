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
multi-dataset model trains a different condition's norms per rank and step) exchanges a usage flag
per parameter - the counterpart of DDP's ``find_unused_parameters`` bitmap exchange.  Which parameters
received a gradient is known to the HOST as soon as backward has been enqueued, so the flags travel
host to host (a gloo all-reduce of one small CPU tensor: the group itself in the CPU tests, a gloo
twin of the default group beside RCCL) and the device never has to be waited for; reading reduced
flags back from the device - the round-4 form, still the fallback for sub-groups and for
PV2_GSYNC_HOST_FLAGS=0 - blocks the host until backward and the reduction have finished, which made
the multi-dataset step host-bound (34.9 ms against 21.5, profiles/r05_workloads.txt).

Arena blocks.  The sparse backbone's native executor hands ALL its parameter gradients back as views
of one flat buffer (ponderv2_amd/spunet_native.py: 177 tensors, 150 of the step's 160 MB).  When the
first synchronised step finds such a family - gradients that are contiguous views of one storage -
the flat buffer mirrors the family's layout (same relative offsets, alignment gaps included), and
from then on the family moves with ONE copy each way instead of one per tensor (1.8 ms of host time
per step for 2 x 229 small copies, profiles/r04_grad_sync_host_profile.txt).  The collectives always
run on the flat buffer, whose layout is fixed after the first step and agreed between the ranks
(one all-reduce of a layout signature, once): a step - or a rank - whose gradients do not sit at the
recorded offsets simply falls back to per-tensor copies into the same places, so the ranks can
never disagree about what they reduce.

Overlap with the backward pass (round 5; what the reference gets from DDP's buckets,
ponder/engines/defaults.py:22-43).  With ``overlap=True`` the sparse executor reports its
parameter-gradient arena from INSIDE the backward node (``spunet_native.GRAD_SLAB_HOOK``): it walks
the units last to first, so the arena is final from its END towards its beginning, and it records
an event pair (training stream / weight-gradient side stream) per finished slab of ~``slab_mb``
(pv2_unet_backward_ev).  Each slab is all-reduced IN PLACE on the arena, behind its events, through a
communication stream - while the units below are still running - and the parameters' ``.grad``
(views of the arena) hold the averages when ``sync()`` has waited for the handles: no copy into the
flat buffer and none back for 150 of the step's 160 MB.  Everything else (the dense U-Net's arena,
the heads, the stem) travels through the flat buffer as before.  The collectives are issued in the
same order on every rank (slabs in completion order, then the flat slices); a step whose backward
did not run natively on this rank reduces the same slabs from a staging buffer inside ``sync()``.
"""
import os

import torch
import torch.distributed as dist


def _dense_strides(g):
    """The strides of ``g`` when its elements occupy exactly ``numel`` consecutive storage elements in
    SOME dimension order (contiguous, channels-last, ...), else None."""
    if g.is_contiguous():
        return tuple(g.stride())
    expect = 1
    for st, sz in sorted((st, sz) for sz, st in zip(g.shape, g.stride()) if sz > 1):
        if st != expect:
            return None
        expect *= sz
    return tuple(g.stride())


STREAM_OPS = os.environ.get("PV2_GSYNC_STREAM_OPS", "1") != "0"


class _OnStream:
    """What ``wait()`` of an asynchronous collective's handle does, for work that is simply queued on a
    stream: the caller's current stream waits for everything that stream holds so far."""

    def __init__(self, stream):
        self.event = torch.cuda.Event()
        self.event.record(stream)

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class FlatGradSync:
    def __init__(self, params, process_group=None, slice_mb: float = 64.0, uniform_usage: bool = True,
                 use_blocks: bool = True, overlap: bool = False, slab_mb: float = 48.0,
                 alias_grads: bool = True, host_flags: bool = True):
        self.params = [p for p in params if p.requires_grad]
        # after the reduction ``.grad`` of a parameter outside the arenas becomes the flat buffer's view
        # instead of receiving a copy of it (False: copy back into the tensors autograd produced)
        self.alias_grads = bool(alias_grads)
        # in-place reduction of the sparse executor's arena behind per-slab events (see the module
        # docstring).  Every rank issues the slab collectives in the same order whatever the model does
        # with its OTHER parameters (the arena's layout is agreed once, ``_arm``; a rank whose backward did
        # not fill it stages zeros / its gradients, ``_finish_inplace``) - so the route also serves
        # ``uniform_usage=False`` (round 6: the multi-dataset model, whose arena holds the conv weights only)
        self.overlap = bool(overlap)
        self.slab_elems = max(int(slab_mb * 2 ** 20 // 4), 1)
        self._index_of = {id(p): i for i, p in enumerate(self.params)}
        self._arena_layout = None     # (arena numel, ((lo, hi), ...), ((param index, offset, numel), ...))
        self._covered = frozenset()   # parameter indices whose gradients live in the in-place arena
        self._inflight = None         # this step's (arena, [(work, view)])
        self._stage = None
        self._comm = {}
        # the in-place route is ARMED by the first sync(): until every rank has confirmed that its
        # backward reported the same arena layout, the hook issues no collective of its own (a rank
        # whose first backward ran module by module would otherwise pair its first flat slice with the
        # others' first slab)
        self._armed = False
        self._pending_first = None    # (arena, slabs) of the step that arms
        self.group = process_group
        self.numel = sum(p.numel() for p in self.params)
        self.slice_elems = max(int(slice_mb * 2 ** 20 // 4), 1)
        self._flat = None
        self._views = None
        self._blocks = []
        self.use_blocks = use_blocks
        self._flag_key, self._flags = None, None
        # None: not decided yet; False: the flags ride with the data and are read back from the device
        self._host_group = None if host_flags else False
        self._member_strides = {}     # parameter index -> strides of its gradient inside an arena block
        self.uniform_usage = uniform_usage
        # uniform usage is an ASSUMPTION about the model; it is checked, not trusted (see sync)
        self._last_used, self._steps, self.check_every = None, 0, 64

    def _arena_families(self, min_bytes=1 << 20):
        """Families of used parameters whose gradients are DENSE views of ONE storage (contiguous, or
        any other dimension order that fills ``numel`` consecutive elements - the dense U-Net keeps its
        weight gradients channels-last): [(storage ptr, span start (elements from the storage base), span
        length, [(param index, offset in the span)])], largest first.  The strides go to
        ``_member_strides``: the flat buffer's view of such a member has the same ones, so that ONE copy
        of the span moves every member."""
        by_storage, strides = {}, {}
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None or g.dtype != torch.float32 or i in self._covered:
                continue
            sd = _dense_strides(g)
            if sd is None:
                continue
            strides[i] = sd
            st = g.untyped_storage()
            by_storage.setdefault(st.data_ptr(), []).append((i, (g.data_ptr() - st.data_ptr()) // 4, g.numel()))
        self._candidate_strides = strides
        fams = []
        for ptr, members in by_storage.items():
            if len(members) < 8:
                continue
            members.sort(key=lambda m: m[1])
            # a block is written back as ONE span: it may hold nothing but its members and the
            # allocator's alignment padding between them (< 64 floats, spunet_native._ALIGN).  A larger
            # gap could hide another tensor (a frozen parameter's gradient, another group's, the padded
            # stem weight's) - the family is cut there; overlapping members end it too
            runs, run = [], [members[0]]
            for m in members[1:]:
                gap = m[1] - (run[-1][1] + run[-1][2])
                if 0 <= gap < 64:
                    run.append(m)
                else:
                    runs.append(run)
                    run = [m]
            runs.append(run)
            for run in runs:
                lo, hi = run[0][1], run[-1][1] + run[-1][2]
                if len(run) >= 8 and sum(n for _, _, n in run) * 4 >= min_bytes:
                    fams.append((ptr, lo, hi - lo, [(i, o - lo) for i, o, _ in run]))
        fams.sort(key=lambda f: -f[2])
        return fams

    def _buffers(self):
        if self._flat is None:
            ref = self.params[0]
            # layout: arena blocks (with their gaps) first, then the remaining parameters, then one
            # usage flag per parameter.  ``self.numel`` = everything in front of the flags.
            fams = self._arena_families() if (self.use_blocks and self.uniform_usage) else []
            if (self.use_blocks and self.uniform_usage and dist.is_initialized()
                    and dist.get_world_size(self.group) > 1):
                # the ranks must agree on the layout: one small reduction, once - issued by EVERY rank
                # whether or not it found a family itself (a rank without one contributes signature 0:
                # skipping the collective there would pair this reduce with the other ranks' first
                # gradient slice - ADVICE round 4).  The signature covers every member's (parameter
                # index, offset) and every span, so two different layouts of equal size do not pass.
                sig = 0
                for k, (_, _, span, m) in enumerate(fams):
                    sig = (sig * 1000003 + (k + 1) * 7919 + span) % 2147483629
                    for i, o in m:
                        sig = (sig * 1000003 + i * 31 + o) % 2147483629
                        for sd in self._candidate_strides[i]:
                            sig = (sig * 1000003 + sd) % 2147483629
                t = torch.tensor([float(sig), -float(sig)], dtype=torch.float64, device=ref.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                hi, lo = t.tolist()
                if hi != -lo:
                    fams = []
            in_block, off, blocks = {}, 0, []
            for _, _, span, members in fams:
                blocks.append((off, span, members))
                for i, o in members:
                    in_block[i] = off + o
                off += span
            offsets = []
            for i, p in enumerate(self.params):
                if i in in_block:
                    offsets.append(in_block[i])
                elif i in self._covered:   # reduced in place on the executor's arena: no room here
                    offsets.append(None)
                else:
                    offsets.append(off)
                    off += p.numel()
            self.numel = off
            self._blocks = blocks
            self._flat = torch.zeros(self.numel + len(self.params), dtype=torch.float32, device=ref.device)
            self._offsets = offsets
            self._member_strides = {i: self._candidate_strides[i] for i in in_block}
            self._views = []
            for i, (o, p) in enumerate(zip(offsets, self.params)):
                if o is None:
                    self._views.append(None)
                    continue
                piece = self._flat[o:o + p.numel()]
                sd = self._member_strides.get(i)
                # (a block member's view mirrors its gradient's strides: the block copy is a byte copy)
                self._views.append(piece.view_as(p) if sd is None or sd == tuple(p.stride())
                                   else piece.as_strided(tuple(p.shape), sd))
        return self._flat, self._views

    # ------------------------------------------------------------------ overlap with the backward
    _post_reduce = None   # (test hook, see _reduce_slabs)
    _warned_accumulate = False

    def attach(self):
        """Register with the sparse executor (``overlap=True``): from now on its backward node reports
        the parameter-gradient arena and the per-slab events to ``_on_arena``."""
        if self.overlap:
            from ponderv2_amd import spunet_native

            spunet_native.GRAD_SLAB_HOOK = self
        return self

    def detach(self):
        from ponderv2_amd import spunet_native

        if spunet_native.GRAD_SLAB_HOOK is self:
            spunet_native.GRAD_SLAB_HOOK = None

    def wants(self, tensors) -> bool:
        """Called by the executor before it lays out the slabs: True when every gradient of ``tensors``
        becomes the ``.grad`` of one of this object's parameters (leaves; no accumulation pending)."""
        if not (self.overlap and dist.is_available() and dist.is_initialized()):
            return False
        if self._inflight is not None or self._pending_first is not None:
            # a second backward before sync() (gradient accumulation, a model that runs the backbone
            # twice, a skipped sync): its gradients must not land in an arena that is being reduced in
            # place.  This backward takes the staged route (the executor keeps its own arena; sync()
            # reduces those gradients through the flat buffer) - said once, not raised from inside an
            # autograd node.
            if not self._warned_accumulate:
                self._warned_accumulate = True
                import warnings

                warnings.warn("FlatGradSync(overlap=True): a backward ran before the previous one was "
                              "sync()ed; its gradients are reduced after the backward, not overlapped")
            if self._inflight is not None:      # nothing may add into a slab that is still being reduced
                for work, _ in self._inflight[1]:
                    work.wait()
            return False
        return all(id(t) in self._index_of and t.is_leaf and t.grad is None for t in tensors)

    def _reduce_slabs(self, arena, slabs, events=None):
        """all-reduce ``arena[lo:hi]`` for every slab, in order, asynchronously; on a device behind the
        slab's events on the communication stream."""
        avg = dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        works = []
        if arena.is_cuda:
            dev = arena.device
            comm = self._comm.get(dev.index)
            if comm is None:
                comm = self._comm[dev.index] = torch.cuda.Stream(device=dev)
            cur = torch.cuda.current_stream(dev)
            with torch.cuda.stream(comm):
                if events is None:
                    comm.wait_stream(cur)
                arena.record_stream(comm)
                for j, (lo, hi) in enumerate(slabs):
                    if events is not None:
                        for ev in events[j]:
                            comm.wait_event(ev)
                    view = arena[lo:hi]
                    if STREAM_OPS:
                        # a SYNCHRONOUS collective issued under the communication stream: stream-ordered on
                        # it (the host does not block), and where the backend runs such ops on the caller's
                        # stream, one HIP stream fewer per rank than the asynchronous form's internal one
                        dist.all_reduce(view, op=op, group=self.group)
                        works.append((_OnStream(comm), view))
                    else:
                        works.append((dist.all_reduce(view, op=op, group=self.group, async_op=True), view))
                    if self._post_reduce is not None:
                        # test hook (tests/test_gpu_grad_overlap.py): an in-place edit of the slab right
                        # behind its reduction on the communication stream.  With ONE rank a reduction is
                        # the identity, so a slab reduced BEFORE the side stream wrote its weight
                        # gradients would go unnoticed; the edit would not - it would be overwritten.
                        works[-1][0].wait()
                        self._post_reduce(view)
        else:
            for lo, hi in slabs:
                view = arena[lo:hi]
                works.append((dist.all_reduce(view, op=op, group=self.group, async_op=True), view))
        return works, avg

    def _on_arena(self, arena, members, slabs, events):
        """From inside the executor's backward: ``arena`` the flat parameter-gradient buffer,
        ``members`` [(parameter, offset, numel)], ``slabs`` [(lo, hi)] in completion order, ``events``
        [(main, side)] per slab (None on the host)."""
        layout = (arena.numel(), tuple(slabs),
                  tuple((self._index_of[id(t)], off, n) for t, off, n in members))
        if self._arena_layout is None:
            if self._flat is not None:
                raise RuntimeError("FlatGradSync(overlap=True): the executor's arena appeared after the "
                                   "flat layout was fixed; attach() before the first step")
            self._arena_layout = layout
            self._covered = frozenset(i for i, _, _ in layout[2])
        elif layout != self._arena_layout:
            raise RuntimeError("FlatGradSync(overlap=True): the executor's gradient arena changed its "
                               "layout between steps; use overlap=False for this model")
        if not self._armed:           # first step: no collective before the ranks have agreed (sync())
            self._pending_first = (arena, tuple(slabs))
            return
        works, avg = self._reduce_slabs(arena, slabs, events)
        self._inflight = (arena, works, avg)

    def _arm(self, world):
        """First sync(): do ALL ranks hold the same arena layout?  Then the in-place route is on from now on
        (and this step's slabs are reduced here, in place); otherwise it is off everywhere and the arena's
        parameters travel through the flat buffer like any others."""
        have = self._arena_layout is not None
        agreed = have
        if world > 1:
            sig = 0
            if have:
                numel, slabs, members = self._arena_layout
                sig = numel % 2147483629
                for lo, hi in slabs:
                    sig = (sig * 1000003 + lo * 31 + hi) % 2147483629
                for i, off, n in members:
                    sig = (sig * 1000003 + i * 131 + off * 7 + n) % 2147483629
                sig += 1
            ref = self.params[0]
            t = torch.tensor([float(sig), -float(sig)], dtype=torch.float64, device=ref.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            hi, lo = t.tolist()
            agreed = have and hi == -lo and hi > 0
        if agreed:
            self._armed = True
            arena, slabs = self._pending_first
            works, avg = self._reduce_slabs(arena, list(slabs))
            self._inflight = (arena, works, avg)
        else:
            self.overlap = False
            self._arena_layout, self._covered = None, frozenset()
        self._pending_first = None

    def _finish_inplace(self, world):
        """Wait for the slab reductions of this step - issuing them from a staging buffer first when the
        backward did not run natively on this rank - and leave the averages in the gradients."""
        if self._arena_layout is None:
            return
        numel, slabs, members = self._arena_layout
        staged = None
        if self._inflight is None:
            ref = self.params[0]
            if self._stage is None or self._stage.device != ref.device:
                self._stage = torch.empty(numel, dtype=torch.float32, device=ref.device)
            staged = self._stage
            staged.zero_()
            live = [(i, off, n) for i, off, n in members if self.params[i].grad is not None]
            if live:
                torch._foreach_copy_([staged[off:off + n].view_as(self.params[i]) for i, off, n in live],
                                     [self.params[i].grad for i, _, _ in live])
            works, avg = self._reduce_slabs(staged, list(slabs))
        else:
            _, works, avg = self._inflight
        for work, view in works:
            work.wait()
            if not avg:
                view.div_(world)
        if staged is not None:
            live = [(i, off, n) for i, off, n in members if self.params[i].grad is not None]
            if live:
                torch._foreach_copy_([self.params[i].grad for i, _, _ in live],
                                     [staged[off:off + n].view_as(self.params[i]) for i, off, n in live])
        self._inflight = None

    def _block_sources(self):
        """For every arena block: the live gradient storage span as one tensor when this step's
        gradients sit exactly where the layout says (then ONE copy moves the block), else None."""
        out = []
        for off, span, members in self._blocks:
            i0, o0 = members[0]
            g0 = self.params[i0].grad
            src = None
            if g0 is not None and tuple(g0.stride()) == self._member_strides.get(i0):
                st = g0.untyped_storage()
                start = (g0.data_ptr() - st.data_ptr()) // 4 - o0          # span start in the storage
                ok = start >= 0 and (start + span) * 4 <= st.nbytes()
                base = st.data_ptr() + 4 * start
                for i, o in members:
                    g = self.params[i].grad
                    if (g is None or tuple(g.stride()) != self._member_strides.get(i)
                            or g.data_ptr() != base + 4 * o):
                        ok = False
                        break
                if ok:
                    src = torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, start, (span,))
            out.append(src)
        return out

    def _host_flag_group(self):
        """The group the usage flags are exchanged on from host memory, or False.  Decided ONCE, by every
        rank together (creating a group is collective, and a rank that failed to create its twin must
        not leave the others waiting on it)."""
        if self._host_group is not None:
            return self._host_group
        import os
        if dist.get_backend(self.group) == "gloo":
            self._host_group = self.group if self.group is not None else dist.group.WORLD
            return self._host_group
        ok, twin = 0.0, False
        if os.environ.get("PV2_GSYNC_HOST_FLAGS", "1") != "0" and (self.group is None or self.group is dist.group.WORLD):
            try:
                twin, ok = dist.new_group(backend="gloo"), 1.0
            except Exception:   # noqa: BLE001 - no gloo transport on this machine: the device read remains
                twin = False
        t = torch.tensor([ok], device=self._buffers()[0].device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        self._host_group = twin if float(t) > 0.5 else False
        return self._host_group

    @torch.no_grad()
    def sync(self):
        """Average ``.grad`` over the ranks of the group (call between backward and the optimiser
        step).  A no-op on a single process without a process group."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.group)
        if self.overlap and not self._armed:
            self._arm(world)          # (one small collective, once; may switch the in-place route off)
        if self.overlap:
            self._finish_inplace(world)   # (same position in every rank's sequence of collectives)
        flat, views = self._buffers()
        used = [i for i, p in enumerate(self.params) if p.grad is not None and i not in self._covered]
        if not self.uniform_usage:
            # a silent rank contributes zeros: clear the slots of the parameters WITHOUT a gradient on this
            # rank (they hold last step's averages) - one multi-tensor launch over exactly those slots.
            # The used slots are overwritten below, or ARE the gradients (``alias_grads`` + in-place
            # accumulation: clearing the whole buffer here wiped them); nothing else of the
            # 160 MB buffer is touched.
            used_set = set(used)
            unused = [views[i] for i in range(len(self.params)) if i not in used_set and i not in self._covered]
            if unused:
                torch._foreach_zero_(unused)
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
            # (fill_ takes the scalar as a kernel argument.  ``flat[i] = python_float`` is a host -> device
            # copy from pageable memory, which waits for everything queued on the stream: 12 ms of
            # blocked host per step, profiles/r04_grad_sync_host_profile.txt)
            flat[self.numel:self.numel + 1].fill_(float(sum(i + 1 for i in used)))
        moved = set()
        sources = self._block_sources() if self._blocks else []
        for (off, span, members), src in zip(self._blocks, sources):
            if src is not None:     # the whole arena block in one copy
                flat[off:off + span].copy_(src)
                moved.update(i for i, _ in members)
        rest = [i for i in used if i not in moved]
        if rest:
            self._gather(rest, flat, views)
        host_flags = flags_work = None
        if not self.uniform_usage:
            key = tuple(used)
            hg = self._host_flag_group()
            if self._flag_key != key:  # the local set changes rarely: its flag vector is cached
                f = torch.zeros(len(self.params), dtype=torch.float32)
                f[used] = 1.0
                self._flag_key, self._flags = key, (f if hg else f.to(flat.device))
            if hg:
                # host to host: nothing here waits for the device (the data follows below, on RCCL)
                # (asynchronous: the host goes on to enqueue the data reduction while the flags travel)
                host_flags = self._flags.clone()
                flags_work = dist.all_reduce(host_flags, group=hg, async_op=True)
            else:
                flat[self.numel:].copy_(self._flags)
        # RCCL averages in the reduction itself; other backends (gloo in the CPU tests) sum
        avg = dist.get_backend(self.group) == "nccl" and self.uniform_usage
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        end = self.numel + 1 if self.uniform_usage else (self.numel if host_flags is not None else flat.numel())
        handles = [dist.all_reduce(flat[a:min(a + self.slice_elems, end)], op=op, group=self.group,
                                   async_op=True) for a in range(0, end, self.slice_elems)]
        for h in handles:
            h.wait()
        if flags_work is not None:
            flags_work.wait()
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
            anywhere = [p.grad is not None and i not in self._covered for i, p in enumerate(self.params)]
        elif host_flags is not None:
            anywhere = (host_flags > 0).tolist()
        else:  # somebody else's parameters: one small (blocking) device read, only when this rank skipped some
            anywhere = (flat[self.numel:] > 0).tolist()
        if not avg:
            flat[:self.numel].div_(world)
        for (off, span, members), src in zip(self._blocks, sources):
            if src is not None:     # averages back into the arena, one copy
                src.copy_(flat[off:off + span])
        targets, origins = [], []
        for i, p in enumerate(self.params):
            if not anywhere[i] or i in moved:
                continue
            if self.alias_grads and p.grad is not None and views[i].stride() == p.grad.stride():
                # the average stays where it is: ``.grad`` becomes the flat buffer's view (what
                # DistributedDataParallel(gradient_as_bucket_view=True) does) - no copy back, no launch
                p.grad = views[i]
                continue
            if p.grad is None:
                p.grad = torch.empty_like(p)
            targets.append(p.grad)
            origins.append(views[i])
        if targets:
            torch._foreach_copy_(targets, origins)

    def _gather(self, rest, flat, views):
        """The gradients of ``rest`` (ascending parameter indices) into their flat slots.  Adjacent slots
        are filled by ONE ``torch.cat`` (a batched kernel: the per-tensor route is a hipMemcpyAsync each
        on ROCm - 40 of them per step for the heads and norms of this model); gradients that already ARE
        their flat views (``alias_grads`` + in-place accumulation) are left alone."""
        runs, run = [], []
        for i in rest:
            g = self.params[i].grad
            if g.data_ptr() == views[i].data_ptr() and g.stride() == views[i].stride():
                if run:
                    runs.append(run)
                    run = []
                continue
            if run and (self._offsets[run[-1]] + self.params[run[-1]].numel() != self._offsets[i]
                        or not g.is_contiguous()):
                runs.append(run)
                run = []
            if g.is_contiguous() and views[i].is_contiguous():
                run.append(i)
            else:
                views[i].copy_(g)
        if run:
            runs.append(run)
        for run in runs:
            if len(run) == 1:
                views[run[0]].copy_(self.params[run[0]].grad)
                continue
            lo = self._offsets[run[0]]
            hi = self._offsets[run[-1]] + self.params[run[-1]].numel()
            torch.cat([self.params[i].grad.reshape(-1) for i in run], out=flat[lo:hi])
