"""Worker functions for the world_size-2 gloo tests (run in spawned processes, CPU only)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def tiny_model_cfg():
    import golden_cases as gc

    return gc.indoor_model_cfg(dict(gc.SMALL_BACKBONE, channels=(16, 32, 48, 64, 64, 48, 32, 96)),
                               grid_shape=(32, 32, 8), ray_nsample=6)


def tiny_scene(seed):
    from ponderv2_amd.ponder.datasets import make_scene

    return make_scene(seed, n_raw=5000, num_views=2, image_hw=(24, 32))


def grad_sync_worker(rank, world, port, out_dir):
    """DDP gradient == mean over ranks of the local gradients; scenes are sharded by rank."""
    from oracle import cpu_backend
    from oracle.detweights import fill_deterministic
    from ponderv2_amd.ponder.datasets import collate_fn
    from ponderv2_amd.ponder.engines.defaults import create_ddp_model
    from ponderv2_amd.ponder.models import build_model
    from ponderv2_amd.ponder.utils import comm
    from ponderv2_amd.ponder.utils.config import ConfigDict

    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    with cpu_backend.installed():
        model = build_model(ConfigDict(tiny_model_cfg()))
        fill_deterministic(model)
        model.train()
        ddp = create_ddp_model(model, broadcast_buffers=False, find_unused_parameters=True)
        assert comm.get_world_size() == world and comm.get_rank() == rank
        batch = collate_fn([tiny_scene(50 + rank)])  # each rank renders its own scene

        def run(sync):
            torch.manual_seed(7)
            ddp.zero_grad(set_to_none=True)
            ctx = torch.enable_grad() if sync else ddp.no_sync()
            with ctx:
                out = ddp({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
                out["loss"].backward()
            return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, out

        g_ddp, out = run(sync=True)
        # BatchNorm running stats advanced by the first pass do not enter train-mode gradients
        g_loc, _ = run(sync=False)
        worst = 0.0
        for n in g_ddp:
            avg = g_loc[n].clone()
            dist.all_reduce(avg)
            avg /= world
            worst = max(worst, (avg - g_ddp[n]).abs().max().item() / (avg.abs().max().item() + 1e-12))
        # the DDP-free reduction (utils/grad_sync.py) averages the same local gradients to the
        # same result - with every rank using the same parameters ...
        from ponderv2_amd.ponder.utils.grad_sync import FlatGradSync

        names = [n for n, p in model.named_parameters() if p.requires_grad]

        def local_backward():
            torch.manual_seed(7)
            model.zero_grad(set_to_none=True)
            model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})["loss"].backward()

        local_backward()
        FlatGradSync(model.parameters(), slice_mb=0.05).sync()
        worst_flat = max((dict(model.named_parameters())[n].grad - g_ddp[n]).abs().max().item()
                         / (g_ddp[n].abs().max().item() + 1e-12) for n in g_ddp)
        unused_stay_none = all(p.grad is None for n, p in model.named_parameters() if n not in g_ddp)
        # ... and with rank 1 skipping one parameter that rank 0 trains (the multi-dataset case):
        # both ranks end with half of rank 0's gradient; parameters unused everywhere stay None
        # (the usage flags exchanged host to host, and riding with the data + read back: same result)
        skip_ok = True
        for host_flags in (True, False):
            local_backward()
            victim = dict(model.named_parameters())[sorted(g_ddp)[0]]
            mine = victim.grad.clone()
            if rank == 1:
                victim.grad = None
            FlatGradSync(model.parameters(), uniform_usage=False, host_flags=host_flags).sync()
            ref0 = mine.clone()
            dist.broadcast(ref0, src=0)
            skip_ok = skip_ok and bool(
                victim.grad is not None and torch.allclose(victim.grad, ref0 / world, atol=1e-7)
                and all(p.grad is None for n, p in model.named_parameters() if n not in g_ddp)
                and all(torch.allclose(dict(model.named_parameters())[n].grad, g_ddp[n], rtol=1e-5, atol=1e-7)
                        for n in sorted(g_ddp)[1:]))
        # ... and with gradients KEPT IN PLACE between steps (zero_grad(set_to_none=False)): after the
        # first sync ``.grad`` is a view of the flat buffer (alias_grads), the next backward accumulates
        # into it, and the next sync must reduce THAT - not a buffer it has just cleared
        inplace_ok = True
        for uniform in (False, True):
            local_backward()
            reducer = FlatGradSync(model.parameters(), uniform_usage=uniform)
            reducer.sync()
            for step in range(2):
                torch.manual_seed(7)
                model.zero_grad(set_to_none=False)
                model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})["loss"].backward()
                reducer.sync()
                inplace_ok = inplace_ok and all(
                    torch.allclose(dict(model.named_parameters())[n].grad, g_ddp[n], rtol=1e-5, atol=1e-7)
                    for n in g_ddp)
        torch.save(dict(rank=rank, worst=worst, loss=float(out["loss"]), n_grads=len(g_ddp),
                        worst_flat=worst_flat, unused_stay_none=unused_stay_none, skip_ok=skip_ok,
                        inplace_ok=inplace_ok, n_params=len(names)),
                   os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def arena_sync_worker(rank, world, port, out_dir):
    """FlatGradSync with gradients that live in ONE flat arena (what the sparse backbone's native
    executor hands back): the block route (one copy each way) gives the rank means, keeps working
    when a later step's gradients are ordinary tensors again, and when only ONE rank's are."""
    from ponderv2_amd.ponder.utils.grad_sync import FlatGradSync

    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    torch.manual_seed(0)
    # (two 5-D conv weights in front: in the arena their gradients are CHANNELS-LAST views, as the dense
    # U-Net node hands them back - dense, not contiguous; they must travel inside the block too)
    model = torch.nn.Sequential(torch.nn.Conv3d(8, 16, 3), torch.nn.Conv3d(16, 8, 3),
                                *[torch.nn.Linear(96, 96) for _ in range(28)], torch.nn.Linear(96, 3))
    params = list(model.parameters())
    sync = FlatGradSync(params, slice_mb=0.02)

    def local(step, in_arena):
        g = torch.Generator().manual_seed(100 * step + rank)
        grads = [torch.randn(p.shape, generator=g) for p in params]
        if in_arena:   # views of one buffer, 64-float aligned like spunet_native._Arena (gaps = garbage)
            sizes = [(p.numel() + 63) // 64 * 64 for p in params[:-2]]
            arena = torch.full((sum(sizes),), float("nan"))
            off = 0
            for p, gr, n in zip(params[:-2], grads, sizes):
                if p.dim() == 5:
                    co, ci, k = p.shape[0], p.shape[1], p.shape[2]
                    view = arena[off:off + p.numel()].view(co, k, k, k, ci).permute(0, 4, 1, 2, 3)
                    assert not view.is_contiguous() and view.shape == p.shape
                else:
                    view = arena[off:off + p.numel()].view_as(p)
                view.copy_(gr)
                p.grad = view
                off += n
            for p, gr in zip(params[-2:], grads[-2:]):   # the last layer: ordinary tensors
                p.grad = gr.clone()
        else:
            for p, gr in zip(params, grads):
                p.grad = gr.clone()
        return grads

    ok, modes = True, [True, True, False, rank == 0, True]
    for step, in_arena in enumerate(modes):
        grads = local(step, in_arena)
        want = []
        for gr in grads:
            t = gr.clone()
            dist.all_reduce(t)
            want.append(t / world)
        sync.sync()
        ok = ok and all(torch.allclose(p.grad, w, atol=1e-6) for p, w in zip(params, want))
        ok = ok and all(torch.isfinite(p.grad).all() for p in params)
    torch.save(dict(rank=rank, ok=bool(ok), blocks=len(sync._blocks),
                    block_members=sum(len(m) for _, _, m in sync._blocks)),
               os.path.join(out_dir, f"arena{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def overlap_sync_worker(rank, world, port, out_dir):
    """FlatGradSync(overlap=True): slabs of an executor-style arena reduced in place from the hook give,
    BIT FOR BIT, the averages of the flat (everything after backward) form - in steps where the hook
    fired on both ranks, on one rank only (the other stages the same slabs inside sync()), on neither;
    and a rank that finds no arena family on its first step still issues the layout-agreement reduce
    (ADVICE round 4)."""
    from ponderv2_amd.ponder.utils.grad_sync import FlatGradSync

    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    torch.manual_seed(0)

    def make():
        torch.manual_seed(0)
        return torch.nn.Sequential(*[torch.nn.Linear(96, 96) for _ in range(30)], torch.nn.Linear(96, 3))

    model_a, model_b = make(), make()
    pa, pb = list(model_a.parameters()), list(model_b.parameters())
    over = FlatGradSync(pa, slice_mb=0.02, overlap=True, slab_mb=0.1)
    flat = FlatGradSync(pb, slice_mb=0.02, overlap=False)
    # the "executor" owns all but the first and the last layer (stem-like / head-like parameters stay
    # ordinary tensors); arena laid out first to last, 64-float aligned
    owned = list(range(2, len(pa) - 2))
    offs, off = {}, 0
    for i in owned:
        offs[i] = off
        off += (pa[i].numel() + 63) // 64 * 64
    arena_numel = off
    spans, hi = [], arena_numel
    for i in reversed(owned):
        if hi - offs[i] >= over.slab_elems or i == owned[0]:
            spans.append((offs[i], hi))
            hi = offs[i]
    assert len(spans) >= 3

    def local(step):
        g = torch.Generator().manual_seed(1000 * step + rank)
        return [torch.randn(p.shape, generator=g) for p in pa]

    ok = True
    for step, fired in enumerate([True, True, rank == 0, False, True]):
        grads = local(step)
        for p, gr in zip(pb, grads):
            p.grad = gr.clone()
        for p in pa:
            p.grad = None
        assert over.wants([pa[i] for i in owned])      # (what the executor asks before it lays out slabs)
        if fired:
            arena = torch.full((arena_numel,), float("nan"))
            for i in owned:
                arena[offs[i]:offs[i] + pa[i].numel()].copy_(grads[i].reshape(-1))
            over._on_arena(arena, [(pa[i], offs[i], pa[i].numel()) for i in owned], spans, None)
            for i in owned:   # what AccumulateGrad does with the executor's views
                pa[i].grad = arena[offs[i]:offs[i] + pa[i].numel()].view_as(pa[i])
        else:
            for i in owned:
                pa[i].grad = grads[i].clone()
        for i in set(range(len(pa))) - set(owned):
            pa[i].grad = grads[i].clone()
        over.sync()
        flat.sync()
        ok = ok and all(torch.equal(a.grad, b.grad) for a, b in zip(pa, pb))
        want = []
        for gr in grads:
            t = gr.clone()
            dist.all_reduce(t)
            want.append(t / world)
        ok = ok and all(torch.allclose(a.grad, w, atol=1e-6) for a, w in zip(pa, want))
    armed = over._armed and len(over._covered) == len(owned)
    # a reducer whose FIRST backward reported an arena on rank 0 only: the in-place route must switch itself
    # off on both ranks (no collective was issued from the hook), and everything still averages correctly
    model_d = make()
    pd = list(model_d.parameters())
    lone = FlatGradSync(pd, slice_mb=0.02, overlap=True, slab_mb=0.1)
    for step in range(2):
        grads = local(50 + step)
        for p in pd:
            p.grad = None
        if lone.wants([pd[i] for i in owned]) and rank == 0 and step == 0:
            arena = torch.full((arena_numel,), float("nan"))
            for i in owned:
                arena[offs[i]:offs[i] + pd[i].numel()].copy_(grads[i].reshape(-1))
            lone._on_arena(arena, [(pd[i], offs[i], pd[i].numel()) for i in owned], spans, None)
            for i in owned:
                pd[i].grad = arena[offs[i]:offs[i] + pd[i].numel()].view_as(pd[i])
        else:
            for i in owned:
                pd[i].grad = grads[i].clone()
        for i in set(range(len(pd))) - set(owned):
            pd[i].grad = grads[i].clone()
        lone.sync()
        for gr, p in zip(grads, pd):
            t = gr.clone()
            dist.all_reduce(t)
            ok = ok and torch.allclose(p.grad, t / world, atol=1e-6)
    ok = ok and (not lone.overlap) and (not lone._armed) and len(lone._covered) == 0 and armed
    # Round 6: the multi-dataset model's plan (SpUNet-v1m3 PDNorm) under the overlapped route: only the conv
    # weights are members of the executor's arena - the BatchNorm pairs of that model are computed tensors,
    # their parameters ordinary gradients that differ from rank to rank in WHICH of them exist this step
    # (one condition per rank and step) - with uniform_usage=False.  Bit for bit the flat form's result.
    model_e, model_f = make(), make()
    pe, pf = list(model_e.parameters()), list(model_f.parameters())
    over2 = FlatGradSync(pe, slice_mb=0.02, overlap=True, slab_mb=0.1, uniform_usage=False)
    flat2 = FlatGradSync(pf, slice_mb=0.02, overlap=False, uniform_usage=False)
    owned_w = [i for i in owned if pe[i].dim() == 2]           # weights only
    offs_w, off = {}, 0
    for i in owned_w:
        offs_w[i] = off
        off += (pe[i].numel() + 63) // 64 * 64
    spans_w, hi = [], off
    for i in reversed(owned_w):
        if hi - offs_w[i] >= over2.slab_elems or i == owned_w[0]:
            spans_w.append((offs_w[i], hi))
            hi = offs_w[i]
    pd_ok = len(spans_w) >= 2
    for step in range(4):
        grads = local(200 + step)
        # the "conditions": every third bias exists on one rank only, alternating with the step
        silent = {i for i in range(len(pe)) if pe[i].dim() == 1 and i % 3 == 0 and (i // 3 + step + rank) % 2 == 0}
        for p, gr, i in zip(pf, grads, range(len(pf))):
            p.grad = None if i in silent else gr.clone()
        for p in pe:
            p.grad = None
        assert over2.wants([pe[i] for i in owned_w]), (step, over2.overlap, over2._inflight is None, over2._pending_first is None, [pe[i].grad is None for i in owned_w][:4])
        arena = torch.full((off,), float("nan"))
        for i in owned_w:
            arena[offs_w[i]:offs_w[i] + pe[i].numel()].copy_(grads[i].reshape(-1))
        over2._on_arena(arena, [(pe[i], offs_w[i], pe[i].numel()) for i in owned_w], spans_w, None)
        for i in owned_w:
            pe[i].grad = arena[offs_w[i]:offs_w[i] + pe[i].numel()].view_as(pe[i])
        for i in set(range(len(pe))) - set(owned_w):
            pe[i].grad = None if i in silent else grads[i].clone()
        over2.sync()
        flat2.sync()
        for i, (a, b) in enumerate(zip(pe, pf)):
            pd_ok = pd_ok and (a.grad is None) == (b.grad is None)
            if a.grad is not None:
                pd_ok = pd_ok and torch.equal(a.grad, b.grad)
        for i, gr in enumerate(grads):
            t = torch.zeros_like(gr) if i in silent else gr.clone()
            dist.all_reduce(t)
            if pe[i].grad is not None:
                pd_ok = pd_ok and torch.allclose(pe[i].grad, t / world, atol=1e-6)
    pd_ok = pd_ok and over2._armed and len(over2._covered) == len(owned_w)
    ok = ok and pd_ok
    # rank 1 has NO arena family on the first synchronised step, rank 0 has one
    model_c = make()
    pc = list(model_c.parameters())
    third = FlatGradSync(pc, slice_mb=0.02)
    grads = local(77)
    if rank == 0:
        sizes = [(p.numel() + 63) // 64 * 64 for p in pc]
        arena = torch.zeros(sum(sizes))
        o = 0
        for p, gr, n in zip(pc, grads, sizes):
            v = arena[o:o + p.numel()].view_as(p)
            v.copy_(gr)
            p.grad = v
            o += n
    else:
        for p, gr in zip(pc, grads):
            p.grad = gr.clone()
    third.sync()
    for gr, p in zip(grads, pc):
        t = gr.clone()
        dist.all_reduce(t)
        ok = ok and torch.allclose(p.grad, t / world, atol=1e-6)
    torch.save(dict(rank=rank, ok=bool(ok), slabs=len(spans), covered=len(over._covered),
                    blocks_third=len(third._blocks)),
               os.path.join(out_dir, f"overlap{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def trainer_main(cfg):
    """main_func for engines.launch(): two optimisation steps of the hook-driven Trainer."""
    from oracle import cpu_backend
    from ponderv2_amd.ponder.engines import default_setup
    from ponderv2_amd.ponder.engines.train import TRAINERS

    torch.set_num_threads(2)
    with cpu_backend.installed():
        cfg = default_setup(cfg)
        trainer = TRAINERS.build(dict(type=cfg.train.type, cfg=cfg))
        trainer.train()
        if trainer.writer is None:
            return
    return
