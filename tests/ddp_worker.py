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
        local_backward()
        victim = dict(model.named_parameters())[sorted(g_ddp)[0]]
        mine = victim.grad.clone()
        if rank == 1:
            victim.grad = None
        FlatGradSync(model.parameters(), uniform_usage=False).sync()
        ref0 = mine.clone()
        dist.broadcast(ref0, src=0)
        skip_ok = bool(victim.grad is not None and torch.allclose(victim.grad, ref0 / world, atol=1e-7)
                       and all(p.grad is None for n, p in model.named_parameters() if n not in g_ddp))
        torch.save(dict(rank=rank, worst=worst, loss=float(out["loss"]), n_grads=len(g_ddp),
                        worst_flat=worst_flat, unused_stay_none=unused_stay_none, skip_ok=skip_ok,
                        n_params=len(names)),
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
    model = torch.nn.Sequential(*[torch.nn.Linear(96, 96) for _ in range(30)], torch.nn.Linear(96, 3))
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
