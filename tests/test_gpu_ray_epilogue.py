"""csrc/ray_epilogue.hip + ponderv2_amd/ray_epilogue.py through the C ABI: the per-ray epilogue of the
composite rows, the semantic head and EVERY loss term as one node, against the torch statement of the
same formulas (the reference's renderers.py:5-75 and base_surface_model.py:102-211) in float64."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


class _Head(torch.nn.Module):
    """The semantic decoder's parameters as the fused head holds them (n_blocks = 0: fc_c[0] then the
    last linear layer, fc_p multiplied by an exact zero)."""

    def __init__(self, n_in, hidden, c_sem):
        super().__init__()
        self.fc_p = torch.nn.Linear(3, hidden)
        self.fc_c = torch.nn.ModuleList([torch.nn.Linear(n_in, hidden)])
        self.lin0 = torch.nn.Linear(hidden, c_sem)

    @property
    def last_linear(self):
        return self.lin0


def _statement(comp, sdf, grad, starts, head, targets, cfg, B, bg, n_f2, n_geo):
    """The torch ops the node replaces, in whatever dtype the inputs have."""
    nv = comp.shape[1]
    R = comp.shape[0]
    f2c, geoc, g3c, nrm, rgbc, tcol, wsum, _ = comp.split([n_f2, n_geo, 3, 3, 3, 1, 1, nv - n_f2 - n_geo - 11], dim=1)
    bgt = torch.tensor(bg, dtype=comp.dtype, device=comp.device)
    rgb = torch.addcmul(rgbc + bgt, wsum, bgt, value=-1.0)
    xbar = torch.cat([g3c, f2c, geoc], dim=1)
    hidden = F.linear(xbar, head.fc_c[0].weight) + head.fc_c[0].bias * wsum
    sem = F.linear(hidden, head.last_linear.weight) + head.last_linear.bias * wsum
    depth = tcol / (wsum + 1e-10)
    lo, hi = torch.aminmax(starts.reshape(B, -1), dim=1)
    lo = lo[:, None].expand(B, R // B).reshape(-1, 1)
    hi = hi[:, None].expand(B, R // B).reshape(-1, 1)
    depth = torch.clamp(depth, lo, hi)
    lw = cfg["weights"]
    depth_gt = targets["depth"]
    valid = depth_gt > 0.0
    out = {}
    out["depth_loss"] = torch.sum(valid * torch.abs(depth_gt - depth)) / torch.clamp(torch.sum(valid), min=1.0) * lw["depth_loss"]
    out["rgb_loss"] = torch.mean(torch.abs(rgb - targets["rgb"])) * lw["rgb_loss"]
    out["psnr"] = 20.0 * torch.log10(1.0 / torch.mean((rgb - targets["rgb"]).pow(2)).sqrt())
    sem_pred = F.normalize(sem, dim=-1)
    sem_gt = targets["semantic"]
    ok = (valid * sem_gt.any(dim=-1, keepdim=True)).squeeze(-1).bool()
    logits = torch.mm(sem_pred, sem_gt.t()) / cfg["temperature"]
    labels = torch.arange(R, device=comp.device)
    labels = torch.where(ok, labels, torch.full_like(labels, -100))
    out["semantic_loss"] = F.cross_entropy(logits, labels, reduction="sum") / ok.sum().clamp(min=1) * lw["semantic_loss"]
    z = starts
    trunc = cfg["trunc"]
    front = valid & (z < (depth_gt - trunc))
    back = valid & (z > (depth_gt + trunc))
    near = valid & (~front) & (~back)
    out["free_space_loss"] = torch.sum(F.relu(trunc - sdf) * front) / torch.clamp(torch.sum(front), min=1.0) * lw["free_space_loss"]
    out["sdf_loss"] = torch.sum(torch.abs(z + sdf - depth_gt) * near) / torch.clamp(torch.sum(near), min=1.0) * lw["sdf_loss"]
    out["eikonal_loss"] = torch.mean((grad.norm(2, dim=-1) - 1) ** 2) * lw["eikonal_loss"]
    return out


@pytest.mark.parametrize("B,R,S,c_sem,hidden", [(2, 96, 33, 48, 40), (1, 257, 20, 512, 128)])
def test_ray_loss_node_equals_the_torch_statement(device, B, R, S, c_sem, hidden):
    from ponderv2_amd import ray_epilogue
    from ponderv2_amd.ponder.utils.config import ConfigDict

    torch.manual_seed(3)
    n_f2, n_geo, nv = 64, 64, 140
    if R % B:
        R -= R % B
    comp32 = torch.randn(R, nv, device=device) * 0.5
    comp32[:, n_f2 + n_geo + 10] = torch.rand(R, device=device) * 0.9 + 0.05    # sum w
    comp32[:, n_f2 + n_geo + 9] = torch.rand(R, device=device) * 1.5            # sum w t
    comp32[3, n_f2 + n_geo + 10] = 0.0                                          # an empty ray
    comp32[3, n_f2 + n_geo + 9] = 0.0
    starts = torch.sort(torch.rand(R, S, device=device) * 2.0 + 0.1, dim=1).values
    sdf32 = torch.randn(R, S, device=device) * 0.1
    grad32 = torch.randn(R, S, 3, device=device)
    depth_gt = torch.rand(R, 1, device=device) * 2.0
    depth_gt[::7] = 0.0
    sem_gt = torch.randn(R, c_sem, device=device)
    sem_gt = sem_gt / sem_gt.norm(dim=-1, keepdim=True)
    sem_gt[1::5] = 0.0                                                          # class 0: no target (Q3)
    targets = dict(depth=depth_gt, rgb=torch.rand(R, 3, device=device), semantic=sem_gt)
    weights = dict(depth_loss=1.0, rgb_loss=10.0, semantic_loss=0.1, free_space_loss=1.0, sdf_loss=10.0,
                   eikonal_loss=0.01)
    cfg = dict(weights=weights, temperature=0.07, trunc=0.05)
    bg = (0.1, 0.2, 0.3)
    head32 = _Head(3 + n_f2 + n_geo, hidden, c_sem).to(device)

    # float64 statement
    head64 = _Head(3 + n_f2 + n_geo, hidden, c_sem).to(device).double()
    head64.load_state_dict({k: v.double() for k, v in head32.state_dict().items()})
    leaves64 = [t.double().clone().requires_grad_(True) for t in (comp32, sdf32, grad32)]
    ref = _statement(leaves64[0], leaves64[1], leaves64[2], starts.double(), head64,
                     {k: v.double() for k, v in targets.items()}, cfg, B, bg, n_f2, n_geo)
    ref_total = sum(v for k, v in ref.items() if "loss" in k)
    ref_total.backward()

    # the node
    leaves = [t.clone().requires_grad_(True) for t in (comp32, sdf32, grad32)]
    fused = dict(comp=leaves[0], starts=starts, sdf=leaves[1], grad=leaves[2], semantic=head32, n_f2=n_f2,
                 n_geo=n_geo, num_scenes=B, background=bg)
    preds = ray_epilogue.RenderOutputs({}, fused, lambda: {})
    loss_cfg = ConfigDict(dict(sensor_depth_truncation=cfg["trunc"], temperature=cfg["temperature"],
                               weights=weights))
    assert ray_epilogue.usable(preds, targets, loss_cfg)
    before = ray_epilogue.CALLS
    got = ray_epilogue.ray_losses(preds, targets, loss_cfg)
    assert ray_epilogue.CALLS == before + 1
    assert list(got) == list(ref)          # same keys, same order
    got.total.backward()
    for k in ref:
        a, b = float(got[k]), float(ref[k])
        assert abs(a - b) <= 3e-6 * (1 + abs(b)), (k, a, b)
    assert abs(float(got.total) - float(ref_total)) <= 3e-6 * (1 + abs(float(ref_total)))
    for name, a, b in zip(("comp", "sdf", "grad"), leaves, leaves64):
        err = float((a.grad.double() - b.grad).abs().max())
        assert err <= 2e-5 * float(b.grad.abs().max()) + 1e-9, (name, err, float(b.grad.abs().max()))
    for (n, p), (_, q) in zip(head32.named_parameters(), head64.named_parameters()):
        if n.startswith("fc_p"):
            assert p.grad is not None and float(p.grad.abs().max()) == 0.0, n   # an exact zero gradient
            continue
        err = float((p.grad.double() - q.grad).abs().max())
        assert err <= 2e-5 * float(q.grad.abs().max()) + 1e-9, (n, err)

    # two runs: identical bits (every sum in a fixed order)
    leaves2 = [t.clone().requires_grad_(True) for t in (comp32, sdf32, grad32)]
    fused2 = dict(fused, comp=leaves2[0], sdf=leaves2[1], grad=leaves2[2])
    got2 = ray_epilogue.ray_losses(ray_epilogue.RenderOutputs({}, fused2, lambda: {}), targets, loss_cfg)
    got2.total.backward()
    assert torch.equal(got2.total, got.total)
    assert torch.equal(leaves2[0].grad, leaves[0].grad)
