import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc
from oracle.detweights import fill_deterministic, formula_tensor
from ponderv2_amd.ponder.models import build_model
from ponderv2_amd.ponder.utils.config import ConfigDict

def run():
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(gc.GOLDEN, "spunet_small.npz"))
    coords = g["coords"]; counts = np.bincount(coords[:, 0])
    model = build_model(ConfigDict(gc.SMALL_BACKBONE)); fill_deterministic(model); model = model.to(dev).train()
    n = len(coords)
    feat = formula_tensor("spunet.feat", (n, 6), 1.0).to(dev).requires_grad_(True)
    out = model(dict(grid_coord=torch.from_numpy(coords[:, 1:].astype(np.int64)).to(dev), feat=feat,
                     offset=torch.from_numpy(np.cumsum(counts)).long().to(dev)))
    probe = formula_tensor("spunet.probe", tuple(out.shape), 1.0).to(dev)
    (out * probe).sum().backward()
    return {k: p.grad.detach().cpu().double() for k, p in model.named_parameters() if p.grad is not None}

a = run()
torch.save(a, "/tmp/grads_%s.pt" % os.environ.get("PV2_SPCONV_GENERIC", "0"))
if os.path.exists("/tmp/grads_1.pt") and os.path.exists("/tmp/grads_0.pt"):
    g1, g0 = torch.load("/tmp/grads_1.pt"), torch.load("/tmp/grads_0.pt")
    for k in g1:
        e = (g1[k] - g0[k]).abs().max().item() / (g1[k].abs().max().item() + 1e-30)
        if e > 1e-4 and "weight" in k and g1[k].dim() > 1:
            print("%-40s %s rel diff %.2e" % (k, tuple(g1[k].shape), e))
errs, _ = gc.run_spunet(torch.device("cuda:0"), torch.float32)
print("GENERIC=%s vs golden:" % os.environ.get("PV2_SPCONV_GENERIC", "0"), {k: float("%.2e" % v) for k, v in errs.items()})
