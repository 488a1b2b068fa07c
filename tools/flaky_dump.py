import sys, os, json
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import golden_cases as gc
from oracle.detweights import fill_deterministic, formula_tensor
from ponderv2_amd.ponder.models import build_model
from ponderv2_amd.ponder.utils.config import ConfigDict
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
res = {k: float(p.grad.double().abs().sum()) for k, p in model.named_parameters() if p.grad is not None}
res["__dfeat"] = float(feat.grad.double().abs().sum()); res["__out"] = float(out.double().abs().sum())
print(json.dumps(res))
