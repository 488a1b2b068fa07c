import sys, os, json
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import golden_cases as gc
from oracle.detweights import fill_deterministic, formula_tensor
from ponderv2_amd import kernels as K, rownorm
from ponderv2_amd.ponder.models import build_model
from ponderv2_amd.ponder.utils.config import ConfigDict
log = []
def cs(t): return None if t is None else float(t.double().abs().sum())
o_fwd, o_wg = K.spconv_forward, K.spconv_backward_weight
def fwd(feats, w, rb, out=None):
    r = o_fwd(feats, w, rb, out); log.append(("conv", tuple(w.shape), rb.n_in, rb.n_out, cs(feats), cs(w), cs(r))); return r
def wg(feats, g, rb, c_out, tile=None):
    r = o_wg(feats, g, rb, c_out, tile); log.append(("wgrad", c_out, cs(feats), cs(g), cs(r))); return r
K.spconv_forward, K.spconv_backward_weight = fwd, wg
o_bnb = rownorm._FusedBNFunction.backward
def bnb(ctx, dy):
    r = o_bnb(ctx, dy); log.append(("bn_bwd", tuple(dy.shape), cs(dy), cs(r[0]), cs(r[3]))); return r
rownorm._FusedBNFunction.backward = staticmethod(bnb)
dev = torch.device("cuda:0")
g = np.load(os.path.join(gc.GOLDEN, "spunet_small.npz"))
coords = g["coords"]; counts = np.bincount(coords[:, 0])
model = build_model(ConfigDict(gc.SMALL_BACKBONE)); fill_deterministic(model); model = model.to(dev).train()
n = len(coords)
feat = formula_tensor("spunet.feat", (n, 6), 1.0).to(dev).requires_grad_(True)
out = model(dict(grid_coord=torch.from_numpy(coords[:, 1:].astype(np.int64)).to(dev), feat=feat,
                 offset=torch.from_numpy(np.cumsum(counts)).long().to(dev)))
nfwd = len(log)
probe = formula_tensor("spunet.probe", tuple(out.shape), 1.0).to(dev)
(out * probe).sum().backward()
print(json.dumps(dict(nfwd=nfwd, dfeat=cs(feat.grad), log=log)))
