import os, sys, copy, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc
from ponderv2_amd.ponder.datasets import collate_fn, make_scene
from ponderv2_amd.ponder.models import build_model
from ponderv2_amd.ponder.utils.config import ConfigDict
from ponderv2_amd.ponder.models.ponder.graphed_render import GraphedRenderHead

dev = torch.device("cuda:0")
cfg = gc.indoor_model_cfg(dict(gc.SMALL_BACKBONE, channels=(16, 32, 48, 64, 64, 48, 32, 96)), grid_shape=(32, 32, 8), ray_nsample=24)
cfg["graph_render_head"] = False
model = build_model(ConfigDict(cfg)).to(dev).train()
kw = dict(n_raw=16000, num_views=2, image_hw=(48, 64))
b = collate_fn([make_scene(200, **kw), make_scene(201, **kw)])
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
d = model.extract_feature(b)
ray_dict, d = model.prepare_ray(d)
vol = model.prepare_volume(d)[0].detach()
head = GraphedRenderHead(model)
torch.autograd.set_detect_anomaly(True, check_nan=False)
try:
    head._capture(vol, ray_dict)
    print("capture OK")
    torch.cuda.synchronize()
    outs = head._run(vol, ray_dict); torch.cuda.synchronize()
    print("replay OK", [float(o) for o in outs])
except Exception:
    traceback.print_exc()
