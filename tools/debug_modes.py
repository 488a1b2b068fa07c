#!/usr/bin/env python
"""Which kernel-selection switch moves a golden comparison?  Runs a golden case under the default
selection and with switches flipped, each in a FRESH process (a corrupting mode must not take the
next one down), printing the gradient errors."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = [("default", {}),
         ("convbn off", {"USE_CONVBN": False}),
         ("convbn off, side off", {"USE_CONVBN": False, "side": False}),
         ("convbn off, wgrad atomics", {"USE_CONVBN": False, "USE_WGRAD_DET": False}),
         ("convbn off, pr off", {"USE_CONVBN": False, "USE_PR": False}),
         ("convbn off, pr subm", {"USE_CONVBN": False, "USE_PR": "subm"}),
         ("wgrad atomics", {"USE_WGRAD_DET": False}),
         ("round-2 atomics", {"USE_PR": False, "USE_WGRAD_DET": False, "USE_CONVBN": False, "USE_OS": False})]

if len(sys.argv) > 2 and sys.argv[1] == "--one":
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_cases as gc
    from ponderv2_amd import kernels as K, sidestream
    name, over = MODES[int(sys.argv[2])]
    case = getattr(gc, sys.argv[3])
    for k, v in over.items():
        if k != "side":
            setattr(K, k, v)
    sidestream.ENABLED = over.get("side", True)
    for rep in range(2):
        res = case(torch.device("cuda:0"))
        errs = res[0] if isinstance(res, tuple) else res
        show = {k.replace("grad_", "g_")[:40]: float("%.3g" % v) for k, v in errs.items()
                if k.startswith("grad_") or k == "loss"}
        print("%-28s run %d %s" % (name, rep, show), flush=True)
else:
    case = sys.argv[1] if len(sys.argv) > 1 else "run_ponder_indoor"
    for i in range(len(MODES)):
        r = subprocess.run([sys.executable, __file__, "--one", str(i), case], capture_output=True, text=True)
        out = [l for l in r.stdout.splitlines() if "run " in l]
        print("\n".join(out) if out else "%-28s FAILED rc=%d %s" % (MODES[i][0], r.returncode, r.stderr[-300:].replace("\n", " | ")), flush=True)
