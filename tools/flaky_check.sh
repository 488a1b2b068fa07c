#!/bin/bash
# run the SpUNet golden check N times in fresh processes; print max error each time
for i in $(seq 1 ${1:-8}); do
  python - <<'PY' 2>/dev/null
import sys, os; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch, golden_cases as gc
e, _ = gc.run_spunet(torch.device('cuda:0'), torch.float32)
print("max err %.2e" % max(e.values()), {k: float('%.1e' % v) for k, v in e.items() if v > 1e-4})
PY
done
