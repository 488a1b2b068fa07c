"""GPU parity: the HIP-backed product models against the reference-generated golden vectors."""
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu


def test_spunet_gpu_vs_reference_golden(device):
    errs = gc.run_spunet(device, torch.float32)
    print(errs)
    assert errs["out"] < 1e-4 and errs["dfeat"] < 1e-3, errs
    assert max(errs.values()) < 2e-3, errs


def test_neus_head_gpu_vs_reference_golden(device):
    errs = gc.run_neus(device)
    print(errs)
    outs = {k: v for k, v in errs.items() if k.startswith(("out_", "loss_"))}
    assert max(outs.values()) < 1e-4, errs
    assert max(errs.values()) < 2e-3, errs


def test_ponder_indoor_gpu_vs_reference_golden(device):
    errs = gc.run_ponder_indoor(device)
    print(errs)
    losses = {k: v for k, v in errs.items() if not k.startswith("grad_")}
    assert max(losses.values()) < 1e-3, errs
    assert max(errs.values()) < 2e-2, errs
