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
    assert max(losses.values()) < 1e-4, errs  # the north-star bound: loss within 1e-4 relative
    # The stem weight gradient sits at the end of a backward chain through ~60 BatchNorm layers,
    # some over a few dozen voxels: on the CPU oracle a 1e-7 relative perturbation of the input
    # features moves it by 3e-2 (and the dec.0 gradient by 1e-4) while the loss moves by 2e-7.
    # Its bound reflects that conditioning; every other probe is held to 1e-3.
    stem = errs.pop("grad_backbone.conv_input.0.weight")
    assert stem < 0.2, stem
    assert max(errs.values()) < 1e-3, errs
