"""CPU restatement of the compositing step (checker only): straight transcription of
ponder/models/ponder/render_utils/rays.py:83-105 (weights / transmittance from alphas) and
renderers.py:5-75 (weighted sums) in torch ops, plus the closed-form gradients the HIP kernels
implement, so both the values and the hand-derived backward formulas can be checked on the host."""
import torch


def weights_from_alphas(alphas):
    """alphas (R, S, 1) -> weights (R, S, 1), transmittance (R, S + 1, 1); rays.py:95-104."""
    ones = torch.ones((alphas.shape[0], 1, 1), dtype=alphas.dtype, device=alphas.device)
    transmittance = torch.cumprod(torch.cat([ones, 1.0 - alphas + 1e-7], dim=1), dim=1)
    return alphas * transmittance[:, :-1, :], transmittance


def weighted_sum(weights, values):
    """renderers.py:16,27,39,44: sum over the sample axis of weights * values."""
    return torch.sum(weights * values, dim=-2)


def grad_alpha_closed_form(alphas, grad_weights):
    """d alpha_s = gw_s T_s - (sum_{k>s} gw_k w_k) / (1 - alpha_s + 1e-7)   (csrc/raymarch.hip)."""
    w, t = weights_from_alphas(alphas)
    q = grad_weights * w
    after = torch.flip(torch.cumsum(torch.flip(q, dims=[1]), dim=1), dims=[1]) - q
    return grad_weights * t[:, :-1, :] - after / (1.0 - alphas + 1e-7)


def weighted_sum_grads_closed_form(weights, values, grad_out):
    gw = (values * grad_out[:, None, :]).sum(-1, keepdim=True)
    gx = weights * grad_out[:, None, :]
    return gw, gx
