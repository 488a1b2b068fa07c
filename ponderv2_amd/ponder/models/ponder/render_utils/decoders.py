"""Point-feature MLP heads of the SDF field (ponder/models/ponder/render_utils/decoders.py:
SDFDecoder :6-36 Softplus(beta=100), RGBDecoder :39-76 ReLU + sigmoid, SemanticDecoder :79-109
ReLU).  All three share one skeleton: x = fc_p(p) * points_factor; per layer l:
x = lin_l(x + fc_c[l](feat)), activation on all but the last layer."""
import torch
import torch.nn as nn

from ponderv2_amd.linear import linear


def _lin(layer, x):
    """nn.Linear parameters, MFMA GEMM kernels (ponderv2_amd.linear) for big device batches."""
    return linear(x, layer.weight, layer.bias)


class _ConditionedMLP(nn.Module):
    def __init__(self, in_dim, out_dim, hidden_size, n_blocks, points_factor, activation,
                 out_activation=None, fc_p_first=False):
        super().__init__()
        dims = [hidden_size] * (n_blocks + 1) + [out_dim]
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            setattr(self, f"lin{l}", nn.Linear(dims[l], dims[l + 1]))
        # construction order = the reference's (decoders.py:22-25 fc_c then fc_p in SDFDecoder, :59-63
        # and :99-103 fc_p then fc_c in the other two): the same seed then draws the same initial weights
        # and state_dict() lists the keys in the same order
        if fc_p_first:
            self.fc_p = nn.Linear(3, hidden_size)
        self.fc_c = nn.ModuleList(nn.Linear(in_dim, hidden_size) for _ in range(self.num_layers - 1))
        if not fc_p_first:
            self.fc_p = nn.Linear(3, hidden_size)
        self.activation = activation
        self.out_activation = out_activation
        self.points_factor = points_factor

    def hidden(self, points, point_feats):
        """Input of the LAST linear layer.  When the head has no output activation the last layer
        commutes with alpha compositing: sum_k w_k lin(h_k) = lin.weight (sum_k w_k h_k) +
        lin.bias sum_k w_k, so callers may composite ``hidden`` and apply ``last_linear`` per ray
        instead of per sample (removes the (R*S, 512) semantic activations)."""
        if self.points_factor == 0.0:
            # the positional branch contributes exactly zero; keep fc_p in the graph with an exact
            # zero gradient (the reference's fc_p(points) * 0.0 does the same, so weight decay
            # still acts on it) without materialising an (M, hidden) tensor
            x = (self.fc_p.weight.sum() + self.fc_p.bias.sum()) * 0.0
        else:
            x = _lin(self.fc_p, points) * self.points_factor
        last = self.num_layers - 2
        for l in range(last):
            x = self.activation(_lin(getattr(self, f"lin{l}"), x + _lin(self.fc_c[l], point_feats)))
        return x + _lin(self.fc_c[last], point_feats)

    @property
    def last_linear(self):
        return getattr(self, f"lin{self.num_layers - 2}")

    def forward(self, points, point_feats):
        x = _lin(self.last_linear, self.hidden(points, point_feats))
        return x if self.out_activation is None else self.out_activation(x)


class SDFDecoder(_ConditionedMLP):
    def __init__(self, in_dim, out_dim, hidden_size=256, n_blocks=5, points_factor=1.0, **kwargs):
        super().__init__(in_dim, out_dim, hidden_size, n_blocks, points_factor,
                         nn.Softplus(beta=100))


class RGBDecoder(_ConditionedMLP):
    def __init__(self, in_dim, out_dim=3, hidden_size=256, n_blocks=5, points_factor=1.0, **kwargs):
        super().__init__(in_dim, out_dim, hidden_size, n_blocks, points_factor, nn.ReLU(),
                         out_activation=torch.sigmoid, fc_p_first=True)


class SemanticDecoder(_ConditionedMLP):
    def __init__(self, in_dim, out_dim, hidden_size=256, n_blocks=5, points_factor=1.0, **kwargs):
        super().__init__(in_dim, out_dim, hidden_size, n_blocks, points_factor, nn.ReLU(),
                         fc_p_first=True)
