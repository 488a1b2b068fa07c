"""torch-CPU restatement of the sparse conv arithmetic on a rulebook (differentiable by autograd).

out[pair_out[p]] += W[k(p)] @ in[pair_in[p]]   with W in spconv layout [Cout, K, Cin]
(reference call sites: ponder/models/sparse_unet/spconv_unet_v1m1_base.py:41,47,58,112,135,171).
"""
import torch


def sparse_conv(features, weight_okc, pair_in, pair_out, kstart, n_out):
    """features [N_in, Cin]; weight_okc [Cout, K, Cin]; pair_* int64 tensors; kstart python ints."""
    c_out, K, c_in = weight_okc.shape
    out = features.new_zeros((n_out, c_out))
    for k in range(K):
        a, b = int(kstart[k]), int(kstart[k + 1])
        if a == b:
            continue
        gathered = features.index_select(0, pair_in[a:b])
        out = out.index_add(0, pair_out[a:b], gathered @ weight_okc[:, k, :].t())
    return out
