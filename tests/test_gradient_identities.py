"""The per-channel gradient identities of the fused blocks' training pass against autograd on the oracle's
ConvNeXt block (oracle/models_ref.py::CNBlock = models/convnext.py:15-50): in exact (fp64) arithmetic they ARE the gradients.

    d(gamma)[c] = (sum_j W2[c,j] dW2[c,j] + b2[c] d(b2)[c]) / gamma[c]
    d(ln_b)[c]  = sum_j W1[j,c] d(b1)[j]
    d(ln_w)[c]  = (sum_j W1[j,c] dW1[j,c] - ln_b[c] d(ln_b)[c]) / ln_w[c]

The HIP kernels (cnx_block_dgamma / cnx_block_dln) are checked against the direct sums in tests/test_gpu_model_ops.py; this file pins
the algebra itself on the CPU, where the driver's CPU tier runs."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import models_ref  # noqa: E402  (test infrastructure: the oracle is the checker here)


@pytest.mark.parametrize("dim,hw,fb", [(8, 6, False), (16, 5, True), (24, 7, False)])
def test_layer_scale_and_layernorm_gradients_follow_from_the_linear_layers_weight_gradients(dim, hw, fb):
    torch.manual_seed(dim + hw)
    blk = models_ref.CNBlock(dim, ls_init=0.3, fb_names=fb).double()
    _, norm, fc1, fc2 = blk.parts()
    with torch.no_grad():                                   # generic values: LayerNorm affine and gamma away from their initial constants
        norm.weight.copy_(1 + 0.3 * torch.randn(dim, dtype=torch.float64))
        norm.bias.copy_(0.2 * torch.randn(dim, dtype=torch.float64))
        blk.gamma.copy_(0.3 * torch.randn(dim, dtype=torch.float64) + 0.5)
    x = torch.randn(3, dim, hw, hw, dtype=torch.float64)
    (blk(x) * torch.randn(3, dim, hw, hw, dtype=torch.float64)).sum().backward()
    W1, W2 = fc1.weight.detach(), fc2.weight.detach()      # [4C, C], [C, 4C]
    dgamma = ((W2 * fc2.weight.grad).sum(1) + fc2.bias.detach() * fc2.bias.grad) / blk.gamma.detach()
    dlb = W1.t() @ fc1.bias.grad
    dlw = ((W1 * fc1.weight.grad).sum(0) - norm.bias.detach() * dlb) / norm.weight.detach()
    for got, ref in ((dgamma, blk.gamma.grad), (dlb, norm.bias.grad), (dlw, norm.weight.grad)):
        assert float((got - ref).abs().max()) <= 1e-10 * float(ref.abs().max() + 1)
