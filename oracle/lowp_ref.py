"""TEST INFRASTRUCTURE -- the torch-CPU oracle (oracle/unet_ref.py) evaluated "at 16-bit storage": arithmetic in the
module's own dtype (float64 for ground truth work), but every tensor the HIP engine keeps in HBM in its 16-bit compute
dtype is rounded to that dtype at the point where the engine stores it -- forward activations AND the gradients that flow
back through them.  Never imported by the product.

Why it exists: the gradient of a 100+-layer ReLU network is discontinuous in its activations (a pre-activation that
crosses zero flips a ReLU mask), so ANY implementation that stores activations with 8 (bf16) or 11 (fp16) mantissa bits
ends up tens of percent away from the fp32 gradient, however exact its kernels are -- the reference's own fp32 path is
already 5e-3 (median over tensors; 1e-2 worst) away from a float64 evaluation at ResNet101 / 128x128 / batch 4.  The parity
statement for the 16-bit modes is therefore "as close to the float64 gradient as the reference's arithmetic is once it
stores what the engine stores", which this module makes measurable (tests/test_gpu_parity_timed.py).

Storage points mirrored (open-solution-mapping-challenge_amd/unet_models.py): the network input (NHWC4 stem image), the
16-bit compute copies of conv / deconv weights (master weights and weight gradients stay fp32), every conv / deconv
output, every BN(+residual)+ReLU output, the BN output of the downsample branch; NOT the logits (fp32 NCHW) and not the
final 1x1's weights (fp32).  Rounding commutes with ReLU and with max-pooling, so rounding a conv output and then its ReLU
again is idempotent.  Gradients: the cotangent of each of those tensors is rounded the same way (the engine stores dout /
dy in the compute dtype); accumulation order differs (the engine rounds after every accumulating writer, here the sum of
all consumers is rounded once).
"""
import types

import torch
import torch.nn.functional as F
from torch import nn

DTYPES = {'bf16': torch.bfloat16, 'fp16': torch.float16}


class _Store(torch.autograd.Function):
    """value and cotangent both pass through the storage dtype"""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


def _compute_copy(w, dtype):
    """the 16-bit compute copy of a master weight; the gradient reaches the master unrounded (fp32 weight gradients)"""
    return w + (w.to(dtype).to(w.dtype) - w).detach()


def install(net, compute_dtype):
    """patch the forward methods of a UNetResNetRef so that it evaluates at `compute_dtype` ('bf16' | 'fp16') storage"""
    dt = DTYPES[compute_dtype]

    def conv_forward(self, x):
        return _Store.apply(F.conv2d(x, _compute_copy(self.weight, dt), self.bias, self.stride, self.padding), dt)

    def deconv_forward(self, x):
        return _Store.apply(F.conv_transpose2d(x, _compute_copy(self.weight, dt), self.bias, self.stride, self.padding), dt)

    def relu_forward(self, x):
        return _Store.apply(torch.relu(x), dt)

    def bn_store_forward(self, x, _orig=nn.BatchNorm2d.forward):
        return _Store.apply(_orig(self, x), dt)

    for name, m in net.named_modules():
        if name == 'final':
            continue
        if isinstance(m, nn.ConvTranspose2d):
            m.forward = types.MethodType(deconv_forward, m)
        elif isinstance(m, nn.Conv2d):
            m.forward = types.MethodType(conv_forward, m)
        elif isinstance(m, nn.ReLU):
            m.forward = types.MethodType(relu_forward, m)
        elif isinstance(m, nn.BatchNorm2d) and name.endswith('.downsample.1'):
            m.forward = types.MethodType(bn_store_forward, m)
    orig_forward = net.forward

    def forward(x):
        return orig_forward(x.to(dt).to(x.dtype))
    net.forward = forward
    return net
