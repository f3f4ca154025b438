"""TEST INFRASTRUCTURE -- torch-CPU fp32 oracle for the losses and the optimizer step.

Restates src/steps/pytorch/validation.py:8-28 (DiceLoss, multiclass_segmentation_loss) and
src/models.py:310-454 (distance x size weighted cross entropy, soft Dice on softmax, the mix), plus
the Adam+L2 update the reference configures at src/models.py:57,287-292 (torch.optim.Adam with
weight_decay).  Pinned against the literal reference functions in
tests/test_oracle.py and tests/golden/loss_*.npz.
"""
import math

import torch
import torch.nn.functional as F


def segmentation_ce(output, target):
    # validation.py:25-28
    return F.cross_entropy(output, target.squeeze(1).long())


def loss_weights(weight_target, w0, sigma, imsize):
    # src/models.py:339-381.  weight_target = target[:, 1:] : ch0 distances, ch1 sqrt(sizes)
    d = weight_target[:, 0]
    s = weight_target[:, 1]
    c = math.sqrt(imsize[0] * imsize[1]) / 2
    dist_w = 1.0 + w0 * torch.exp(-(d ** 2) / (sigma ** 2))
    dist_w = torch.where(d == 0, torch.ones_like(dist_w), dist_w)
    s1 = torch.where(s == 0, torch.ones_like(s), s)
    size_w = torch.where(s1 == 1, torch.ones_like(s1), torch.tensor(c, dtype=s.dtype) / s1)
    return dist_w * size_w


def weighted_ce(output, target, w0, sigma, imsize):
    # src/models.py:310-336
    w = loss_weights(target[:, 1:], w0, sigma, imsize)
    per_pixel = F.cross_entropy(output, target[:, 0].long(), reduction='none')
    return torch.mean(per_pixel * w)


def dice(output, target_cls, smooth=0.0, eps=1e-7, excluded=(0,), activation='softmax'):
    # src/models.py:421-454 with validation.py:8-16; softmax over channels (or elementwise sigmoid, :437-442), sums over the WHOLE batch
    if activation not in ('softmax', 'sigmoid'):
        raise NotImplementedError('only sigmoid and softmax are implemented')
    p = torch.softmax(output, dim=1) if activation == 'softmax' else torch.sigmoid(output)
    loss = 0
    for c in range(output.shape[1]):
        if c in excluded:
            continue
        t = (target_cls == c).float()
        pc = p[:, c]
        loss = loss + (1 - (2 * torch.sum(pc * t) + smooth) / (torch.sum(pc) + torch.sum(t) + smooth + eps))
    return loss


def mixed_dice_ce(output, target, dice_weight=0.2, ce_weight=1.0, smooth=1.0, w0=50.0, sigma=10.0,
                  imsize=(256, 256), dice_activation='softmax'):
    # src/models.py:384-418 as configured by PyTorchUNetWeighted (:149-161) and neptune.yaml:42-57
    return dice_weight * dice(output, target[:, 0].long(), smooth, activation=dice_activation) + \
        ce_weight * weighted_ce(output, target, w0, sigma, imsize)


def adam_l2_step(p, g, m, v, step, lr=5e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-4):
    """One torch.optim.Adam update with L2 folded into the gradient (in place on p, m, v)."""
    g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p


from synthetic_inputs import synthetic_target      # noqa: E402,F401  (input generator, shared with bench.py)
