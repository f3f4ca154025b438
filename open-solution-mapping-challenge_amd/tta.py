"""Test-time augmentation on the device (reference: src/loaders.py:401-517).

The reference builds the augmented copies on the host (PIL/skimage per image, a ThreadPool for the aggregation).
Here the variants of a batch are index permutations of the normalised network input already in HBM
(`msc_tta_transform`), the network runs on V*N images, and one kernel un-permutes and aggregates the V
probability maps per image (`msc_tta_aggregate`): flips / quarter turns commute with the per-pixel normalisation.
Colour-shift variants (imgaug, host side) are not on the device path.
"""
from itertools import product

import numpy as np
import torch

from . import _lib

METHODS = {'mean': 0, 'gmean': 1, 'max': 2, 'min': 3}


def tta_specs(flip_ud=False, flip_lr=False, rotation=False, color_shift_runs=False):
    """TestTimeAugmentationGenerator._get_tta_data (src/loaders.py:415-435): identity first, then every other combination."""
    if color_shift_runs:
        raise NotImplementedError('colour-shift TTA is a host-side imgaug augmentation; not available on the device path')
    specs = [{'ud_flip': False, 'lr_flip': False, 'rotation': 0, 'color_shift': False}]
    for ud, lr, rot in product([True, False] if flip_ud else [False], [True, False] if flip_lr else [False],
                               [0, 90, 180, 270] if rotation else [0]):
        if not ud and not lr and rot == 0:
            continue
        specs.append({'ud_flip': ud, 'lr_flip': lr, 'rotation': rot, 'color_shift': False})
    return specs


def encode(specs):
    codes = [int(bool(s['ud_flip'])) | (int(bool(s['lr_flip'])) << 1) | (((s['rotation'] // 90) & 3) << 2) for s in specs]
    return np.asarray(codes, np.int32), any(s['rotation'] % 180 for s in specs)


def transform_batch(x, specs):
    """x cuda f32 [N,C,H,W] -> cuda f32 [V*N,C,H,W], variant-major"""
    codes, quarter = encode(specs)
    N, Cc, H, W = x.shape
    x = x.contiguous().float()
    out = torch.empty((len(codes) * N, Cc, H, W), dtype=torch.float32, device=x.device)
    dc = torch.from_numpy(codes).to(x.device)
    _lib.call('msc_tta_transform', x.data_ptr(), out.data_ptr(), dc.data_ptr(), N, Cc, H, W, len(codes), int(quarter),
              torch.cuda.current_stream(x.device).cuda_stream)
    return out


def aggregate_batch(preds, specs, method='gmean'):
    """preds cuda f32 [V*N,C,H,W] (variant-major, as transform_batch orders them) -> cuda f32 [N,C,H,W]"""
    codes, quarter = encode(specs)
    V = len(codes)
    VN, Cc, H, W = preds.shape
    N = VN // V
    preds = preds.contiguous().float()
    out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=preds.device)
    dc = torch.from_numpy(codes).to(preds.device)
    _lib.call('msc_tta_aggregate', preds.data_ptr(), out.data_ptr(), dc.data_ptr(), N, Cc, H, W, V, METHODS[method], int(quarter),
              torch.cuda.current_stream(preds.device).cuda_stream)
    return out


def predict_tta(model, x, specs, method='gmean'):
    """probabilities of `model` (UNetResNet) for batch x aggregated over the TTA variants: cuda f32 [N,2,H,W]"""
    N = x.shape[0]
    xt = transform_batch(x, specs)
    # V forward passes of the batch size the program is tuned for; each variant's probabilities go straight into their slice of ONE
    # buffer (the program's own output buffer is overwritten by the next pass): one copy per variant, no list / cat of clones
    probs = torch.empty((len(specs) * N, 2) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    for v in range(len(specs)):
        probs[v * N:(v + 1) * N].copy_(model.predict_proba(xt[v * N:(v + 1) * N]))
    return aggregate_batch(probs, specs, method)
