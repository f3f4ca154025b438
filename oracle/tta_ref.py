"""TEST INFRASTRUCTURE -- numpy oracle for test-time augmentation (reference: src/loaders.py:401-517).

  tta_specs            TestTimeAugmentationGenerator._get_tta_data (:415-435): identity first, then the product of the
                       enabled options minus the all-off combination (colour shift needs imgaug: not restated)
  transform            test_time_augmentation_transform (:470-480): ud-flip, ELSE lr-flip (the reference's elif chain never
                       combines them), then rotation by a multiple of 90 degrees (skimage.rotate: counter-clockwise)
  inverse_transform    test_time_augmentation_inverse_transform (:483-491)
  aggregate            aggregate_augmentations + agg_method (:437-467): mean / max / min / scipy gmean over the variants
Pinned against the reference functions (through the skimage shim, where rotate(angle=90k) == np.rot90) in
tests/test_oracle.py.
"""
from itertools import product

import numpy as np


def tta_specs(flip_ud=False, flip_lr=False, rotation=False):
    specs = [{'ud_flip': False, 'lr_flip': False, 'rotation': 0, 'color_shift': False}]
    for ud, lr, rot in product([True, False] if flip_ud else [False], [True, False] if flip_lr else [False],
                               [0, 90, 180, 270] if rotation else [0]):
        if not ud and not lr and rot == 0:
            continue
        specs.append({'ud_flip': ud, 'lr_flip': lr, 'rotation': rot, 'color_shift': False})
    return specs


def transform(image, spec):
    """image [..., H, W] (square when rotating by 90/270)"""
    if spec['ud_flip']:
        image = image[..., ::-1, :]
    elif spec['lr_flip']:
        image = image[..., :, ::-1]
    return np.rot90(image, spec['rotation'] // 90, axes=(-2, -1))


def inverse_transform(image, spec):
    image = np.rot90(image, -(spec['rotation'] // 90), axes=(-2, -1))
    if spec['ud_flip']:
        image = image[..., ::-1, :]
    elif spec['lr_flip']:
        image = image[..., :, ::-1]
    return image


def aggregate(preds, specs, method='gmean'):
    """preds: [V][C,H,W] predictions of the transformed inputs -> [C,H,W]"""
    stack = np.stack([inverse_transform(p, s) for p, s in zip(preds, specs)], axis=-1)
    if method == 'mean':
        return stack.mean(-1)
    if method == 'max':
        return stack.max(-1)
    if method == 'min':
        return stack.min(-1)
    if method == 'gmean':
        with np.errstate(divide='ignore'):
            return np.exp(np.log(stack).mean(-1))
    raise KeyError(method)
