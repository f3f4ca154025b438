"""TEST INFRASTRUCTURE (oracle shim) -- skimage.transform.resize restated for the one call the
hot path makes (src/postprocessing.py:60: resize(image[C,H,W], (C,)+target, mode='constant')).
For a 3-D output whose last dim differs from the input's, skimage takes its n-dimensional
branch: coords = factor*(i+0.5)-0.5 per axis, scipy.ndimage.map_coordinates(order=1,
mode='constant', cval=0), then clips to the input range.  Output is float64 (img_as_float of
the era).  Parity with skimage itself is UNPINNED."""
import numpy as np
from scipy import ndimage as ndi


def resize(image, output_shape, order=1, mode='constant', cval=0, clip=True, preserve_range=False,
           anti_aliasing=None, anti_aliasing_sigma=None):
    image = np.asarray(image)
    output_shape = tuple(output_shape)
    assert len(output_shape) == image.ndim, 'oracle shim: only same-rank resize is on the hot path'
    assert mode == 'constant' and order == 1
    factors = np.asarray(image.shape, dtype=float) / np.asarray(output_shape, dtype=float)
    coord_arrays = [factors[i] * (np.arange(d) + 0.5) - 0.5 for i, d in enumerate(output_shape)]
    coord_map = np.array(np.meshgrid(*coord_arrays, sparse=False, indexing='ij'))
    img = image.astype(np.float64)
    out = ndi.map_coordinates(img, coord_map, order=1, mode='constant', cval=cval)
    if clip:
        lo, hi = min(img.min(), cval), max(img.max(), cval)
        np.clip(out, lo, hi, out=out)
    return out


def rotate(image, angle, resize=False, center=None, order=1, mode='constant', cval=0, clip=True,
           preserve_range=False):
    k = int(round(angle / 90.0))
    assert abs(angle - 90 * k) < 1e-9, 'oracle shim: only multiples of 90 degrees'
    return np.rot90(image, k)
