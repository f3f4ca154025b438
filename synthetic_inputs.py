"""Seeded synthetic INPUTS (no network, no datasets, no checkpoints here): noise tiles, training targets in the reference's format, blob-like
probability maps and deterministic weights for any module with the reference's key set.  Neither product nor oracle: bench.py's measured legs
take their inputs from here (they may not import oracle/), the oracle modules re-export the same functions so that tests, smoke() and the
CPU baseline see identical data."""
import numpy as np
import torch
from scipy import ndimage as ndi

_SEEDED = {}


def seeded_state_dict(module, seed=1234):
    """Deterministic, torch-RNG-independent weights for any module with the reference's key set.

    Values depend only on (key, shape, seed): every tensor is drawn from its own
    numpy Generator seeded with (seed, crc32(key)), so aliasing / key order cannot change them.
    Conv / deconv weights ~ N(0, 2/fan_in); biases and BN beta ~ N(0, .05); BN gamma ~ U(.8,1.2)
    (U(.1,.3) on the last BN of every residual branch so activations stay O(1) in eval mode with
    un-calibrated running stats); running_mean ~ N(0,.1), running_var ~ U(.8,1.2).
    """
    import zlib
    sd = module.state_dict()
    bottleneck = any('.bn3.' in k for k in sd)
    last_bn = '.bn3.' if bottleneck else '.bn2.'
    out = {}
    for key, t in sd.items():
        shape = tuple(t.shape)
        memo = (seed, key, shape, t.dtype, bottleneck)      # the draw is a function of exactly these: drawn once per process, handed out as copies
        if memo in _SEEDED:
            out[key] = _SEEDED[memo].clone()
            continue
        rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
        if key.endswith('num_batches_tracked'):
            out[key] = torch.zeros(shape, dtype=t.dtype)
            continue
        leaf = key.rsplit('.', 1)[-1]
        if t.dim() == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            if 'block.1' in key:                       # ConvTranspose2d weight is [Cin, Cout, kh, kw]
                fan_in = shape[0] * shape[2] * shape[3] / 4.0
            v = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        elif t.dim() == 2:
            v = rng.standard_normal(shape) * 0.01
        elif leaf == 'running_var':
            v = rng.uniform(0.8, 1.2, shape)
        elif leaf == 'running_mean':
            v = rng.standard_normal(shape) * 0.1
        elif leaf == 'weight':                         # BN gamma; small on the residual branch's last BN
            v = rng.uniform(0.1, 0.3, shape) if last_bn in key else rng.uniform(0.8, 1.2, shape)
        else:                                          # BN beta / conv bias / fc bias
            v = rng.standard_normal(shape) * 0.05
        _SEEDED[memo] = torch.from_numpy(np.asarray(v, dtype=np.float32))
        out[key] = _SEEDED[memo].clone()
    return out



def synthetic_batch(n, h, w, seed=1234):
    """Normalised network input f32[n,3,h,w] from uint8 noise tiles: uniform 0..255,
    /255, minus MEAN over STD (src/pipeline_config.py:19-20)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(n, 3, h, w), dtype=np.uint8).astype(np.float32) / 255.0
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    return torch.from_numpy((img - mean) / std)


def synthetic_target(n, h, w, seed=1234):
    """Seeded f32[n,3,h,w] training target in the reference's format: ch0 mask
    {0,1}, ch1 distance map (sum of the two nearest building distances, cast to uint8 as `to_pil`
    does, src/utils.py:284-285), ch2 sqrt(component size) (uint8 as well)."""
    import numpy as np
    from scipy import ndimage as ndi
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 3, h, w), np.float32)
    for i in range(n):
        z = ndi.gaussian_filter(rng.standard_normal((h, w)), 5.0, mode='wrap')
        mask = z > np.quantile(z, 0.75)
        lab, k = ndi.label(mask)
        if k >= 2:
            dists = np.stack([ndi.distance_transform_edt(lab != j) for j in range(1, k + 1)], -1)
            dists.sort(-1)
            dist = dists[..., 0] + dists[..., 1]
        else:
            dist = np.zeros((h, w))
        sizes = np.zeros((h, w))
        for j in range(1, k + 1):
            sizes[lab == j] = np.sqrt((lab == j).sum())
        out[i, 0] = mask
        out[i, 1] = (dist * ~mask).astype(np.uint16).astype(np.uint8)
        out[i, 2] = sizes.astype(np.uint8)
    return torch.from_numpy(out)


def synthetic_probs(n, h, w, seed=1234, smooth=4.0):
    """Seeded f32[n,2,h,w] softmax maps with blob structure (~20-60 components per image):
    softmax of low-pass filtered Gaussian noise."""
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n, 2, h, w)).astype(np.float32)
    z = ndi.gaussian_filter(z, sigma=(0, 0, smooth, smooth), mode='wrap') * np.float32(8.0 * smooth)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
