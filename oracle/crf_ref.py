"""TEST INFRASTRUCTURE -- CPU oracle for `dense_crf` (reference: src/postprocessing.py:183-225).

PARITY UNPINNED.  The reference delegates to pydensecrf (unpinned git HEAD, environment.yml:15),
a wrapper of Kraehenbuehl & Koltun's densecrf C++ (NIPS 2011, arXiv:1210.5644), which is neither
vendored in /root/reference nor installable here, and the reference has no call site or test for
`dense_crf`.  This file restates the published algorithm as densecrf implements it:

    Q <- softmax(-U)
    repeat n times:   Q <- softmax(-U + sum_k w_k * (K_k Q))          (Potts compatibility)
    (K_k Q)_i = n_i * sum_j k(f_i, f_j) * n_j * Q_j ,  n_i = 1/sqrt(sum_j k(f_i,f_j) + 1e-20)
                                                         (NORMALIZE_SYMMETRIC, self term included)
    Gaussian  kernel features  f = (x/sxy, y/sxy)
    bilateral kernel features  f = (x/sxy, y/sxy, r/srgb, g/srgb, b/srgb)
    k(f_i,f_j) = exp(-|f_i-f_j|^2 / 2)

with ONE deliberate difference: densecrf evaluates K Q approximately on a permutohedral lattice;
here it is evaluated EXACTLY over a square window of radius ceil(5*sxy) pixels (both kernels of
the reference use sxy=1, so everything outside the window is < 4e-6 of the centre weight).  The
HIP kernel (msc_dense_crf) implements exactly this windowed definition.
"""
import math

import numpy as np


def window_radius(sxy):
    return int(math.ceil(5.0 * sxy))


def _shift(a, dy, dx):
    """out[..., y, x] = a[..., y+dy, x+dx], zero outside."""
    h, w = a.shape[-2:]
    out = np.zeros_like(a)
    ys0, ys1 = max(0, -dy), min(h, h - dy)
    xs0, xs1 = max(0, -dx), min(w, w - dx)
    if ys0 < ys1 and xs0 < xs1:
        out[..., ys0:ys1, xs0:xs1] = a[..., ys0 + dy:ys1 + dy, xs0 + dx:xs1 + dx]
    return out


def _pair_weights(h, w, kind, sxy, srgb, rgb):
    r = window_radius(sxy)
    offs = [(dy, dx) for dy in range(-r, r + 1) for dx in range(-r, r + 1)]
    ks = np.zeros((len(offs), h, w), dtype=np.float32)
    valid = np.ones((h, w), dtype=np.float32)
    img = None if rgb is None else rgb.astype(np.float32).transpose(2, 0, 1)  # 3,H,W
    for t, (dy, dx) in enumerate(offs):
        inside = _shift(valid, dy, dx)
        d2 = np.float32((dy * dy + dx * dx) / (sxy * sxy))
        if kind == 'bilateral':
            diff = (_shift(img, dy, dx) - img) / np.float32(srgb)
            d2 = d2 + (diff * diff).sum(0)
        ks[t] = np.exp(-0.5 * d2).astype(np.float32) * inside
    norm = (1.0 / np.sqrt(ks.sum(0) + 1e-20)).astype(np.float32)
    return offs, ks, norm


def _exp_and_normalize(x):
    x = x - x.max(0, keepdims=True)
    e = np.exp(x)
    return (e / e.sum(0, keepdims=True)).astype(np.float32)


def mean_field(unary, pairwise, n_iterations):
    """unary: f32[M,H,W] energies; pairwise: list of (kind, sxy, srgb, rgb_u8[H,W,3]|None, compat)."""
    unary = np.asarray(unary, dtype=np.float32)
    m, h, w = unary.shape
    pre = [(_pair_weights(h, w, kind, sxy, srgb, rgb), compat) for kind, sxy, srgb, rgb, compat in pairwise]
    q = _exp_and_normalize(-unary)
    for _ in range(n_iterations):
        tmp = -unary.copy()
        for (offs, ks, norm), compat in pre:
            qn = q * norm
            acc = np.zeros_like(q)
            for t, (dy, dx) in enumerate(offs):
                acc += ks[t] * _shift(qn, dy, dx)
            tmp += np.float32(compat) * (acc * norm)
        q = _exp_and_normalize(tmp)
    return q


def dense_crf(img, output_probs, mean, std, compat_gaussian=3, sxy_gaussian=1, compat_bilateral=10,
              sxy_bilateral=1, srgb=50, iterations=5):
    """Restatement of src/postprocessing.py:183-225 (img: normalised f[3,H,W], probs f[C,H,W])."""
    probs = np.asarray(output_probs)
    unary = -np.log(np.clip(probs, 1e-5, 1.0)).astype(np.float32)          # unary_from_softmax
    org = np.asarray(img) * np.array(std).reshape(3, 1, 1) + np.array(mean).reshape(3, 1, 1)
    org = np.ascontiguousarray((org * 255.).transpose(1, 2, 0), dtype=np.uint8)
    pw = [('gaussian', float(sxy_gaussian), None, None, float(compat_gaussian)),
          ('bilateral', float(sxy_bilateral), float(srgb), org, float(compat_bilateral))]
    return mean_field(unary, pw, iterations).reshape(probs.shape)
