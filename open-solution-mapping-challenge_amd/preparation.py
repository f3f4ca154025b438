"""Target preparation before the hot path: the per-image body of the reference's `overlay_masks`
(`src/preparation.py:18-99`, called from `src/pipeline_manager.py:68-85`) on the device: the plain overlay of the
shipped configuration (`erode_selem_size = 0`, `neptune.yaml:69`, in which `dilate_selem_size` plays no part,
`src/preparation.py:57-60`) and the eroded / eroded+dilated variants (`:61-77`).

The reference builds one full-image Euclidean distance transform per building, stacks them with `np.dstack`
and sorts the stack per pixel (O(buildings x H x W) host memory and time, `src/preparation.py:146-163`), then
writes `masks/*.png`, `distances/*` and `sizes/*` joblib side files that the weighted loader reads back
(`src/loaders.py:147-153`).  `prepare_targets` produces the same three arrays from the decoded instance masks
in a handful of kernel launches (`csrc/prep.hip`), so they can also be generated on the fly.

Inputs are the decoded annotation masks (`cocomask.decode(cocomask.frPyObjects(...))`): polygon rasterisation
and COCO json handling stay with pycocotools on the host.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _device():
    if not torch.cuda.is_available():
        raise _lib.MscError('target preparation needs a ROCm GPU: the product has no CPU path')
    return torch.device('cuda', torch.cuda.current_device())


def get_size_matrix(mask):
    """src/preparation.py:181-187: per pixel the area of its 4-connected component of `mask != 0`, 1 on background.
    mask: [H,W] numpy / cuda tensor; returns int64 numpy [H,W]."""
    dev = _device()
    m = (torch.as_tensor(mask).to(dev) != 0).to(torch.uint8).contiguous()
    return _size_matrix(m[None])[0].cpu().numpy().astype(np.int64)


def _size_matrix(mask_u8):
    """u8 cuda [B,H,W] -> int32 cuda [B,H,W]"""
    lib = _lib.load()
    B, H, W = mask_u8.shape
    dev = mask_u8.device
    stream = torch.cuda.current_stream(dev).cuda_stream
    labels = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    ws = torch.empty(lib.msc_label_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)
    _lib.check(lib.msc_label4(mask_u8.data_ptr(), labels.data_ptr(), counts.data_ptr(), ws.data_ptr(), B, H, W, stream), 'msc_label4')
    max_labels = (H * W + 1) // 2                 # a 4-connected labelling cannot have more components
    areas = torch.empty(B * (max_labels + 1), dtype=torch.int32, device=dev)
    sizes = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    _lib.check(lib.msc_size_matrix(labels.data_ptr(), sizes.data_ptr(), areas.data_ptr(), B, H, W, max_labels, stream), 'msc_size_matrix')
    return sizes


def prepare_targets(masks, category_nr=None, border_width=0, erode=0, dilate=0, small_annotations_size=14, return_details=False):
    """overlay_mask_one_image (src/preparation.py:44-84) for one image.

    masks: uint8 [n,H,W] decoded instance masks in annotation order (n may be 0: pass an array of shape [0,H,W]);
    category_nr: per instance the index of its category in CATEGORY_IDS (default 1 = the single building class).
    Returns (mask_overlayed uint8 [H,W], distances float16 [H,W], sizes int64 [H,W]) as numpy arrays -- what the
    reference writes to masks/, distances/ and sizes/.  return_details adds (second_nearest f64, kept i32[n])."""
    if erode < 0 or dilate < 0:
        raise ValueError('erode and dilate cannot be negative')                   # src/preparation.py:54-55
    lib = _lib.load()
    dev = _device()
    m = torch.as_tensor(masks)
    if m.dim() != 3:
        raise ValueError('masks must be [n, H, W]')
    m = (m.to(dev) != 0).to(torch.uint8).contiguous()
    n, H, W = m.shape
    stream = torch.cuda.current_stream(dev).cuda_stream
    cats_h = np.ones(n, np.int32) if category_nr is None else np.asarray(category_nr, dtype=np.int32)
    if cats_h.size != n:
        raise ValueError('category_nr must have one entry per mask')
    order = np.argsort(cats_h, kind='stable')        # the reference walks the categories in turn (:49-52)
    if n and (order != np.arange(n)).any():
        m = m[torch.from_numpy(order).to(dev)].contiguous()
    cats_s = cats_h[order]
    cat = torch.from_numpy(np.ascontiguousarray(cats_s)).to(dev) if n else None

    def targets(masks_used, border_masks, cat_t, k):
        overlay = torch.empty((H, W), dtype=torch.uint8, device=dev)
        dist = torch.empty((H, W), dtype=torch.int16, device=dev)       # float16 bit patterns
        second = torch.empty((H, W), dtype=torch.float64, device=dev)
        kept = torch.zeros(max(k, 1), dtype=torch.int32, device=dev)
        ws = torch.empty(lib.msc_prep_workspace_bytes(k, H, W), dtype=torch.uint8, device=dev)
        _lib.check(lib.msc_prep_targets(masks_used.data_ptr() if k else None, border_masks.data_ptr() if (k and border_masks is not None) else None,
                                        cat_t.data_ptr() if cat_t is not None else None, k, H, W, overlay.data_ptr(), dist.data_ptr(),
                                        second.data_ptr(), kept.data_ptr(), ws.data_ptr(), stream), 'msc_prep_targets')
        return overlay, dist, second, kept

    if erode > 0 and n:
        # :61-77 -- every instance is replaced by its eroded (big) / unchanged or dilated (small) form before it enters the
        # overlay and the distance stack; is_on_border still looks at the annotation itself
        chosen = torch.empty_like(m)
        ws = torch.empty(lib.msc_prep_morph_workspace_bytes(n, H, W), dtype=torch.uint8, device=dev)
        _lib.check(lib.msc_prep_morph(m.data_ptr(), n, H, W, int(erode), int(dilate), int(small_annotations_size), chosen.data_ptr(),
                                      ws.data_ptr(), stream), 'msc_prep_morph')
        overlay, dist, second, kept = targets(chosen, m, cat, n)
        if dilate == 0:
            # :62-71 -- per category: the eroded overlay plus every component of the plain overlay that erosion wiped out
            from . import postprocessing as post
            overlay = torch.zeros((H, W), dtype=torch.uint8, device=dev)
            for c in sorted(set(cats_s.tolist())):
                idx = torch.from_numpy(np.flatnonzero(cats_s == c)).to(dev)
                k = int(idx.numel())
                plain = targets(m[idx].contiguous(), None, None, k)[0]
                eroded = targets(chosen[idx].contiguous(), m[idx].contiguous(), None, k)[0]
                mask_c = post.add_dropped_batch(plain[None], eroded[None])[0]
                _lib.check(lib.msc_prep_paint(overlay.data_ptr(), mask_c.data_ptr(), int(c), H, W, stream), 'msc_prep_paint')
    else:
        overlay, dist, second, kept = targets(m, None, cat, n)
    sizes = _size_matrix(overlay[None])[0]          # from the overlay BEFORE the border class is painted (:70 before :73)
    if border_width > 0:
        scratch = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.msc_prep_border(overlay.data_ptr(), second.data_ptr(), C.c_double(float(border_width)), scratch.data_ptr(), H, W, stream),
                   'msc_prep_border')
    out = (overlay.cpu().numpy(), dist.cpu().numpy().view(np.float16), sizes.cpu().numpy().astype(np.int64))
    if return_details:
        kept_h = np.zeros(n, np.int32)
        kept_h[order] = kept[:n].cpu().numpy()          # back to the caller's instance order
        out += (second.cpu().numpy(), kept_h)
    return out
