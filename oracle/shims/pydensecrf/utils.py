"""TEST INFRASTRUCTURE (oracle shim): pydensecrf.utils.unary_from_softmax restated."""
import numpy as np


def unary_from_softmax(sm, scale=None, clip=1e-5):
    num_cls = sm.shape[0]
    if scale is not None:
        assert 0 < scale <= 1
        uniform = np.ones(sm.shape) / num_cls
        sm = scale * sm + (1 - scale) * uniform
    if clip is not None:
        sm = np.clip(sm, clip, 1.0)
    return -np.log(sm).reshape([num_cls, -1]).astype(np.float32)
