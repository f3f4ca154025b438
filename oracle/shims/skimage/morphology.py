"""TEST INFRASTRUCTURE (oracle shim) -- skimage.morphology.{rectangle,erosion,dilation,
binary_erosion,binary_dilation} restated on scipy.ndimage the way skimage/morphology/grey.py
does it (call sites in the reference: src/postprocessing.py:148-154,172-179,
src/preparation.py:172-185)."""
import numpy as np
from scipy import ndimage as ndi


def rectangle(width, height, dtype=np.uint8):
    # skimage.morphology.selem.rectangle(width, height) -> ones((width, height))
    return np.ones((width, height), dtype=dtype)


def square(width, dtype=np.uint8):
    return np.ones((width, width), dtype=dtype)


def _shift_selem(selem, shift_x, shift_y):
    # even-sized 2-D structuring elements are padded to odd with a zero row on top /
    # zero column on the left (shift_x / shift_y False, the default)
    if selem.ndim != 2:
        return selem
    m, n = selem.shape
    if m % 2 == 0:
        extra_row = np.zeros((1, n), selem.dtype)
        selem = np.vstack((selem, extra_row)) if shift_x else np.vstack((extra_row, selem))
        m += 1
    if n % 2 == 0:
        extra_col = np.zeros((m, 1), selem.dtype)
        selem = np.hstack((selem, extra_col)) if shift_y else np.hstack((extra_col, selem))
    return selem


def _invert_selem(selem):
    return selem[(slice(None, None, -1),) * selem.ndim]


def erosion(image, selem=None, out=None, shift_x=False, shift_y=False):
    image = np.asarray(image)
    selem = _shift_selem(np.array(selem), shift_x, shift_y)
    src = image.astype(np.uint8) if image.dtype == bool else image
    res = ndi.grey_erosion(src, footprint=selem)
    return res.astype(image.dtype) if image.dtype == bool else res


def dilation(image, selem=None, out=None, shift_x=False, shift_y=False):
    image = np.asarray(image)
    selem = _shift_selem(np.array(selem), shift_x, shift_y)
    # ndi.grey_dilation mirrors the footprint internally; skimage pre-mirrors to cancel it
    selem = _invert_selem(selem)
    src = image.astype(np.uint8) if image.dtype == bool else image
    res = ndi.grey_dilation(src, footprint=selem)
    return res.astype(image.dtype) if image.dtype == bool else res


def binary_erosion(image, selem=None, out=None):
    return ndi.binary_erosion(image, structure=selem, border_value=True)


def binary_dilation(image, selem=None, out=None):
    return ndi.binary_dilation(image, structure=selem)
