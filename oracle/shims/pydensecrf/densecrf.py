"""TEST INFRASTRUCTURE (oracle shim): DenseCRF2D API on top of oracle/crf_ref.py."""
import numpy as np

from oracle import crf_ref

DIAG_KERNEL = 1
NORMALIZE_SYMMETRIC = 3


class DenseCRF2D:
    def __init__(self, w, h, nlabels):
        self.w, self.h, self.m = w, h, nlabels
        self.unary = None
        self.pairwise = []

    def setUnaryEnergy(self, u):
        self.unary = np.asarray(u, dtype=np.float32).reshape(self.m, self.h, self.w)

    def addPairwiseGaussian(self, sxy, compat=3, kernel=DIAG_KERNEL, normalization=NORMALIZE_SYMMETRIC):
        self.pairwise.append(('gaussian', float(sxy), None, None, float(compat)))

    def addPairwiseBilateral(self, sxy, srgb, rgbim, compat=10, kernel=DIAG_KERNEL,
                             normalization=NORMALIZE_SYMMETRIC):
        assert rgbim.shape == (self.h, self.w, 3) and rgbim.dtype == np.uint8
        self.pairwise.append(('bilateral', float(sxy), float(srgb), rgbim, float(compat)))

    def inference(self, n_iterations):
        q = crf_ref.mean_field(self.unary, self.pairwise, n_iterations)
        return q.reshape(self.m, -1)
