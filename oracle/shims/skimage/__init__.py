"""TEST INFRASTRUCTURE (oracle shim): scikit-image is unpinned in the reference
(environment.yml:8) and not installed here; the few functions the hot path calls are
restated on scipy.ndimage exactly as skimage (0.13-0.15 era, the versions contemporary with
the reference) implements them.  Parity with skimage itself is therefore UNPINNED."""
from . import morphology, transform  # noqa: F401
