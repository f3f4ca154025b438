"""TEST INFRASTRUCTURE (oracle shim): the part of imgaug's interface the reference's loaders touch
(src/augmentation.py, src/steps/pytorch/utils.py:108-129), so that the reference's OWN loader Steps can feed
the pipeline of BASELINE.json configs[0] offline.  The random augmenters (flips, affine, colour shifts) are the
IDENTITY here: the plumbing test needs batches of the right shape and type, not augmentation statistics."""
from . import augmenters  # noqa: F401
