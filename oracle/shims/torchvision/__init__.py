"""TEST INFRASTRUCTURE (oracle shim): minimal stand-in for torchvision==0.2.0 so the
reference's src/unet_models.py and src/loaders.py import unmodified."""
from . import models, transforms  # noqa: F401
