"""TEST INFRASTRUCTURE (oracle shim): the handful of torchvision.transforms the reference
loaders use (src/loaders.py:291-317)."""
import numpy as np
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class ToTensor:
    def __call__(self, pic):
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
        return t.float().div(255) if t.dtype == torch.uint8 else t.float()


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, t):
        m = torch.tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        s = torch.tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return (t - m) / s


class Lambda:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, x):
        return self.fn(x)


class Resize:
    def __init__(self, size, interpolation=2):
        self.size = size

    def __call__(self, img):
        return img.resize((self.size[1], self.size[0]))


Scale = Resize
