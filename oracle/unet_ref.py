"""TEST INFRASTRUCTURE -- torch-CPU fp32 oracle for the network on the hot path.

Restates `UNetResNet` (reference: src/unet_models.py:315-403) with the decoder pieces it is made
of (`ConvRelu` :25-34, `DecoderBlockV2` deconv branch :136-141) on top of the ResNet restatement
in oracle/shims/torchvision/models.py (torchvision==0.2.0 is not vendored in the reference).
The module tree and therefore the state_dict keys are the reference's, including the aliasing
of the encoder stages (`conv1.0.* == encoder.conv1.*`, `conv2.* == encoder.layer1.*`, ...), so a
state_dict moves freely between the reference class, this oracle and the HIP engine.

Pinned against the literal reference class in tests/test_oracle.py (bitwise equal
outputs for identical weights, run where /root/reference exists) and by tests/golden/unet_*.npz.
"""
import numpy as np
import torch
from torch import nn

from oracle.shims.torchvision import models as tvm

ENCODERS = {34: (tvm.resnet34, 512), 101: (tvm.resnet101, 2048), 152: (tvm.resnet152, 2048)}


class ConvReluRef(nn.Module):
    # src/unet_models.py:25-34  conv3x3(pad 1, bias) -> ReLU
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1)

    def forward(self, x):
        return torch.relu(self.conv(x))


class DecoderBlockRef(nn.Module):
    # src/unet_models.py:136-141  ConvRelu -> ConvTranspose2d(k4,s2,p1) -> ReLU
    def __init__(self, cin, cmid, cout):
        super().__init__()
        self.block = nn.Sequential(ConvReluRef(cin, cmid),
                                   nn.ConvTranspose2d(cmid, cout, kernel_size=4, stride=2, padding=1),
                                   nn.ReLU())

    def forward(self, x):
        return self.block(x)


class UNetResNetRef(nn.Module):
    def __init__(self, encoder_depth, num_classes=2, num_filters=32):
        super().__init__()
        if encoder_depth not in ENCODERS:
            raise NotImplementedError('only 34, 101, 152 version of Resnet are implemented')
        ctor, bottom = ENCODERS[encoder_depth]
        nf = num_filters
        self.encoder = ctor()
        self.pool = nn.MaxPool2d(2, 2)
        # stem uses a 2x2/2 pool, not the encoder's own 3x3/2 maxpool (src/unet_models.py:356,360-363)
        self.conv1 = nn.Sequential(self.encoder.conv1, self.encoder.bn1, self.encoder.relu, self.pool)
        self.conv2, self.conv3 = self.encoder.layer1, self.encoder.layer2
        self.conv4, self.conv5 = self.encoder.layer3, self.encoder.layer4
        self.center = DecoderBlockRef(bottom, nf * 16, nf * 8)                  # :373
        self.dec5 = DecoderBlockRef(bottom + nf * 8, nf * 16, nf * 8)           # :374
        self.dec4 = DecoderBlockRef(bottom // 2 + nf * 8, nf * 16, nf * 8)      # :375
        self.dec3 = DecoderBlockRef(bottom // 4 + nf * 8, nf * 8, nf * 2)       # :377
        self.dec2 = DecoderBlockRef(bottom // 8 + nf * 2, nf * 4, nf * 4)       # :379
        self.dec1 = DecoderBlockRef(nf * 4, nf * 4, nf)                         # :381
        self.dec0 = ConvReluRef(nf, nf)                                         # :382
        self.final = nn.Conv2d(nf, num_classes, kernel_size=1)                  # :383

    def forward(self, x):
        c1 = self.conv1(x)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        c4 = self.conv4(c3)
        c5 = self.conv5(c4)
        center = self.center(self.pool(c5))
        d5 = self.dec5(torch.cat([center, c5], 1))
        d4 = self.dec4(torch.cat([d5, c4], 1))
        d3 = self.dec3(torch.cat([d4, c3], 1))
        d2 = self.dec2(torch.cat([d3, c2], 1))
        d1 = self.dec1(d2)
        d0 = self.dec0(d1)
        return self.final(d0)       # dropout2d p=0.0 in every shipped config (src/models.py:34,39,44)


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
