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


from synthetic_inputs import seeded_state_dict, synthetic_batch      # noqa: E402,F401  (input generators: shared with bench.py, which may not import oracle/ on its product legs)
