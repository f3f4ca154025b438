"""UNetResNet on MI355X: drop-in for the reference's `src/unet_models.py:315-403`.

Same constructor, same parameter / buffer names (so reference checkpoints load, including the
`module.` prefix DataParallel adds, src/steps/pytorch/models.py:148-160), same
`forward(x: f32[N,3,H,W]) -> logits f32[N,2,H,W]` -- but every FLOP runs in the hand-written
gfx950 kernels behind include/msc.h.  torch is used for device memory, the module / parameter
containers and (optionally) to hang the whole network into autograd as ONE node.

Execution model: for a given (N, H, W, dtype, train/eval) the network is compiled once into a
static launch list over pre-allocated NHWC buffers (288 GB of HBM: nothing is recomputed or
re-allocated, the skip-concats of :395-399 are channel slices of one buffer), which is also what
makes the whole train step capturable into one hipGraph.

  eval : conv -> (BN folded into the conv epilogue) -> ReLU, 1 launch per conv
  train: conv (raw output + per-tile BN partial sums in the epilogue) -> bn_finalize -> bn_apply
         backward: bn_bwd_reduce -> bn_bwd_finalize -> bn_bwd_apply -> wgrad -> dgrad
"""
import ctypes as C

import torch
from torch import nn

from . import _lib
from ._lib import BneckDesc, BnInput, ConvDesc, WgradDesc, F32, BF16, F16

_DTYPES = {'bf16': (torch.bfloat16, BF16), 'fp16': (torch.float16, F16), 'fp32': (torch.float32, F32)}

_ENC = {34: ('basic', [3, 4, 6, 3], 512), 101: ('bottle', [3, 4, 23, 3], 2048), 152: ('bottle', [3, 8, 36, 3], 2048)}
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


# ----------------------------------------------------------------------------- parameter containers
# Plain torch.nn modules used ONLY as named parameter/buffer holders (never called): the names are the
# torchvision ResNet ones the reference's state_dict carries (src/unet_models.py:345-371).
class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = down
        self.stride = stride


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = down
        self.stride = stride


class _ResNetParams(nn.Module):
    def __init__(self, kind, layers):
        super().__init__()
        block = _BasicBlock if kind == 'basic' else _Bottleneck
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)      # present in the reference's module tree, unused (:356)
        cin = 64
        for li, (planes, n, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 2)), start=1):
            blocks = []
            for b in range(n):
                s = stride if b == 0 else 1
                down = None
                if b == 0 and (s != 1 or cin != planes * block.expansion):
                    down = nn.Sequential(nn.Conv2d(cin, planes * block.expansion, 1, s, bias=False),
                                         nn.BatchNorm2d(planes * block.expansion))
                blocks.append(block(cin, planes, s, down))
                cin = planes * block.expansion
            setattr(self, 'layer%d' % li, nn.Sequential(*blocks))
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(cin, 1000)            # dead weights kept so reference checkpoints load strictly


class _ConvReluParams(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1)
        self.activation = nn.ReLU(inplace=True)


class _DecoderParams(nn.Module):
    def __init__(self, cin, cmid, cout):
        super().__init__()
        self.in_channels = cin
        self.block = nn.Sequential(_ConvReluParams(cin, cmid),
                                   nn.ConvTranspose2d(cmid, cout, kernel_size=4, stride=2, padding=1),
                                   nn.ReLU(inplace=True))


# ----------------------------------------------------------------------------- NHWC activation views
class Act:
    """Channel slice [c0, c0+C) of an NHWC buffer [N,H,W,Ct]."""
    __slots__ = ('buf', 'c0', 'C')

    def __init__(self, buf, c0=0, C=None):
        self.buf, self.c0 = buf, c0
        self.C = buf.shape[3] - c0 if C is None else C

    @property
    def N(self): return self.buf.shape[0]

    @property
    def H(self): return self.buf.shape[1]

    @property
    def W(self): return self.buf.shape[2]

    @property
    def ld(self): return self.buf.shape[3]

    @property
    def ptr(self): return self.buf.data_ptr() + self.c0 * self.buf.element_size()

    @property
    def pixels(self): return self.buf.shape[0] * self.buf.shape[1] * self.buf.shape[2]

    def view(self):
        return self.buf[..., self.c0:self.c0 + self.C]


def _p(t):
    return None if t is None else t.data_ptr()


def _stream_of(device):
    return torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0


import os as _os_env
# measured on MI355X (R101 train step, hipGraph): 18.8 ms with the wgrad kernels on a second stream vs 17.6 ms
# serial -- the concurrent kernels fight over L2/LDS-DMA bandwidth -- so the fork/join path is opt-in
_OVERLAP_WGRAD = _os_env.environ.get('MSC_OVERLAP_WGRAD', '0') == '1'
_WGRAD_FLUSH_DEFAULT = ''
# round 6: the launches of a weight-gradient group (one per tile shape) as parallel branches of the step's graph, on this many side streams (1 = in a row)
_WGRAD_PARTS = int(_os_env.environ.get('MSC_WGRAD_PARTS', '1'))
_PART_STREAMS = {}
_SIDE_STREAMS = {}


def _quiesce(device):
    """Before the flat buffers / a program's buffers are allocated and its kernels timed: collect what earlier users of the device left behind and
    wait for EVERYTHING they still have in flight (a device-wide synchronise, not a stream one).  Round 5: a captured training step built while another
    component's work was still pending could replay with garbage gradients (tests/test_gpu_configs.py in file order).  The cause turned out to be the
    memset / memcpy NODES hipMemsetAsync / hipMemcpyAsync become under capture -- msc_memset_zero / msc_copy are kernels since then (csrc/elementwise.hip,
    DESIGN.md section 3) -- and this synchronise was the first mitigation that worked; it stays, once per program, because it costs nothing."""
    if _os_env.environ.get('MSC_NO_QUIESCE') == '1':          # measurement switch (tests/test_gpu_replay_hazard.py, probes/replay_order_probe.py)
        return
    if device is not None and torch.device(device).type == 'cuda' and torch.cuda.is_available():
        import gc
        gc.collect()
        torch.cuda.synchronize()


def _side_stream(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class _Program:
    """Static launch list for one (N,H,W,dtype,training) configuration."""

    def __init__(self):
        self.fwd, self.bwd, self.keep = [], [], []
        self.groups = []        # msc_wgrad_group handles owned by this program
        self.x_in = None        # f32 NCHW input staging buffer
        self.logits = None      # f32 NCHW
        self.probs = None       # f32 NCHW (eval)
        self.dlogits = None     # f32 NCHW (train)
        self.bytes = 0

    def __del__(self):
        try:
            lib = _lib.load()
            for h in self.groups:
                lib.msc_wgrad_group_destroy(h)
        except Exception:
            pass
        self.groups = []

    @staticmethod
    def run(launches, stream):
        for fn, args in launches:
            rc = fn(*args, stream)
            if rc != 0:
                _lib.check(rc, fn.__name__)

    SIDE = ('msc_conv_wgrad', 'msc_wgrad_group_run', 'msc_stem_unpack_grad')

    @staticmethod
    def run_group_parts(handle, device, nstreams):
        """The launches of one weight-gradient group (one per tile shape; disjoint gradient buffers) as parallel branches: part i on side stream
        i % nstreams, forked from the current stream and joined back into it (graph edges under capture).  (round 6, MSC_WGRAD_PARTS=n; msc_wgrad_group_run_part, ABI v11)"""
        lib = _lib.load()
        main = torch.cuda.current_stream(device)
        if getattr(handle, 'ordered', False):           # the deterministic form keeps its launches in a row (the finishing pass follows all of them)
            rc = lib.msc_wgrad_group_run(handle, main.cuda_stream)
            if rc != 0:
                _lib.check(rc, 'msc_wgrad_group_run')
            return
        nb = lib.msc_wgrad_group_launches(handle)
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        pool = _PART_STREAMS.setdefault(key, [])
        while len(pool) < nstreams:
            pool.append(torch.cuda.Stream(device=device))
        ev = torch.cuda.Event()
        ev.record(main)
        used = []
        for i in range(nb):
            st = pool[i % nstreams]
            if st not in used:
                st.wait_event(ev)
                used.append(st)
            rc = lib.msc_wgrad_group_run_part(handle, i, st.cuda_stream)
            if rc != 0:
                _lib.check(rc, 'msc_wgrad_group_run_part')
        for st in used:
            done = torch.cuda.Event()
            done.record(st)
            main.wait_event(done)

    @staticmethod
    def run_backward(launches, device):
        """Backward launch list with the weight-gradient kernels on a second HIP stream.  A wgrad only reads buffers
        that are final once it is reached in list order (dy, the layer input) and atomically adds into its own slice
        of the flat gradient buffer, so it can run beside the rest of the backward chain (BN backward -> dgrad -> ...),
        which on the small layers of the encoder leaves most CUs idle.  The side stream waits for everything launched
        so far before each wgrad and is joined at the end; under hipGraph capture this becomes graph edges."""
        if device.type == 'cuda' and _WGRAD_PARTS > 1 and not _OVERLAP_WGRAD:
            mh = _stream_of(device)
            for fn, args in launches:
                if fn.__name__ == 'msc_wgrad_group_run':
                    _Program.run_group_parts(args[0], device, _WGRAD_PARTS)
                else:
                    rc = fn(*args, mh)
                    if rc != 0:
                        _lib.check(rc, fn.__name__)
            return
        if device.type != 'cuda' or not _OVERLAP_WGRAD:
            return _Program.run(launches, _stream_of(device))
        main = torch.cuda.current_stream(device)
        side = _side_stream(device)
        mh, sh = main.cuda_stream, side.cuda_stream
        used, fresh = False, True
        for fn, args in launches:
            if fn.__name__ in _Program.SIDE:
                if fresh:                          # main-stream work was issued since the last fork point
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                    fresh = False
                rc = fn(*args, sh)
                used = True
            else:
                rc = fn(*args, mh)
                fresh = True
            if rc != 0:
                _lib.check(rc, fn.__name__)
        if used:
            ev = torch.cuda.Event()
            ev.record(side)
            main.wait_event(ev)


# ----------------------------------------------------------------------------- the network
class UNetResNet(nn.Module):
    """PyTorch-API U-Net with ResNet(34, 101 or 152) encoder executing on hand-written HIP kernels.

    Args mirror src/unet_models.py:338-339.  `pretrained` cannot download ImageNet weights offline
    and is ignored (weights come from load_state_dict); `is_deconv` must be True and `dropout_2d`
    0.0, the only values any shipped configuration uses (src/models.py:32-46).
    Extra keyword: compute_dtype 'bf16' (throughput mode, default), 'fp16' (BASELINE.json configs[4]; training in it needs
    the static loss scale TrainStep applies, inference does not) or 'fp32' (exact-f32 parity mode).
    deterministic (or MSC_DETERMINISTIC=1; read when a training program is built): the weight gradients' partial sums are added in a
    fixed order instead of by fp32 atomics (msc_wgrad_group_create MSC_WGRAD_ORDERED, msc_final_bwd ordered_ws) -- two runs of the same
    steps from the same state give the same bits, at the price of one more pass over the split planes per step.  What stays atomic in
    this mode: the BatchNorm / bias sums (per-XCD slots of float64 added in arrival order -- fp32 partials summed in double, so the
    result can differ between runs only when a rounding of the DOUBLE sum falls differently, which the repeat test has not seen but the
    arithmetic does not exclude); it needs MSC_WGRAD_GROUP > 1 and raises otherwise.
    """

    def __init__(self, encoder_depth, num_classes, num_filters=32, dropout_2d=0.2, pretrained=False,
                 is_deconv=False, compute_dtype='bf16', autotune=None, deterministic=None):
        super().__init__()
        self.deterministic = (_os_env.environ.get('MSC_DETERMINISTIC', '0') == '1') if deterministic is None else bool(deterministic)
        if encoder_depth not in _ENC:
            raise NotImplementedError('only 34, 101, 152 version of Resnet are implemented')
        if num_classes != 2:
            raise NotImplementedError('HIP path implements the 2-class head of the mapping challenge (num_classes=2)')
        if not is_deconv:
            raise NotImplementedError('HIP path implements the is_deconv=True decoder (every shipped config, src/models.py:32-46)')
        if dropout_2d != 0.0:
            raise NotImplementedError('HIP path implements dropout_2d=0.0 (every shipped config, src/models.py:34,39,44)')
        if num_filters % 32:
            raise NotImplementedError('num_filters must be a multiple of 32')
        self.num_classes, self.dropout_2d, self.encoder_depth = num_classes, dropout_2d, encoder_depth
        # the reference downloads ImageNet weights for pretrained=True (torchvision model zoo); there is no network here:
        # the request is remembered and the transformers warn if training starts from random encoder weights
        self.pretrained_requested, self.weights_loaded = bool(pretrained), False
        kind, layers, bottom = _ENC[encoder_depth]
        self._kind, self._bottom, self._nf = kind, bottom, num_filters
        nf = num_filters
        self.encoder = _ResNetParams(kind, layers)
        self.pool = nn.MaxPool2d(2, 2)
        self.relu = nn.ReLU(inplace=True)
        # same aliasing as the reference (:360-371): state_dict carries both key sets
        self.conv1 = nn.Sequential(self.encoder.conv1, self.encoder.bn1, self.encoder.relu, self.pool)
        self.conv2, self.conv3 = self.encoder.layer1, self.encoder.layer2
        self.conv4, self.conv5 = self.encoder.layer3, self.encoder.layer4
        self.center = _DecoderParams(bottom, nf * 16, nf * 8)
        self.dec5 = _DecoderParams(bottom + nf * 8, nf * 16, nf * 8)
        self.dec4 = _DecoderParams(bottom // 2 + nf * 8, nf * 16, nf * 8)
        self.dec3 = _DecoderParams(bottom // 4 + nf * 8, nf * 8, nf * 2)
        self.dec2 = _DecoderParams(bottom // 8 + nf * 2, nf * 4, nf * 4)
        self.dec1 = _DecoderParams(nf * 4, nf * 4, nf)
        self.dec0 = _ConvReluParams(nf, nf)
        self.final = nn.Conv2d(nf, num_classes, kernel_size=1)
        self.set_compute_dtype(compute_dtype)
        # time the valid kernel configurations of every conv / wgrad launch when a program is built (MSC_AUTOTUNE=0 disables)
        import os as _os
        self.autotune = (_os.environ.get('MSC_AUTOTUNE', '1') != '0') if autotune is None else bool(autotune)
        self._programs = {}
        self._flat = None
        self._version = 0          # bumped whenever master weights change -> packed copies are stale
        self._packed_version = -1
        self._folded_version = -1

    # ------------------------------------------------------------------ configuration
    def set_compute_dtype(self, compute_dtype):
        if compute_dtype not in _DTYPES:
            raise ValueError("compute_dtype must be 'bf16', 'fp16' or 'fp32'")
        self.compute_dtype = compute_dtype
        self._tdtype, self._dt = _DTYPES[compute_dtype]
        self._programs = {}
        self._pack = None
        return self

    def weights_changed(self):
        """Call after modifying parameters outside HipAdam / load_state_dict."""
        self._version += 1

    def load_state_dict(self, state_dict, strict=True, **kw):
        # accept DataParallel-prefixed checkpoints (reference Model.load wraps first, models.py:151-156)
        if state_dict and all(k.startswith('module.') for k in state_dict):
            state_dict = {k[len('module.'):]: v for k, v in state_dict.items()}
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._version += 1
        self.weights_loaded = True
        return out

    def load_encoder_state_dict(self, state_dict):
        """ImageNet weights for the encoder from a local torchvision ResNet checkpoint (a state_dict or a path to one):
        what `pretrained=True` fetches from the model zoo in the reference (src/unet_models.py:344-352)"""
        if isinstance(state_dict, str):
            state_dict = torch.load(state_dict, map_location='cpu')
        out = self.encoder.load_state_dict(state_dict, strict=False)
        if out.unexpected_keys:
            raise KeyError('not a torchvision ResNet%d state_dict: unexpected keys %s' % (self.encoder_depth, out.unexpected_keys[:4]))
        self._version += 1
        self.weights_loaded = True
        return out

    # ------------------------------------------------------------------ flat parameter storage
    def _trainable(self):
        return [(n, p) for n, p in self.named_parameters() if not n.startswith('encoder.fc.')]

    def flatten_parameters(self, device=None):
        """Move all trainable parameters into ONE flat fp32 buffer (and gradients into another);
        conv / deconv weights are stored channels-last, i.e. physically [Cout][KH][KW][Cin]
        ([Cin][KH][KW][Cout] for ConvTranspose2d) -- the k-contiguous layout the kernels read --
        while keeping their torch-logical shapes, so state_dicts stay interchangeable."""
        device = torch.device(device if device is not None else 'cuda')
        _quiesce(device)
        params = self._trainable()
        offs, total = [], 0
        for _, p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        grad = torch.zeros(total, dtype=torch.float32, device=device)
        gptr = {}
        for (name, p), off in zip(params, offs):
            n = p.numel()
            if p.dim() == 4 and not name.endswith('encoder.conv1.weight') and p.shape[2] * p.shape[3] * p.shape[1] > 1:
                a, b, kh, kw = p.shape
                view = flat[off:off + n].view(a, kh, kw, b).permute(0, 3, 1, 2)
                gview = grad[off:off + n].view(a, kh, kw, b).permute(0, 3, 1, 2)
            else:
                view = flat[off:off + n].view(p.shape)
                gview = grad[off:off + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = gview
            gptr[id(p)] = gview.data_ptr()
        for b in self.buffers():
            b.data = b.data.to(device)
        self.encoder.fc.to(device)
        self._flat = (flat, grad, params[0][1].data_ptr())
        self._gptr = gptr
        self._programs = {}
        self._pack = None
        self._version += 1
        return flat, grad

    def _ensure_flat(self, device):
        first = self._trainable()[0][1]
        if self._flat is None or self._flat[0].device != device or first.data_ptr() != self._flat[2]:
            self.flatten_parameters(device)
        return self._flat[0], self._flat[1]

    def _g(self, param):
        """device address of `param`'s gradient inside the flat gradient buffer"""
        return self._gptr[id(param)]

    @property
    def flat_params(self):
        return None if self._flat is None else self._flat[0]

    @property
    def flat_grads(self):
        return None if self._flat is None else self._flat[1]

    # ------------------------------------------------------------------ weight packing
    def _conv_list(self):
        """[(name, module, kind)] for every conv on the path, in forward order."""
        out = [('encoder.conv1', self.encoder.conv1, 'stem')]
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(self.encoder, 'layer%d' % li)):
                base = 'encoder.layer%d.%d' % (li, bi)
                names = ['conv1', 'conv2'] + (['conv3'] if self._kind == 'bottle' else [])
                for cn in names:
                    out.append(('%s.%s' % (base, cn), getattr(blk, cn), 'conv'))
                if blk.downsample is not None:
                    out.append((base + '.downsample.0', blk.downsample[0], 'conv'))
        for dn in ('center', 'dec5', 'dec4', 'dec3', 'dec2', 'dec1'):
            d = getattr(self, dn)
            out.append((dn + '.block.0.conv', d.block[0].conv, 'conv'))
            out.append((dn + '.block.1', d.block[1], 'deconv'))
        out.append(('dec0.conv', self.dec0.conv, 'conv'))
        return out

    def _build_pack(self):
        """Allocate compute copies of the weights and the launch list that refreshes them: ONE msc_pack_multi
        launch over a device-resident table (plus the stem's own layout change)."""
        import numpy as np
        lib = _lib.load()
        dev = self._flat[0].device
        pk = {'ops': [], 'w': {}, 'wt': {}, 'keep': []}
        items, blk_item, blk_local = [], [], []

        def add(kind, src, dst, a, t, b, n):
            idx = len(items)
            items.append((src.data_ptr(), dst.data_ptr(), kind, a, t, b, n))
            nb = (n + 2047) // 2048 if kind == 0 else t * ((a + 63) // 64) * ((b + 31) // 32)     # as msc_pack_multi tiles
            blk_item.append(np.full(nb, idx, np.int32))
            blk_local.append(np.arange(nb, dtype=np.int32))

        for name, m, kind in self._conv_list():
            w = m.weight
            if kind == 'stem':
                buf = torch.empty(64 * 7 * 32, dtype=self._tdtype, device=dev)
                pk['w'][name] = buf
                pk['ops'].append((lib.msc_stem_pack, (w.data_ptr(), buf.data_ptr(), self._dt, 64)))
                continue
            a, b, kh, kw = w.shape            # physical [a][kh][kw][b]
            n = w.numel()
            direct = w if self._dt == F32 else None
            if direct is None:
                direct = torch.empty(n, dtype=self._tdtype, device=dev)
                add(0, w, direct, a, kh * kw, b, n)
            trans = torch.empty(n, dtype=self._tdtype, device=dev)
            add(1, w, trans, a, kh * kw, b, n)
            if kind == 'conv':       # master [Cout][..][Cin]: forward uses direct, dgrad the transpose
                pk['w'][name], pk['wt'][name] = direct, trans
            else:                    # master [Cin][..][Cout]: forward uses the transpose, dgrad direct
                pk['w'][name], pk['wt'][name] = trans, direct
        rec = np.zeros(len(items), dtype=np.dtype({'names': ['src', 'dst', 'kind', 'A', 'T', 'B', 'n'],
                                                   'formats': ['<u8', '<u8', '<i4', '<i4', '<i4', '<i4', '<i8'],
                                                   'offsets': [0, 8, 16, 20, 24, 28, 32], 'itemsize': 40}))
        for i, it in enumerate(items):
            rec[i] = it
        t_items = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        t_bi = torch.from_numpy(np.concatenate(blk_item)).to(dev)
        t_bl = torch.from_numpy(np.concatenate(blk_local)).to(dev)
        pk['keep'] += [t_items, t_bi, t_bl]
        pk['ops'].append((lib.msc_pack_multi, (t_items.data_ptr(), t_bi.data_ptr(), t_bl.data_ptr(), int(t_bi.numel()), self._dt)))
        return pk

    def _refresh_weights(self, stream):
        if self._pack is None:
            self._pack = self._build_pack()
            self._programs = {}
            self._packed_version = -1
        if self._packed_version != self._version:
            _Program.run(self._pack['ops'], stream)
            self._packed_version = self._version

    # ------------------------------------------------------------------ program construction
    def _program(self, N, H, W, training, device):
        key = (N, H, W, training, self.compute_dtype)
        prog = self._programs.get(key)
        if prog is None:
            _quiesce(device)
            if H % 64 or W % 64:
                raise ValueError('UNetResNet needs H and W divisible by 64 (got %dx%d): the reference fails at the '
                                 'first skip-concat otherwise (src/unet_models.py:360-363,392-397)' % (H, W))
            prog = _Builder(self, N, H, W, training, device).build()
            self._programs[key] = prog
        return prog

    # ------------------------------------------------------------------ execution
    def _run_forward(self, x, training):
        _require_device(x)
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError('expected input of shape [N,3,H,W], got %s' % (tuple(x.shape),))
        if x.shape[2] % 64 or x.shape[3] % 64:
            raise ValueError('UNetResNet needs H and W divisible by 64 (got %dx%d): the reference fails at the '
                             'first skip-concat otherwise (src/unet_models.py:360-363,392-397)' % (x.shape[2], x.shape[3]))
        _lib.load()
        dev = x.device
        self._ensure_flat(dev)
        stream = _stream_of(dev)
        self._refresh_weights(stream)
        N, _, H, W = x.shape
        prog = self._program(N, H, W, training, dev)
        prog.x_in.copy_(x.detach().to(torch.float32))
        if not training and prog.fold_version != self._version:
            _Program.run(prog.fold, stream)
            prog.fold_version = self._version
        _Program.run(prog.fwd, stream)
        prog.bwd_pending = training
        return prog

    def _run_backward(self, prog, dlogits, zero_grads=True):
        # ONE backward per forward: the BatchNorm backward overwrites the raw conv outputs it reads (dy in place of y), so a
        # second pass over the same forward would differentiate garbage -- fail loudly instead
        if not getattr(prog, 'bwd_pending', False):
            raise _lib.MscError('UNetResNet backward called twice for one forward (or without a training forward): the backward '
                                'consumes the saved activations in place; run the forward again')
        prog.bwd_pending = False
        stream = _stream_of(dlogits.device)
        if zero_grads:
            self._flat[1].zero_()
        if dlogits.data_ptr() != prog.dlogits.data_ptr():
            prog.dlogits.copy_(dlogits)
        prog.stem_dw.zero_()
        _Program.run_backward(prog.bwd, dlogits.device)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return _UNetFunction.apply(x, self, *[p for _, p in self._trainable()])
        prog = self._run_forward(x, training=self.training)
        return prog.logits.clone()

    def predict_proba(self, x):
        """eval forward fused with the channel softmax the reference runs on the host afterwards
        (src/models.py:88-92): returns f32[N,2,H,W] probabilities (a view of the program's buffer)."""
        was = self.training
        self.eval()
        try:
            prog = self._run_forward(x, training=False)
        finally:
            self.train(was)
        return prog.probs

    # training fast path used by models.py / bench.py (no autograd graph)
    def train_forward(self, x):
        return self._run_forward(x, training=True)

    def train_backward(self, prog, dlogits=None):
        self._run_backward(prog, prog.dlogits if dlogits is None else dlogits)


class AlbuNet(UNetResNet):
    """src/unet_models.py:153-221: the ResNet34-encoder U-Net under its other name (`PRETRAINED_NETWORKS['AlbuNet']`,
    src/models.py:29-31): the graph and state_dict of UNetResNet(34) without the dropout argument."""

    def __init__(self, num_classes=1, num_filters=32, pretrained=False, is_deconv=False, compute_dtype='bf16', autotune=None):
        super().__init__(34, num_classes, num_filters=num_filters, dropout_2d=0.0, pretrained=pretrained, is_deconv=is_deconv,
                         compute_dtype=compute_dtype, autotune=autotune)


class _UNetFunction(torch.autograd.Function):
    """The whole network as one autograd node: lets the reference's own training loop
    (`loss.backward(); optimizer.step()`, src/steps/pytorch/models.py:110-111) drive the HIP engine."""

    @staticmethod
    def forward(ctx, x, net, *params):
        prog = net._run_forward(x, training=True)
        ctx.net, ctx.prog = net, prog
        return prog.logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        net, prog = ctx.net, ctx.prog
        flat_g = net._flat[1]
        saved = flat_g.clone()           # .grad accumulated so far lives in the same flat buffer
        net._run_backward(prog, dlogits.contiguous().to(torch.float32), zero_grads=True)
        grads = [g.clone() for g in net._grad_views()]
        flat_g.copy_(saved)              # autograd adds `grads` onto it (or assigns if .grad is None)
        return (None, None) + tuple(grads)


def _flat_views(self, flat):
    """per-parameter views (torch-logical shapes, in _trainable() order) of a flat fp32 buffer laid out like the flat parameter
    buffer: the gradients, or the optimizer's moment buffers"""
    out, total = [], 0
    for name, p in self._trainable():
        n = p.numel()
        if p.dim() == 4 and not name.endswith('encoder.conv1.weight') and p.shape[2] * p.shape[3] * p.shape[1] > 1:
            a, b, kh, kw = p.shape
            out.append(flat[total:total + n].view(a, kh, kw, b).permute(0, 3, 1, 2))
        else:
            out.append(flat[total:total + n].view(p.shape))
        total += (n + 3) // 4 * 4
    return out


def _grad_views(self):
    return self.flat_views(self._flat[1])


UNetResNet.flat_views = _flat_views
UNetResNet._grad_views = _grad_views


def _require_device(x):
    """the product computes on the GPU only: a host tensor is an error, not a fallback"""
    if not x.is_cuda:
        raise _lib.MscError('UNetResNet (HIP) needs a CUDA/ROCm tensor; there is no CPU path in the product')


# ----------------------------------------------------------------------------- program builder
_TUNE_CACHE = {}
_TUNE_FILE = _os_env.environ.get('MSC_TUNE_CACHE')      # optional JSON file: reuse per-layer choices across processes
# per-layer choices measured on MI355X for the BASELINE.json configurations, shipped with the package (like a
# MIOpen perf-db); shapes not in it are timed on first use.  MSC_TUNE_DB=0 ignores it.
_TUNE_DB = _os_env.path.join(_os_env.path.dirname(_os_env.path.abspath(__file__)), 'tune', 'gfx950.json')


def _tune_load():
    if _TUNE_CACHE:
        return
    import json
    files = [_TUNE_DB] if _os_env.environ.get('MSC_TUNE_DB', '1') != '0' else []
    for path in files + ([_TUNE_FILE] if _TUNE_FILE else []):
        if _os_env.path.exists(path):
            with open(path) as f:
                _TUNE_CACHE.update(json.load(f))


def _tune_save():
    if _TUNE_FILE:
        import json
        tmp = _TUNE_FILE + '.tmp'
        with open(tmp, 'w') as f:
            json.dump(_TUNE_CACHE, f)
        _os_env.replace(tmp, _TUNE_FILE)


class _GroupHandle(C.c_void_p):
    """msc_wgrad_group* plus the descriptors it was made from (for FLOP accounting in bench.py)"""


# debugging aid: program buffers start as zeros instead of recycled allocator memory (bits: 1 activations, 2 vectors, 4 gradient buffers, 8 ReLU byte masks)
_ZERO_ALLOC = int(_os_env.environ.get('MSC_ZERO_ALLOC', '0'))
_TUNE_ONLY = _os_env.environ.get('MSC_TUNE_ONLY', '').split(',') if _os_env.environ.get('MSC_TUNE_ONLY') else []      # debugging aid: c0 / c1 / j


class _Builder:
    def __init__(self, net, N, H, W, training, device):
        self.net, self.N, self.H, self.W, self.training, self.dev = net, N, H, W, training, device
        self.lib = _lib.load()
        _tune_load()
        self._tuned_new = False
        self.dt, self.tdtype = net._dt, net._tdtype
        self.es = 4 if self.dt == F32 else 2
        self.tune_dt = BF16 if self.dt == F16 else self.dt      # fp16 runs the bf16 kernels' instruction stream: one set of per-layer choices
        self.prog = _Program()
        self.prog.training = training
        self.prog.fold, self.prog.fold_version = [], -1
        self.prog.grad_writes = []     # (index into bwd, device address of the gradient it writes)
        self.ops = []                 # backward emitters, forward order
        self.gbuf = {}                # id(buffer) -> grad buffer
        self.gwritten = set()         # (id(gbuf), c0, C) already written in this backward
        self.slices = {}              # id(buffer) -> set of (c0, C)
        # Weight gradients are deferred and launched MSC_WGRAD_GROUP at a time (msc_wgrad_group_*): one layer of the
        # encoder cannot fill 256 CUs, a dozen can.  Default: all of them at the end of backward in a single process;
        # 24 under torch.distributed, so that gradient buckets still complete while backward runs (trainer.ddp_plan).
        # 0/1 = one launch per layer (the CPU interpreter always does that).
        env = _os_env.environ
        multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        self.group_max = int(env.get('MSC_WGRAD_GROUP', '24' if multi else '1000')) if (training and device.type == 'cuda') else 0
        self.group_steps = int(env.get('MSC_WGRAD_GROUP_STEPS', '128'))      # k-steps per block: fewer, longer blocks = fewer fp32-atomic passes over dW
        self.group_tile = int(env.get('MSC_WGRAD_GROUP_TILE', '128'))
        self.ordered = bool(getattr(net, 'deterministic', False)) and self.group_max > 1      # fixed-order sums (MSC_WGRAD_ORDERED); the per-layer launches stay atomic
        if training and getattr(net, 'deterministic', False) and not self.ordered and device.type == 'cuda':
            # round-4 advisory: the flag used to be dropped without a word when the grouped launches are off
            raise _lib.MscError('deterministic=True needs the grouped weight-gradient launches (MSC_WGRAD_GROUP > 1, got %d): the per-layer '
                                'launches add their split-K partials with fp32 atomics in arrival order' % self.group_max)
        self.pending = []             # deferred (WgradDesc, gradient address or None)
        # round 6: the grouped launches as a PARALLEL BRANCH of the step's graph.  MSC_WGRAD_FLUSH names the points of the backward at which what
        # is pending goes out ('dec' = the decoder's backward is complete, 'l4'..'l1' = that encoder stage's): with MSC_OVERLAP_WGRAD=1 those
        # launches run on the side stream beside the latency-bound data-gradient chain of the stages that follow (_Program.run_backward)
        self.flush_at = set(f for f in env.get('MSC_WGRAD_FLUSH', _WGRAD_FLUSH_DEFAULT).split(',') if f) if self.group_max > 1 else set()
        self.gcount = {}              # activation slice -> number of launches that write its gradient
        self.gwriter = {}             # activation slice -> ConvDesc of the (mode 0, no residual) dgrad conv that wrote it first
        self.fuse_bn_bwd = _os_env.environ.get('MSC_FUSE_BN_BWD', '1') != '0'
        # BatchNorm + ReLU of a Bottleneck's bn2 applied by conv3 on load (msc_conv_desc.in_bn, ABI v9): training, 16-bit on the device (the
        # CPU interpreter follows the same launch list in fp32).  Round 5 ran it on the MI355X: kernel tests, end-to-end comparison and the batch-32 parity
        # test pass, and three same-box A/Bs put the step at -0.02 / +0.04 / +0.03...0.05 ms (profiles/r5_run1_bn_on_load_ab.txt): the 59 msc_bn_apply
        # launches it removes (-0.27 ms) come back as longer convolutions (+0.2...0.4 ms: the in_bn configurations are the eight-wave tiles, layer3's natural
        # tile is 128x64) -- NEUTRAL, so it stays opt-in: MSC_BN_ON_LOAD=1 (MSC_BN_ON_LOAD_MIN_PIXELS=n restricts it to the layers of >= n pixels)
        self.bn_on_load = (training and _os_env.environ.get('MSC_BN_ON_LOAD', '0') == '1' and (self.dt != F32 or device.type != 'cuda'))
        self.bn_on_load_min_pixels = int(_os_env.environ.get('MSC_BN_ON_LOAD_MIN_PIXELS', '0'))
        # residual joins: the data-gradient conv that ACCUMULATES the last addend of the join's gradient also reduces the sums of the
        # join's BatchNorm backward (stats_kind 1 with stats_z, ABI v6) -- no msc_bn_bwd_reduce pass over three tensors
        self.fuse_join_bwd = _os_env.environ.get('MSC_FUSE_JOIN_BWD', '1') != '0'
        self.glast = {}               # gradient slice -> ConvDesc of its most recent writer, while that is a mode-0 / stride-1 data-gradient conv
        self.bn_writer = {}           # gradient slice -> index (in prog.bwd) of the msc_bn_bwd_apply launch that wrote it as its dres
        # decoder: the data-gradient conv that writes the gradient of a bias+ReLU layer's activation first also applies that
        # layer's ReLU mask and sums its bias gradient (stats_kind 2) -- no msc_relu_bias_grad pass over the tensor
        self.fuse_relu_bwd = _os_env.environ.get('MSC_FUSE_RELU_BWD', '1') != '0'
        self.rwriter = {}             # id(activation buffer) -> (ConvDesc, c0, C) of that first writer
        self.bias_items = []          # (slots address, slot channel count, bias parameter, channels) awaiting flush_bias_slots
        self.fuse_bneck = _os_env.environ.get('MSC_FUSE_BNECK', '1') != '0'
        self.split_k = _os_env.environ.get('MSC_SPLIT_K', '1') != '0'

    # ---- memory
    def buf(self, H, W, C, dtype=None):
        t = (torch.zeros if _ZERO_ALLOC & 1 else torch.empty)((self.N, H, W, C), dtype=dtype or self.tdtype, device=self.dev)
        self.prog.bytes += t.numel() * t.element_size()
        self.prog.keep.append(t)
        return t

    def vec(self, n, dtype=torch.float32, zero=False):
        t = (torch.zeros if (zero or _ZERO_ALLOC & 2) else torch.empty)(n, dtype=dtype, device=self.dev)
        self.prog.bytes += t.numel() * t.element_size()
        self.prog.keep.append(t)
        return t

    def act(self, H, W, C):
        return Act(self.buf(H, W, C))

    def slots(self, C_):
        """device address of a fresh [BN_SLOTS][C][2] f64 block of the statistics arena (forward blocks first, zeroed by the first
        forward launch; the backward's blocks after them, zeroed by the first backward launch)"""
        n = _lib.BN_SLOTS * C_ * 2
        if self.slot_used + n > self.slot_arena.numel():
            raise RuntimeError('BatchNorm statistics arena too small')
        ptr = self.slot_arena.data_ptr() + 8 * self.slot_used
        self.slot_used += n
        return ptr

    def g(self, param):
        """gradient address of `param`; records which backward launch (the next one emitted) writes it, so the
        data-parallel step knows when a suffix of the flat gradient buffer is final (trainer.ddp_plan)"""
        ptr = self.net._g(param)
        self.prog.grad_writes.append((len(self.prog.bwd), ptr))
        return ptr

    def bias_ws(self, pixels, C):
        """shared scratch for msc_bias_grad (launches are stream-ordered, so one buffer serves all)"""
        need = self.lib.msc_bias_grad_workspace_bytes(pixels, C, self.dt) // 4
        if getattr(self, '_bias_ws', None) is None or self._bias_ws.numel() < need:
            self._bias_ws = self.vec(max(need, 1 << 20))
        return self._bias_ws.data_ptr()

    def slice(self, buf, c0, C):
        self.slices.setdefault(id(buf), set()).add((c0, C))
        return Act(buf, c0, C)

    def grad_of(self, a):
        g = self.gbuf.get(id(a.buf))
        if g is None:
            g = torch.zeros_like(a.buf) if _ZERO_ALLOC & 4 else torch.empty_like(a.buf)
            self.prog.bytes += g.numel() * g.element_size()
            self.gbuf[id(a.buf)] = g
            self.slices[id(g)] = self.slices.get(id(a.buf), set())
            self.prog.keep.append(g)
        return Act(g, a.c0, a.C)

    def grad_acc(self, a):
        """1 if the gradient slice was already written during this backward (-> accumulate), else
        marks it (and every registered sub-slice it covers) written and returns 0."""
        key = (id(a.buf), a.c0, a.C)
        self.gcount[key] = self.gcount.get(key, 0) + 1
        for k in [k for k in self.glast if k[0] == key[0] and k[1] < a.c0 + a.C and a.c0 < k[1] + k[2]]:
            del self.glast[k]         # whoever writes next is the last writer of everything it overlaps (the dgrad site re-registers itself)
        if key in self.gwritten:
            return 1
        self.gwritten.add(key)
        for (c0, C) in self.slices.get(id(a.buf), ()):
            if c0 >= a.c0 and c0 + C <= a.c0 + a.C:
                self.gwritten.add((id(a.buf), c0, C))
        return 0

    # ---- launch helpers
    def emit(self, lst, fn, *args):
        lst.append((fn, args))

    def conv_desc(self, x, wt, out, KH, KW, stride, pad, mode=0, flip=0, relu=0, scale=None, shift=None, res=None,
                  stats=None, in_hw=None, in_ld=None, cin=None, out_hw=None, want_stats=False, allow_split=False, in_bn=None):
        d = ConvDesc()
        d.in_, d.wt, d.out = x.ptr, wt.data_ptr(), out.ptr
        if in_bn is not None:          # BnInput: x is the raw output of a BatchNorm'd conv, normalised + rectified on load (ABI v9)
            self.prog.keep.append(in_bn)
            d.in_bn = C.addressof(in_bn)
            d._in_bn = in_bn
        d.res = res.ptr if res is not None else None
        d.scale, d.shift, d.stats = _p(scale), _p(shift), _p(stats)
        d.in_ld = in_ld if in_ld is not None else x.ld
        d.out_ld = out.ld
        d.res_ld = res.ld if res is not None else 0
        d.dtype, d.mode = self.dt, mode
        d.N = self.N
        d.Hi, d.Wi = in_hw if in_hw is not None else (x.H, x.W)
        d.Cin = cin if cin is not None else x.C
        d.Ho, d.Wo = out_hw if out_hw is not None else (out.H, out.W)
        d.Cout = out.C
        d.KH, d.KW, d.stride, d.pad, d.flip, d.relu = KH, KW, stride, pad, flip, relu
        d.cfg = 0
        self.prog.keep.append(d)
        if allow_split and self.split_k and self.dev.type == 'cuda' and mode == 0 and not want_stats and stats is None:
            # few output tiles, long reduction (the decoder's centre / dec5 ConvRelu on the 4x4 / 8x8 maps: 512 pixels x 512 channels x
            # 18432 = 32 tiles of 128x128 for 256 CUs): the reduction runs in slices, a finishing pass applies the epilogue
            pixels, ksteps = self.N * d.Ho * d.Wo, d.Cin * KH * KW * self.es // 128
            tiles = ((pixels + 127) // 128) * ((d.Cout + 127) // 128)
            cus = torch.cuda.get_device_properties(self.dev).multi_processor_count if self.dev.type == 'cuda' else 256
            split = min(16, cus // max(tiles, 1), ksteps // 16)
            if tiles <= 64 and split >= 2 and d.Cout % 8 == 0:
                d.splitk = split
                ws = self.vec(split * pixels * d.Cout)
                d.splitk_ws = ws.data_ptr()
        self.tune_conv(d, want_stats)
        return d

    def conv(self, lst, *a, **k):
        d = self.conv_desc(*a, **k)
        self.emit(lst, self.lib.msc_conv_igemm, C.byref(d))
        return d

    def wgrad(self, lst, p, q, dw, KH, KW, stride, pad, q_hw=None, q_ld=None, B=None):
        d = WgradDesc()
        d.p, d.q, d.dw = p.ptr, q.ptr, dw if isinstance(dw, int) else dw.data_ptr()
        d.p_ld = p.ld
        d.q_ld = q_ld if q_ld is not None else q.ld
        d.dtype = self.dt
        d.N, d.Hp, d.Wp, d.A = self.N, p.H, p.W, p.C
        d.Hq, d.Wq = q_hw if q_hw is not None else (q.H, q.W)
        d.B = B if B is not None else q.C
        d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
        d.cfg = 0
        self.prog.keep.append(d)
        if self.group_max > 1:
            gw = None
            writes = self.prog.grad_writes
            if writes and writes[-1] == (len(self.prog.bwd), d.dw):      # recorded by g(): moves to the group launch
                gw = writes.pop()[1]
            self.pending.append((d, gw))
            if len(self.pending) >= self.group_max:
                self.flush_wgrads()
            return
        self.tune_wgrad(d)
        self.emit(lst, self.lib.msc_conv_wgrad, C.byref(d))

    def flush_wgrads(self):
        if not self.pending:
            return
        n = len(self.pending)
        arr = (WgradDesc * n)(*[d for d, _ in self.pending])
        h = _GroupHandle()
        _lib.check(self.lib.msc_wgrad_group_create(arr, n, self.group_steps, self.group_tile, _lib.WGRAD_ORDERED if self.ordered else 0, C.byref(h)),
                   'msc_wgrad_group_create')
        h.descs = [d for d, _ in self.pending]
        h.ordered = bool(self.ordered)
        self.prog.groups.append(h)
        idx = len(self.prog.bwd)
        self.emit(self.prog.bwd, self.lib.msc_wgrad_group_run, h)
        self.prog.grad_writes += [(idx, gw) for _, gw in self.pending if gw is not None]
        self.pending = []

    # ---- per-layer kernel configuration (like cuDNN's benchmark mode): time the valid configurations once per
    # distinct layer shape on the real buffers and keep the fastest; results are cached process-wide
    def _time(self, fn, dref, reps=5):
        stream = _stream_of(self.dev)
        if fn(dref, stream) != 0:
            return None
        best = 1e30
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn(dref, stream)
            b.record()
            b.synchronize()
            best = min(best, a.elapsed_time(b))
        return best

    def tune_conv(self, d, want_stats):
        if not self.net.autotune or self.dev.type != 'cuda':
            return
        if _TUNE_ONLY and ('c%d' % int(bool(d.flip))) not in _TUNE_ONLY and ('m%d' % d.mode) not in _TUNE_ONLY:      # debugging aid
            return
        key = repr(('c', self.tune_dt, d.mode, d.flip, d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad,
                    bool(d.res), want_stats, d.in_ld, d.out_ld, bool(d.relu)) + ((int(d.splitk),) if d.splitk > 1 else ()) + (('in_bn',) if d.in_bn else ()))
        cache = _TUNE_CACHE
        lib = self.lib
        if key not in cache or (cache[key] and not lib.msc_conv_cfg_ok(C.byref(d), int(cache[key]))):
            best, best_t = 0, 1e30
            bi = getattr(d, '_in_bn', None)
            if bi is not None:         # the timed launches must not move the layer's running statistics (the first block of every launch would)
                keep_rs = (bi.running_mean, bi.running_var)
                bi.running_mean, bi.running_var = None, None
            for c in range(1, lib.msc_conv_num_cfgs() + 1):
                if not lib.msc_conv_cfg_ok(C.byref(d), c):
                    continue
                d.cfg = c
                if want_stats:
                    need = lib.msc_conv_stats_slices(C.byref(d)) * d.Cout * 2
                    if getattr(self, '_tune_stats', None) is None or self._tune_stats.numel() < need:
                        self._tune_stats = torch.empty(max(need, 1 << 22), dtype=torch.float64, device=self.dev)
                    d.stats = self._tune_stats.data_ptr()
                t = self._time(lib.msc_conv_igemm, C.byref(d))
                if t is not None and t < best_t:
                    best, best_t = c, t
            d.stats = None
            if bi is not None:
                bi.running_mean, bi.running_var = keep_rs
            cache[key] = best
            self._tuned_new = True
        d.cfg = cache[key]

    def tune_join(self, d):
        """the configuration of a data-gradient conv that carries a residual join's reductions (stats_kind 1 with stats_z): its epilogue
        reads three more tensors than the launch tune_conv timed, which shifts the best tile towards more, smaller blocks"""
        if not self.net.autotune or self.dev.type != 'cuda' or _os_env.environ.get('MSC_TUNE_JOIN', '1') == '0':
            return
        if _TUNE_ONLY and 'j' not in _TUNE_ONLY:
            return
        key = repr(('j', self.tune_dt, d.N, d.Hi, d.Wi, d.Cin, d.Cout, d.KH, d.KW, bool(d.res), d.in_ld, d.out_ld) +
                   ((d.flip, d.pad) if (d.flip, d.pad) != (1, 0) else ()) + (('t', d.stride) if d.mode else ()))      # (the shipped db holds the 1x1 data-gradient keys without the pair)
        cache, lib = _TUNE_CACHE, self.lib
        if key not in cache or (cache[key] and not lib.msc_conv_cfg_ok(C.byref(d), int(cache[key]))):
            best, best_t, keep = 0, 1e30, d.cfg
            # the launch accumulates in place (res == out): timed on a scratch copy of its output / addend so that the 1 + 5 repetitions per
            # configuration neither grow the values of a live gradient buffer to Inf nor time the kernel on degenerate data (the slots
            # it adds into are reset by the step itself)
            res0, out0 = d.res, d.out
            scratch = torch.zeros(((d.N * d.Ho * d.Wo - 1) * max(d.out_ld, d.res_ld) + d.Cout,), dtype=self.tdtype, device=self.dev)
            if res0 and res0 == out0:
                d.res = d.out = scratch.data_ptr()
            for c in range(1, lib.msc_conv_num_cfgs() + 1):
                if not lib.msc_conv_cfg_ok(C.byref(d), c):
                    continue
                d.cfg = c
                scratch.zero_()
                t = self._time(lib.msc_conv_igemm, C.byref(d), reps=3)
                if t is not None and t < best_t:
                    best, best_t = c, t
            d.res, d.out = res0, out0
            del scratch
            d.cfg = keep
            cache[key] = best
            self._tuned_new = True
        if cache[key]:
            d.cfg = cache[key]

    def tune_wgrad(self, d):
        if not self.net.autotune or self.dev.type != 'cuda':
            return
        key = repr(('w2', self.tune_dt, d.N, d.Hp, d.Wp, d.A, d.Hq, d.Wq, d.B, d.KH, d.KW, d.stride, d.pad, d.p_ld, d.q_ld))
        cache = _TUNE_CACHE
        big_ok = d.A % 128 == 0 and d.B % 128 == 0
        ncfg = self.lib.msc_conv_wgrad_num_cfgs()
        if key not in cache or not (0 <= int(cache[key]) <= ncfg):
            best, best_t = 0, 1e30
            for c in range(1 if big_ok else 6, ncfg + 1):
                d.cfg = c
                t = self._time(self.lib.msc_conv_wgrad, C.byref(d))
                if t is not None and t < best_t:
                    best, best_t = c, t
            cache[key] = best
            self._tuned_new = True
        d.cfg = cache[key]

    # ---- layers -------------------------------------------------------------------------------
    def conv_bn(self, name, x, conv, bn, stride, relu, out, res=None, stem=None, defer=False, pend=None):
        """conv (no bias) + BatchNorm2d (+ residual) (+ ReLU) -> out.  `stem`: (Hp, Wp) of the prepared
        NHWC4 image when this is encoder.conv1 expressed as a 7-tap x 32-wide implicit GEMM.
        defer (training, round 4 / ABI v9): the msc_bn_apply launch is NOT emitted -- the returned BnInput describes the layer (raw output,
        statistics slots, where the coefficients and the activation `out` go) and the consuming conv_bn(..., pend=that) applies
        relu(scale*y + shift) to its operand on load and stores `out` on the way (the weight gradient of the consumer reads it).
        Returns None when the apply was emitted here."""
        net, lib, P, fwd = self.net, self.lib, self.prog, self.prog.fwd
        w = net._pack['w'][name]
        cout = conv.out_channels
        if stem is not None:
            geo = dict(KH=7, KW=1, stride=2, pad=0, in_hw=stem, in_ld=4, cin=32)
        else:
            k = conv.kernel_size[0]
            geo = dict(KH=k, KW=k, stride=stride, pad=conv.padding[0])
        scale, shift = self.vec(cout), self.vec(cout)
        if not self.training:
            # eval: BN folds into the conv epilogue (scale = gamma/sqrt(rv+eps), shift = beta - rm*scale)
            self.emit(P.fold, lib.msc_bn_fold, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                      bn.running_var.data_ptr(), BN_EPS, scale.data_ptr(), shift.data_ptr(), cout)
            self.conv(fwd, x, w, out, relu=int(relu), scale=scale, shift=shift, res=res, **geo)
            return
        y = self.act(out.H, out.W, cout)
        if pend is not None:           # the producer's BatchNorm + ReLU ride on this conv's operand fetch; `x` (its activation) is written on the way
            d = self.conv_desc(Act(pend._y), w, y, want_stats=True, in_bn=pend, **geo)
            if self.dev.type == 'cuda' and not any(lib.msc_conv_cfg_ok(C.byref(d), c) for c in range(1, lib.msc_conv_num_cfgs() + 1)):
                # no kernel configuration applies BatchNorm on load for this layer shape: the producer's msc_bn_apply runs as its own
                # launch after all (on load is opt-in, MSC_BN_ON_LOAD=1; an unusual shape must not fail with it)
                self.emit(fwd, *pend._apply)
                self.prog.on_load_fallbacks = getattr(self.prog, 'on_load_fallbacks', 0) + 1
                d = self.conv_desc(x, w, y, want_stats=True, **geo)
        else:
            d = self.conv_desc(x, w, y, want_stats=True, **geo)
        # batch statistics: the conv epilogue adds (sum, sum of squares) into one slot per XCD; bn_apply sums the slots in its
        # prologue (no finalize launch).  The slots of every layer live in one arena zeroed once per step.
        d.stats = self.slots(cout)
        self.emit(fwd, lib.msc_conv_igemm, C.byref(d))
        mean, invstd = self.vec(cout), self.vec(cout)
        count = self.N * out.H * out.W
        # residual joins: the ReLU sits after the add, so the backward needs [out > 0] -- twice (the BatchNorm backward and the epilogue of
        # the data-gradient conv that carries the join's sums).  msc_bn_apply leaves it as one byte per 16-byte channel vector: 1/16 of the
        # activation's bytes on both reads (round 4; MSC_RELU_BITS=0: the backward reads the activation)
        rmask = None
        if res is not None and relu and _os_env.environ.get('MSC_RELU_BITS', '1') != '0':
            rmask = (torch.zeros if _ZERO_ALLOC & 8 else torch.empty)((self.N, out.H, out.W, cout * self.es // 16), dtype=torch.uint8, device=self.dev)
            self.prog.bytes += rmask.numel()
            self.prog.keep.append(rmask)
        bi = None
        if defer and res is None and relu and stem is None:
            bi = BnInput()
            bi.slots, bi.count = d.stats, count
            bi.gamma, bi.beta, bi.eps, bi.momentum = bn.weight.data_ptr(), bn.bias.data_ptr(), BN_EPS, BN_MOMENTUM
            bi.running_mean, bi.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            bi.scale, bi.shift, bi.save_mean, bi.save_invstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()
            bi.out, bi.out_ld = out.ptr, out.ld
            bi._y = y.buf if y.c0 == 0 and y.C == y.buf.shape[3] else None
            assert bi._y is not None
            # the launch this replaces, kept for a consumer no in_bn configuration takes
            bi._apply = (lib.msc_bn_apply, y.ptr, y.ld, None, 0, out.ptr, out.ld, d.stats, count, bn.weight.data_ptr(), bn.bias.data_ptr(), BN_EPS,
                         BN_MOMENTUM, bn.running_mean.data_ptr(), bn.running_var.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                         invstd.data_ptr(), None, 0, int(relu), self.dt, count, cout)
        else:
            self.emit(fwd, lib.msc_bn_apply, y.ptr, y.ld, res.ptr if res is not None else None, res.ld if res is not None else 0,
                      out.ptr, out.ld, d.stats, count, bn.weight.data_ptr(), bn.bias.data_ptr(), BN_EPS, BN_MOMENTUM,
                      bn.running_mean.data_ptr(), bn.running_var.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                      invstd.data_ptr(), rmask.data_ptr() if rmask is not None else None, rmask.shape[3] if rmask is not None else 0, int(relu), self.dt,
                      count, cout)
        self.ops.append(lambda: self._conv_bn_bwd(name, x, conv, bn, geo, relu, out, res, y, mean, invstd, count, stem, scale, shift, rmask))
        return bi

    def _conv_bn_bwd(self, name, x, conv, bn, geo, relu, out, res, y, mean, invstd, count, stem, scale, shift, rmask=None):
        net, lib, P, bwd = self.net, self.lib, self.prog, self.prog.bwd
        cout = conv.out_channels
        dout = self.grad_of(out)
        # ReLU mask: without a residual the pre-activation is scale*y + shift, recomputed from the y both kernels read
        # anyway (mode 2) instead of reading `out` (mode 1)
        mask = 0 if not relu else (1 if res is not None else 2)
        gkey = (id(out.buf), out.c0, out.C)      # grad_acc() keys gradient slices by their activation
        wd = self.gwriter.get(gkey) if (self.fuse_bn_bwd and mask != 1 and self.gcount.get(gkey, 0) == 1) else None
        bslots = self.slots(cout)                # (sum dh, sum dh*y) per XCD slot, summed in bn_bwd_apply's prologue
        wl = self.glast.get(gkey) if (wd is None and mask == 1 and self.fuse_bn_bwd and self.fuse_join_bwd) else None
        if wl is not None and not wl.stats and not wl.relu and not wl.scale and not wl.shift and wl.Cout == cout:
            # residual join: the last writer of dout accumulated onto the other addend(s) (out = acc + res); its epilogue reduces
            # dh = out * [block output > 0] against y -- the block output is the ReLU output that follows the add
            wl.stats_kind, wl.stats_y, wl.stats_y_ld = 1, y.ptr, y.ld
            if rmask is not None:
                wl.stats_z, wl.stats_z_ld, wl.stats_z_bits = rmask.data_ptr(), rmask.shape[3], 1
            else:
                wl.stats_z, wl.stats_z_ld = out.ptr, out.ld
            wl.stats = bslots
            self.tune_join(wl)
            if wl.cfg and not lib.msc_conv_cfg_ok(C.byref(wl), int(wl.cfg)):
                wl.cfg = 0
        elif wd is not None:
            # dout has exactly one writer, a data-gradient conv that ran earlier in this backward: its epilogue also
            # accumulates (sum dh, sum dh*y), so the column-reduce pass over dout and y is not launched
            wd.stats_kind, wd.stats_y, wd.stats_y_ld = 1, y.ptr, y.ld
            wd.scale, wd.shift = (scale.data_ptr(), shift.data_ptr()) if mask == 2 else (None, None)
            wd.stats = bslots
            if wd.cfg and not lib.msc_conv_cfg_ok(C.byref(wd), int(wd.cfg)):
                wd.cfg = 0                                   # the tuned configuration cannot carry the statistics: heuristic one
        elif (mask == 0 and gkey in self.bn_writer and self.gcount.get(gkey, 0) == 1 and self.fuse_bn_bwd and self.dev.type != 'meta'
              and _os_env.environ.get('MSC_FUSE_BN_BWD_RES', '1') != '0'):
            # dout was written by ONE launch, the msc_bn_bwd_apply of the block's bn3 (as its dres): that launch also reduces this layer's sums
            i = self.bn_writer[gkey]
            fn, a = bwd[i]
            bwd[i] = (fn, a[:21] + (y.ptr, y.ld, bslots) + a[24:])
        else:
            self.emit(bwd, lib.msc_bn_bwd_reduce, dout.ptr, dout.ld, out.ptr, out.ld, y.ptr, y.ld, mask, scale.data_ptr(),
                      shift.data_ptr(), bslots, self.dt, count, cout)
        dres_ptr, dres_ld, dres_acc = None, 0, 0
        if res is not None:
            gres = self.grad_of(res)
            dres_acc = self.grad_acc(res)
            dres_ptr, dres_ld = gres.ptr, gres.ld
        # dy overwrites y in place (each element is read, then written, by the same lane); the prologue also adds dgamma / dbeta
        gw, gb = self.g(bn.weight), self.g(bn.bias)
        if res is not None and not dres_acc:
            # first writer of the residual's gradient: if the residual is a BatchNorm output (the downsample branch), that layer's backward can
            # have its sums from this launch (msc_bn_bwd_apply res_y / res_slots) -- remembered here, patched in by that layer's backward below
            self.bn_writer[(id(res.buf), res.c0, res.C)] = len(bwd)
        mptr, mld, mmode = (rmask.data_ptr(), rmask.shape[3], 3) if (rmask is not None and mask == 1) else (out.ptr, out.ld, mask)
        self.emit(bwd, lib.msc_bn_bwd_apply, dout.ptr, dout.ld, mptr, mld, y.ptr, y.ld, mmode, scale.data_ptr(),
                  shift.data_ptr(), bslots, count, bn.weight.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gw, gb,
                  y.ptr, y.ld, dres_ptr, dres_ld, dres_acc, None, 0, None, self.dt, count, cout)
        dy = y
        if stem is not None:
            self.wgrad(bwd, dy, x, P.stem_dw.data_ptr(), 7, 1, 2, 0, q_hw=stem, q_ld=4, B=32)
            self.flush_wgrads()
            self.emit(bwd, lib.msc_stem_unpack_grad, P.stem_dw.data_ptr(), self.g(conv.weight), 64)
            return                                    # the network input needs no gradient
        self.wgrad(bwd, dy, x, self.g(conv.weight), geo['KH'], geo['KW'], geo['stride'], geo['pad'])
        gx = self.grad_of(x)
        acc = self.grad_acc(x)
        wt = net._pack['wt'][name]
        k = geo['KH']
        if geo['stride'] == 1:
            d = self.conv(bwd, dy, wt, gx, KH=k, KW=k, stride=1, pad=geo['pad'], flip=1, res=gx if acc else None)
            if not acc:
                self.gwriter[(id(x.buf), x.c0, x.C)] = d
            self.glast[(id(x.buf), x.c0, x.C)] = d
        else:
            d = self.conv(bwd, dy, wt, gx, KH=k, KW=k, stride=2, pad=geo['pad'], mode=1, res=gx if acc else None)
            # round 4: a transposed-mode data gradient (the stride-2 3x3 of a stage's first block) that is the only writer of its output
            # can carry the BatchNorm-backward sums of the layer before it as well (the epilogue is the same code in both modes)
            if _os_env.environ.get('MSC_FUSE_BN_BWD_T', '1') != '0':
                if not acc:
                    self.gwriter[(id(x.buf), x.c0, x.C)] = d
                self.glast[(id(x.buf), x.c0, x.C)] = d      # ... or, as the LAST writer of a stage's output gradient (the downsample branch), a join's

    def stem_bn_pool(self, name, x, conv, bn, pooled, stem):
        """encoder.conv1 (7-tap x 32-wide implicit GEMM on the prepared image) + BatchNorm2d (batch statistics) + ReLU + MaxPool2d(2,2)
        -> pooled (src/unet_models.py:360-363)"""
        net, lib, P, fwd = self.net, self.lib, self.prog, self.prog.fwd
        cout = conv.out_channels
        y = self.act(pooled.H * 2, pooled.W * 2, cout)
        d = self.conv_desc(x, net._pack['w'][name], y, want_stats=True, KH=7, KW=1, stride=2, pad=0, in_hw=stem, in_ld=4, cin=32)
        d.stats = self.slots(cout)
        self.emit(fwd, lib.msc_conv_igemm, C.byref(d))
        scale, shift, mean, invstd = self.vec(cout), self.vec(cout), self.vec(cout), self.vec(cout)
        count = self.N * y.H * y.W
        self.emit(fwd, lib.msc_bn_apply_pool, y.ptr, y.ld, pooled.ptr, pooled.ld, d.stats, count, bn.weight.data_ptr(), bn.bias.data_ptr(), BN_EPS,
                  BN_MOMENTUM, bn.running_mean.data_ptr(), bn.running_var.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                  invstd.data_ptr(), self.dt, self.N, pooled.H, pooled.W, cout)

        def bwd():
            dpool = self.grad_of(pooled)
            bslots = self.slots(cout)
            self.emit(P.bwd, lib.msc_bn_pool_bwd_reduce, dpool.ptr, dpool.ld, y.ptr, y.ld, scale.data_ptr(), shift.data_ptr(), bslots, self.dt,
                      self.N, pooled.H, pooled.W, cout)
            gw, gb = self.g(bn.weight), self.g(bn.bias)
            self.emit(P.bwd, lib.msc_bn_pool_bwd_apply, dpool.ptr, dpool.ld, y.ptr, y.ld, scale.data_ptr(), shift.data_ptr(), bslots, count,
                      bn.weight.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gw, gb, self.dt, self.N, pooled.H, pooled.W, cout)
            self.wgrad(P.bwd, y, x, P.stem_dw.data_ptr(), 7, 1, 2, 0, q_hw=stem, q_ld=4, B=32)      # dy overwrote y in place
            self.flush_wgrads()
            self.emit(P.bwd, lib.msc_stem_unpack_grad, P.stem_dw.data_ptr(), self.g(conv.weight), 64)
        self.ops.append(bwd)

    def bottleneck_fused(self, base, x, blk, out):
        """Eval mode, 16-bit compute: the identity Bottleneck `blk` (conv1x1-bn-relu, conv3x3-bn-relu, conv1x1-bn, + x, relu) as ONE
        launch (msc_bottleneck_fused; csrc/bottleneck.hip) when the kernel takes the shape -- the intermediates never leave
        LDS.  Returns False when the block has to run as three convolutions (training: batch statistics sit between them;
        fp32 parity mode; 512-channel stage; map width not a multiple of 16).  MSC_FUSE_BNECK=0 switches it off (A/B)."""
        if self.training or self.dev.type != 'cuda' or self.dt == F32 or not self.fuse_bneck:
            return False
        if blk.downsample is not None or blk.stride != 1 or x.C != out.C or x.C != 4 * blk.conv1.out_channels:
            return False
        net, lib, P = self.net, self.lib, self.prog
        cmid = blk.conv1.out_channels
        d = BneckDesc()
        d.x, d.out, d.x_ld, d.out_ld = x.ptr, out.ptr, x.ld, out.ld
        d.dtype, d.N, d.H, d.W, d.Cmid, d.cfg = self.dt, self.N, x.H, x.W, cmid, 0
        if not lib.msc_bottleneck_ok(C.byref(d)):
            return False
        coef = []
        for bn in (blk.bn1, blk.bn2, blk.bn3):
            sc, sh = self.vec(bn.num_features), self.vec(bn.num_features)
            self.emit(P.fold, lib.msc_bn_fold, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                      bn.running_var.data_ptr(), BN_EPS, sc.data_ptr(), sh.data_ptr(), bn.num_features)
            coef += [sc, sh]
        wpk = self.vec(int(lib.msc_bottleneck_pack_bytes(cmid)), dtype=torch.uint8)
        w = net._pack['w']
        # the fragment-major weight stream is rebuilt with the folded coefficients whenever the master weights changed
        self.emit(P.fold, lib.msc_bottleneck_pack, w[base + '.conv1'].data_ptr(), w[base + '.conv2'].data_ptr(), w[base + '.conv3'].data_ptr(),
                  wpk.data_ptr(), cmid, self.dt)
        d.wpk = wpk.data_ptr()
        d.scale1, d.shift1, d.scale2, d.shift2, d.scale3, d.shift3 = [t.data_ptr() for t in coef]
        P.keep.append(d)
        if net.autotune:                  # patch rows: time the instantiations that take this shape (like tune_conv)
            key = repr(('b', self.tune_dt, self.N, x.H, x.W, cmid, x.ld, out.ld))
            if key not in _TUNE_CACHE:
                best, best_t = 0, 1e30
                for cfg in (2, 4, 8):
                    d.cfg = cfg
                    if lib.msc_bottleneck_ok(C.byref(d)):
                        t = self._time(lib.msc_bottleneck_fused, C.byref(d))
                        if t is not None and t < best_t:
                            best, best_t = cfg, t
                _TUNE_CACHE[key] = best
                self._tuned_new = True
            d.cfg = int(_TUNE_CACHE[key])
            if not lib.msc_bottleneck_ok(C.byref(d)):
                d.cfg = 0
        self.emit(P.fwd, lib.msc_bottleneck_fused, C.byref(d))
        return True

    def _relu_writer(self, x, d, acc):
        """remember the data-gradient conv `d` as the first (non-accumulating) writer of grad(x)"""
        if self.fuse_relu_bwd and not acc and self.dev.type != 'meta':
            self.rwriter[id(x.buf)] = (d, x.c0, x.C)

    def _fused_relu_bwd(self, out, bias):
        """If the gradient of `out` (the activation of a bias+ReLU layer) was written by exactly one data-gradient conv, make
        that launch mask it by [out > 0] and sum the bias gradient (stats_kind 2; include/msc.h) and emit the tiny finalize
        that folds the slots into `bias`'s gradient.  The writer may cover a wider channel range than `out` (the whole
        concat buffer): masking the skip half by ITS activation is idempotent with the mask its own backward applies
        (every skip is the ReLU output of a residual block, mask mode 1 in _conv_bn_bwd).  Returns True when hooked."""
        w = self.rwriter.get(id(out.buf))
        if w is None:
            return False
        d, c0, C_ = w
        key = (id(out.buf), c0, C_)
        if d.stats or self.gcount.get(key, 0) != 1 or not (c0 <= out.c0 and out.c0 + out.C <= c0 + C_):
            return False
        full = Act(out.buf, c0, C_)
        d.stats_kind, d.stats_y, d.stats_y_ld = 2, full.ptr, full.ld
        d.stats = self.slots(C_)
        if not self.lib.msc_conv_cfg_ok(C.byref(d), int(d.cfg)) and d.cfg:
            d.cfg = 0                                    # the tuned configuration cannot carry statistics: heuristic one
        self.bias_items.append((d.stats + 16 * (out.c0 - c0), C_, bias, out.C))
        return True

    def flush_bias_slots(self):
        """ONE launch folds the bias-gradient slots of all decoder layers hooked so far into their gradients (emitted after the
        last decoder layer's backward: the decoder's gradients sit at the tail of the flat buffer and complete first)"""
        items, self.bias_items = self.bias_items, []
        for i in range(0, len(items), _lib.BIAS_SLOTS_MAX):
            part = items[i:i + _lib.BIAS_SLOTS_MAX]
            arr = (_lib.BiasSlotsItem * len(part))()
            for a, (slots, cs, bias, c_) in zip(arr, part):
                a.slots, a.Cs, a.db, a.C = slots, cs, self.g(bias), c_
            self.prog.keep.append(arr)
            self.emit(self.prog.bwd, self.lib.msc_bias_slots_finalize_multi, arr, len(part))

    def conv_relu(self, name, x, conv, out):
        """ConvRelu (src/unet_models.py:25-34): 3x3, pad 1, bias, ReLU."""
        net, lib, P = self.net, self.lib, self.prog
        self.conv(P.fwd, x, net._pack['w'][name], out, KH=3, KW=3, stride=1, pad=1, relu=1, shift=conv.bias, allow_split=True)
        if self.training:
            self.ops.append(lambda: self._conv_relu_bwd(name, x, conv, out))

    def _conv_relu_bwd(self, name, x, conv, out, masked=False):
        net, lib, P, bwd = self.net, self.lib, self.prog, self.prog.bwd
        dout = self.grad_of(out)
        count = out.pixels
        if masked == 'done':
            pass             # mask and bias gradient came with the gradient itself (msc_final_bwd)
        elif not masked:
            # dmask = dout * [out > 0] and the bias gradient: by the conv that wrote dout, or in place in one pass over it
            # (`out` has exactly one consumer)
            if not self._fused_relu_bwd(out, conv.bias):
                self.emit(bwd, lib.msc_relu_bias_grad, dout.ptr, dout.ld, out.ptr, out.ld, dout.ptr, dout.ld, self.g(conv.bias),
                          self.bias_ws(count, out.C), self.dt, count, out.C)
        else:
            self.emit(bwd, lib.msc_bias_grad, dout.ptr, dout.ld, self.g(conv.bias), self.bias_ws(count, out.C), self.dt, count, out.C)
        self.wgrad(bwd, dout, x, self.g(conv.weight), 3, 3, 1, 1)
        gx = self.grad_of(x)
        acc = self.grad_acc(x)
        d = self.conv(bwd, dout, net._pack['wt'][name], gx, KH=3, KW=3, stride=1, pad=1, flip=1, res=gx if acc else None)
        self._relu_writer(x, d, acc)

    def deconv_relu(self, name, x, deconv, out):
        """ConvTranspose2d(k4,s2,p1)+bias+ReLU (src/unet_models.py:138-140) as 4 output-parity phases of 2x2 taps."""
        net, P = self.net, self.prog
        self.conv(P.fwd, x, net._pack['w'][name], out, KH=4, KW=4, stride=2, pad=1, mode=1, relu=1, shift=deconv.bias)
        if self.training:
            self.ops.append(lambda: self._deconv_relu_bwd(name, x, deconv, out))

    def _deconv_relu_bwd(self, name, x, deconv, out):
        net, lib, P, bwd = self.net, self.lib, self.prog, self.prog.bwd
        dout = self.grad_of(out)
        count = out.pixels
        if self._fused_relu_bwd(out, deconv.bias):
            dm = dout                                    # already masked by its writer; read in place (a slice of the concat)
        else:
            dm = self.act(out.H, out.W, out.C)           # compact masked gradient (dout may be a slice of a concat)
            self.emit(bwd, lib.msc_relu_bias_grad, dout.ptr, dout.ld, out.ptr, out.ld, dm.ptr, dm.ld, self.g(deconv.bias),
                      self.bias_ws(count, out.C), self.dt, count, out.C)
        # dW[ci][kh][kw][co] = sum_coarse x[c][ci] * dm[2c-1+k][co]
        self.wgrad(bwd, x, dm, self.g(deconv.weight), 4, 4, 2, 1)
        gx = self.grad_of(x)
        acc = self.grad_acc(x)
        d = self.conv(bwd, dm, net._pack['wt'][name], gx, KH=4, KW=4, stride=2, pad=1, res=gx if acc else None)
        self._relu_writer(x, d, acc)

    def maxpool(self, x, out):
        lib, P = self.lib, self.prog
        self.emit(P.fwd, lib.msc_maxpool2_fwd, x.ptr, x.ld, out.ptr, out.ld, self.dt, self.N, out.H, out.W, out.C)
        if self.training:
            def bwd():
                dout, gx = self.grad_of(out), self.grad_of(x)
                acc = self.grad_acc(x)
                self.emit(P.bwd, lib.msc_maxpool2_bwd, dout.ptr, dout.ld, x.ptr, x.ld, gx.ptr, gx.ld, self.dt, self.N,
                          out.H, out.W, out.C, acc)
            self.ops.append(bwd)

    # ---- the network (src/unet_models.py:385-403) -----------------------------------------------
    def build(self):
        net, lib, P = self.net, self.lib, self.prog
        N, H, W = self.N, self.H, self.W
        nf, bottom = net._nf, net._bottom
        enc = net.encoder
        exp = 1 if net._kind == 'basic' else 4
        P.x_in = torch.empty((N, 3, H, W), dtype=torch.float32, device=self.dev)
        P.logits = torch.empty((N, 2, H, W), dtype=torch.float32, device=self.dev)
        P.probs = torch.empty((N, 2, H, W), dtype=torch.float32, device=self.dev)
        P.dlogits = torch.empty((N, 2, H, W), dtype=torch.float32, device=self.dev) if self.training else None
        P.stem_dw = torch.zeros(64 * 7 * 32, dtype=torch.float32, device=self.dev)
        if self.training:
            # forward + backward sums of every BatchNorm layer of the encoder: [BN_SLOTS][C][2] floats each
            nbn = sum(m.num_features for m in enc.modules() if isinstance(m, nn.BatchNorm2d))
            # ... plus the bias-gradient sums of the decoder's ReLU layers (stats_kind 2; at most every decoder channel once)
            ndec = 2 * (nf * 8 * 4 + nf * 2 + 960 * exp) + 4096
            self.slot_arena, self.slot_used = self.vec((2 * nbn + ndec) * _lib.BN_SLOTS * 2, dtype=torch.float64, zero=True), 0
            self.emit(P.fwd, lib.msc_memset_zero, self.slot_arena.data_ptr(), self.slot_arena.numel() * 8)

        # concat buffers: [decoder part | encoder skip]; encoder stages write straight into their slice
        cat2 = self.buf(H // 4, W // 4, nf * 2 + 64 * exp)
        cat3 = self.buf(H // 8, W // 8, nf * 8 + 128 * exp)
        cat4 = self.buf(H // 16, W // 16, nf * 8 + 256 * exp)
        cat5 = self.buf(H // 32, W // 32, nf * 8 + 512 * exp)
        skips = {1: self.slice(cat2, nf * 2, 64 * exp), 2: self.slice(cat3, nf * 8, 128 * exp),
                 3: self.slice(cat4, nf * 8, 256 * exp), 4: self.slice(cat5, nf * 8, 512 * exp)}

        # stem: conv7x7/2 + BN + ReLU + MaxPool2d(2,2)   (:360-363)
        xp = torch.zeros((N, H + 6, W + 8, 4), dtype=self.tdtype, device=self.dev)
        P.bytes += xp.numel() * xp.element_size()
        self.emit(P.fwd, lib.msc_stem_prepare, P.x_in.data_ptr(), xp.data_ptr(), self.dt, N, H, W)
        cur = self.act(H // 4, W // 4, 64)
        if self.training and _os_env.environ.get('MSC_FUSE_STEM_POOL', '1') != '0':
            # training: BatchNorm + ReLU + MaxPool in one pass over the raw conv output (msc_bn_apply_pool, ABI v7): the full-resolution
            # activation is read by nothing but the pool, so it is not stored; the backward routes the pooled gradient from y alone
            s1 = None
            self.stem_bn_pool('encoder.conv1', Act(xp), enc.conv1, enc.bn1, cur, stem=(H + 6, W + 8))
        else:
            s1 = self.act(H // 2, W // 2, 64)
            self.conv_bn('encoder.conv1', Act(xp), enc.conv1, enc.bn1, 2, True, s1, stem=(H + 6, W + 8))
            self.maxpool(s1, cur)

        # encoder.layer1-4 (torchvision BasicBlock / Bottleneck)
        for li in range(1, 5):
            blocks = getattr(enc, 'layer%d' % li)
            if self.training and ('l%d' % li) in self.flush_at:
                self.ops.append(self.flush_wgrads)      # backward runs the ops in reverse: after this stage's backward
            for bi, blk in enumerate(blocks):
                base = 'encoder.layer%d.%d' % (li, bi)
                s = blk.stride
                hh, ww = cur.H, cur.W
                ho, wo = hh // s, ww // s
                planes = blk.conv1.out_channels
                last = bi == len(blocks) - 1
                out = skips[li] if last else self.act(ho, wo, planes * exp)
                idt = cur
                if blk.downsample is not None:
                    idt = self.act(ho, wo, planes * exp)
                    self.conv_bn(base + '.downsample.0', cur, blk.downsample[0], blk.downsample[1], s, False, idt)
                if net._kind == 'basic':
                    a = self.act(ho, wo, planes)
                    self.conv_bn(base + '.conv1', cur, blk.conv1, blk.bn1, s, True, a)
                    self.conv_bn(base + '.conv2', a, blk.conv2, blk.bn2, 1, True, out, res=idt)
                elif self.bottleneck_fused(base, cur, blk, out):
                    pass
                else:
                    # bn1 / bn2 + ReLU applied by the consuming conv on load instead of msc_bn_apply launches (ABI v9; MSC_BN_ON_LOAD=1): conv3 (1x1,
                    # stride 1) always, conv2 where it is a stride-1 3x3 on a map the halo-tile kernel takes (8 x 16 pixel patches).  Measured for
                    # conv3: +1.3-2.5 us against 3.8-9 us of the launch (profiles/r4_run28_bn_on_load_probe.txt)
                    # layer3's 8 k-pixel maps are where it costs most (conv3 14.6 -> 21.8 us against the 3.8 us launch it saves,
                    # profiles/r5_run6_rocprofv3_kernel_stats_train_bn_on_load_all.txt); restricted to layer1 / layer2 (MSC_BN_ON_LOAD_MIN_PIXELS=30000) it is as neutral
                    on_load = self.bn_on_load and planes <= 512 and self.N * hh * ww >= self.bn_on_load_min_pixels
                    a = self.act(hh, ww, planes)
                    pend1 = self.conv_bn(base + '.conv1', cur, blk.conv1, blk.bn1, 1, True, a,
                                         defer=on_load and s == 1 and ww % 16 == 0 and hh % 8 == 0 and _os_env.environ.get('MSC_BN_ON_LOAD_3X3', '1') != '0')
                    b = self.act(ho, wo, planes)
                    # MSC_BN_ON_LOAD_1X1=0: conv2 (the 3x3 halo kernel: the pass is amortised over nine taps, on its natural tile) applies bn1 on load,
                    # conv3 reads a materialised activation (its pass costs more than the launch it saves, profiles/r5_run1_bn_on_load_ab.txt)
                    pend2 = self.conv_bn(base + '.conv2', a, blk.conv2, blk.bn2, s, True, b,
                                         defer=on_load and _os_env.environ.get('MSC_BN_ON_LOAD_1X1', '1') != '0', pend=pend1)
                    self.conv_bn(base + '.conv3', b, blk.conv3, blk.bn3, 1, True, out, res=idt, pend=pend2)
                cur = out
        c5 = cur

        # decoder (:392-401)
        self.ops.append(self.flush_bias_slots)      # backward runs the ops in reverse: this one right after the decoder's
        if self.training and 'dec' in self.flush_at:
            self.ops.append(self.flush_wgrads)
        pooled = self.act(H // 64, W // 64, c5.C)
        self.maxpool(c5, pooled)
        specs = [('center', pooled, self.slice(cat5, 0, nf * 8)), ('dec5', Act(cat5), self.slice(cat4, 0, nf * 8)),
                 ('dec4', Act(cat4), self.slice(cat3, 0, nf * 8)), ('dec3', Act(cat3), self.slice(cat2, 0, nf * 2))]
        for dn, xin, dst in specs:
            d = getattr(net, dn)
            mid = self.act(xin.H, xin.W, d.block[0].conv.out_channels)
            self.conv_relu(dn + '.block.0.conv', xin, d.block[0].conv, mid)
            self.deconv_relu(dn + '.block.1', mid, d.block[1], dst)
        x = Act(cat2)
        for dn in ('dec2', 'dec1'):
            d = getattr(net, dn)
            mid = self.act(x.H, x.W, d.block[0].conv.out_channels)
            self.conv_relu(dn + '.block.0.conv', x, d.block[0].conv, mid)
            up = self.act(x.H * 2, x.W * 2, d.block[1].out_channels)
            self.deconv_relu(dn + '.block.1', mid, d.block[1], up)
            x = up
        d0 = self.act(H, W, nf)
        fin = net.final
        dd = self.conv_desc(x, net._pack['w']['dec0.conv'], d0, KH=3, KW=3, stride=1, pad=1, relu=1, shift=net.dec0.conv.bias)
        fused_final = False
        fuse_train = self.training and _os_env.environ.get('MSC_FUSE_FINAL_TRAIN', '0') == '1'      # measured neutral (10.64 / 10.71 vs 10.65 / 10.68 ms): opt-in
        if (not self.training or fuse_train) and self.dev.type == 'cuda' and nf == 32 and _os_env.environ.get('MSC_FUSE_FINAL', '1') != '0':
            # eval: dec0's 3x3 conv + ReLU, the final 1x1 conv and the channel softmax in ONE launch (src/unet_models.py:401-403 +
            # src/models.py:88-92): dec0's 134 MB output is neither written nor read back.  Training (round 4, MSC_FUSE_FINAL_TRAIN=1): the same
            # launch with the output STORED (the backward needs it) and logits only -- the separate final 1x1 pass does not re-read the 134 MB
            dd.final_w, dd.final_b = fin.weight.data_ptr(), fin.bias.data_ptr()
            dd.final_logits, dd.final_probs, dd.final_skip_store = P.logits.data_ptr(), (None if self.training else P.probs.data_ptr()), (0 if self.training else 1)
            if lib.msc_conv_cfg_ok(C.byref(dd), _lib.CFG_HALO):
                dd.cfg, fused_final = _lib.CFG_HALO, True
            else:
                dd.final_w = dd.final_b = dd.final_logits = dd.final_probs = None
                dd.final_skip_store = 0
        self.emit(P.fwd, lib.msc_conv_igemm, C.byref(dd))
        if not fused_final:
            self.emit(P.fwd, lib.msc_final_fwd, d0.ptr, d0.ld, fin.weight.data_ptr(), fin.bias.data_ptr(), P.logits.data_ptr(),
                      None if self.training else P.probs.data_ptr(), self.dt, N, H, W, nf)
        if self.training:
            # the backward sums (BatchNorm-backward, bias gradients) are accumulated atomically into the arena's tail: zeroed at
            # the head of EVERY backward, so a second backward for one forward (retain_graph, re-timing prog.bwd) starts clean
            bwd_slots_from = self.slot_used
            zero_at = len(P.bwd)
            self.emit(P.bwd, lib.msc_memset_zero, 0, 0)      # extent known once the backward is built: patched below
            gd0 = self.grad_of(d0)
            self.grad_acc(d0)
            # final 1x1 backward also applies dec0's ReLU mask, so dec0's backward skips it
            # ... and sums dec0's bias gradient while the masked gradient is in registers
            fin_ws = self.vec(_lib.FINAL_BWD_WS_ROWS * (3 * nf + 2)) if self.ordered else None
            self.emit(P.bwd, lib.msc_final_bwd, P.dlogits.data_ptr(), d0.ptr, d0.ld, fin.weight.data_ptr(), gd0.ptr, gd0.ld,
                      self.g(fin.weight), self.g(fin.bias), self.g(net.dec0.conv.bias), fin_ws.data_ptr() if self.ordered else None,
                      self.dt, N, H, W, nf)
            self._conv_relu_bwd('dec0.conv', x, net.dec0.conv, d0, masked='done')
            for op in reversed(self.ops):
                op()
            self.flush_wgrads()
            P.bwd[zero_at] = (lib.msc_memset_zero, (self.slot_arena.data_ptr() + 8 * bwd_slots_from, 8 * (self.slot_used - bwd_slots_from)))
        P.keep += [xp]
        P.acts = {'c1': s1, 'd0': d0, 'cat2': cat2, 'cat3': cat3, 'cat4': cat4, 'cat5': cat5}
        if self._tuned_new:
            _tune_save()
        return P
