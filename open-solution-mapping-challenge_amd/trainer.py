"""One training step of the U-Net on MI355X: forward, fused loss, backward, gradient all-reduce, Adam.

Replaces the body of the reference's `Model._fit_loop` (src/steps/pytorch/models.py:76-113):
    optimizer.zero_grad(); outputs = model(X); loss = loss_function(outputs, target) * weight;
    loss.backward(); optimizer.step()
with the reference's loss (src/models.py:310-454 or validation.py:25-28) and optimizer
(torch.optim.Adam + L2, src/models.py:57,287-292) executed by HIP kernels over flat buffers.

Data parallel (replaces nn.DataParallel, src/models.py:65): one process per GPU; the four loss sums
are all-reduced between the loss's two phases so Dice / CE are those of the GLOBAL batch (the
reference computes the loss on the gathered batch), gradients are summed over ranks with RCCL in
buckets ordered by backward completion; BatchNorm statistics stay per replica, as in the reference.
"""
import math

import torch

from . import _lib
from ._lib import LossCfg


class LossSpec:
    """Loss configuration in the reference's vocabulary."""

    def __init__(self, weighted=False, dice_weight=0.0, ce_weight=1.0, smooth=0.0, w0=50.0, sigma=10.0,
                 imsize=(256, 256), eps=1e-7, dice_activation='softmax'):
        if dice_activation not in ('softmax', 'sigmoid'):
            raise NotImplementedError('only sigmoid and softmax are implemented')        # src/models.py:443
        self.cfg = LossCfg()
        self.cfg.dice_sigmoid = int(dice_activation == 'sigmoid')
        self.dice_activation = dice_activation
        self.cfg.weighted = int(bool(weighted))
        self.cfg.dice_weight, self.cfg.ce_weight = float(dice_weight), float(ce_weight)
        self.cfg.smooth, self.cfg.eps = float(smooth), float(eps)
        self.cfg.w0, self.cfg.sigma = float(w0), float(sigma)
        self.cfg.size_c = math.sqrt(imsize[0] * imsize[1]) / 2.0        # src/models.py:376
        self.target_channels = 3 if weighted else 1

    @classmethod
    def plain_ce(cls):
        """PyTorchUNet: multiclass_segmentation_loss (src/models.py:107)"""
        return cls(weighted=False, dice_weight=0.0, ce_weight=1.0)

    @classmethod
    def mixed(cls, architecture_config):
        """PyTorchUNetWeighted: mixed_dice_cross_entropy_loss (src/models.py:149-161)"""
        wce = architecture_config['weighted_cross_entropy']
        lw = architecture_config['loss_weights']
        dice = architecture_config['dice']
        return cls(weighted=True, dice_weight=lw['dice_mask'], ce_weight=lw['bce_mask'], smooth=dice.get('smooth', 0),
                   w0=wce['w0'], sigma=wce['sigma'], imsize=tuple(wce['imsize']),
                   dice_activation=dice.get('dice_activation', 'softmax'))           # src/models.py:158,437-442


def _run(fn, args, device):
    """one C-ABI launch on `device`'s current stream, through the same hook as the network's launch lists"""
    from .unet_models import _Program, _stream_of
    _Program.run([(fn, args)], _stream_of(device))


def _zero(t, stream):
    """stream-ordered zero fill of a tensor as a C-ABI launch (msc_memset_zero: a kernel, so a kernel node under capture)"""
    from .unet_models import _Program
    _Program.run([(_lib.load().msc_memset_zero, (t.data_ptr(), t.numel() * t.element_size()))], stream)


def _copy(dst, src, stream):
    """stream-ordered same-device copy of equally shaped contiguous tensors as a C-ABI launch"""
    from .unet_models import _Program
    assert dst.numel() == src.numel() and dst.dtype == src.dtype and dst.is_contiguous() and src.is_contiguous()
    _Program.run([(_lib.load().msc_copy, (dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size()))], stream)


def loss_sums(logits, target, spec, sums):
    """phase 1 of the fused loss: the four f64 sums of this rank's batch"""
    import ctypes as C
    N, Cc, H, W = logits.shape
    if Cc != 2:
        raise ValueError('loss kernels implement the 2-class head')
    _run(_lib.load().msc_loss_sums, (logits.data_ptr(), target.data_ptr(), target.shape[1], C.byref(spec.cfg), sums.data_ptr(), N, H, W),
         logits.device)


def loss_grad(logits, target, spec, dlogits, loss_out, sums, total_pixels, grad_scale=1.0, scale_state=None):
    """phase 2: loss value and dlogits from the (all-reduced) sums; total_pixels = pixels of the GLOBAL batch.  `scale_state`: the
    optimizer's device state -- dlogits are additionally multiplied by its (dynamic) loss scale, read on the device"""
    import ctypes as C
    N, _, H, W = logits.shape
    _run(_lib.load().msc_loss_grad, (logits.data_ptr(), target.data_ptr(), target.shape[1], C.byref(spec.cfg), sums.data_ptr(),
                                     float(total_pixels), float(grad_scale), None if scale_state is None else scale_state.data_ptr(),
                                     loss_out.data_ptr(), dlogits.data_ptr(), N, H, W), logits.device)


def loss_forward_backward(logits, target, spec, dlogits, loss_out, sums, world=None, grad_scale=1.0, scale_state=None):
    """Fused loss: fills `dlogits` (f32 NCHW) and `loss_out` (f32[1]); `sums` f64[4] scratch.
    With `world` (a torch.distributed process group wrapper) the sums are all-reduced first."""
    N, _, H, W = logits.shape
    loss_sums(logits, target, spec, sums)
    total = float(N * H * W)
    if world is not None and world.size > 1:
        world.all_reduce(sums)
        total *= world.size
    loss_grad(logits, target, spec, dlogits, loss_out, sums, total, grad_scale, scale_state)


class _LossFunction(torch.autograd.Function):
    """loss value with the fused kernels; backward hands out the dlogits the second kernel produced anyway"""

    @staticmethod
    def forward(ctx, output, target, spec):
        out = output.detach().contiguous().float()
        tgt = target.detach().contiguous().float()
        dev = out.device
        dl = torch.empty_like(out)
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        sums = torch.zeros(4, dtype=torch.float64, device=dev)
        loss_forward_backward(out, tgt, spec, dl, loss, sums)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, grad):
        (dl,) = ctx.saved_tensors
        return dl * grad.reshape(()), None, None


class HipLoss:
    """A reference-style loss callable `fn(output, target) -> loss` (what the transformers' `loss_function` lists hold and
    the reference's callbacks / score_model call, src/steps/pytorch/validation.py:51-76, callbacks.py:58) on the fused HIP
    loss kernels.  Returns a tensor of ONE element (so both `loss.item()` and the reference's `loss.data.cpu().numpy()[0]`
    work); differentiable with respect to `output`, so the reference's own `_fit_loop` can call `.backward()` on it."""

    def __init__(self, spec, name='loss'):
        self.spec, self.__name__ = spec, name

    def __call__(self, output, target):
        if target.dim() == 3:
            target = target.unsqueeze(1)
        if target.shape[1] < self.spec.target_channels:
            raise ValueError('%s needs a target with %d channels (class, distance, sqrt(size)), got %d'
                             % (self.__name__, self.spec.target_channels, target.shape[1]))
        return _LossFunction.apply(output, target, self.spec)


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, weight_decay) semantics over the model's flat parameter buffer.  It IS a
    torch.optim.Optimizer (one param group holding the model's trainable parameters), so the reference's
    `ExponentialLR(self.optimizer, gamma)` scheduler callback (src/steps/pytorch/callbacks.py:222) drives it unchanged:
    the learning rate is read from `param_groups[0]['lr']` at every step.

    One update = msc_adam_tick + msc_adam_pack: the Adam kernel also writes the 16-bit compute copies of the conv weights (own
    layout + the transpose the data gradient reads), so no separate packing pass re-reads the masters.  Step count, learning
    rate and the loss scale live in device memory (`dev_state`, include/msc.h MSC_OPT_*): a captured hipGraph replays with the
    current values.  fp16 training (`set_loss_scale(..., dynamic=True)`): msc_grad_check raises a flag when a gradient is not
    finite, the step is then skipped on the device (parameters, moments and step count untouched) and the scale halves; it
    doubles again after `growth_interval` clean steps."""

    def __init__(self, net, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__([p for _, p in net._trainable()], dict(lr=float(lr), betas=betas, eps=float(eps), weight_decay=float(weight_decay)))
        self.net, self.lr, self.betas, self.eps, self.weight_decay = net, float(lr), betas, float(eps), float(weight_decay)
        self.m = self.v = self.dev_state = None
        self._steps = 0
        self.loss_scale, self.dynamic_scale, self.growth_interval = 1.0, False, 2000
        self._pending = None          # a state_dict loaded before the flat buffers exist
        self._table = None            # (key, keep-alive tensors, msc_adam_pack table arguments)
        self.on_hyper_change = None   # TrainStep: drop captured graphs (betas / eps / weight_decay are launch arguments)
        self._scale_restored = False  # a loaded state_dict brought its loss scale: TrainStep must not reset it to the default
        self._scale_pinned = False    # TrainStep(loss_scale=...) pinned a static scale: it wins over a checkpoint's, in any construction order

    # the step count lives on the device once training runs: a skipped (overflowed) step does not advance it
    @property
    def steps(self):
        if self.dev_state is not None:
            self._steps = int(round(float(self.dev_state[_lib.OPT_STEP].item())))
        return self._steps

    @steps.setter
    def steps(self, value):
        self._steps = int(value)

    @property
    def skipped_steps(self):
        return 0 if self.dev_state is None else int(round(float(self.dev_state[_lib.OPT_SKIPPED].item())))

    def current_loss_scale(self):
        return self.loss_scale if self.dev_state is None else float(self.dev_state[_lib.OPT_SCALE].item())

    def set_loss_scale(self, scale, dynamic=False, growth_interval=2000):
        self.loss_scale, self.dynamic_scale, self.growth_interval = float(scale), bool(dynamic), int(growth_interval)
        if self.dev_state is not None:
            self.dev_state[_lib.OPT_SCALE] = self.loss_scale
            self.dev_state[_lib.OPT_GROWTH] = float(self.growth_interval if self.dynamic_scale else 0)
            self.dev_state[_lib.OPT_GOOD] = 0.0

    def _ensure(self):
        p = self.net.flat_params
        if p is None:
            raise _lib.MscError('HipAdam: model parameters are not flattened yet (run a forward pass first)')
        if self.m is None or self.m.shape != p.shape or self.m.device != p.device:
            self.m, self.v = torch.zeros_like(p), torch.zeros_like(p)
            st = [0.0] * _lib.OPT_STATE
            st[_lib.OPT_STEP], st[_lib.OPT_LR], st[_lib.OPT_SCALE] = float(self._steps), self.lr, self.loss_scale
            st[_lib.OPT_GROWTH] = float(self.growth_interval if self.dynamic_scale else 0)
            self.dev_state = torch.tensor(st, dtype=torch.float32, device=p.device)
            self._table = None
            if self._pending is not None:
                state, self._pending = self._pending, None
                self._apply_state(state)
        return p

    def set_lr(self, lr):
        self.lr = float(lr)
        self.param_groups[0]['lr'] = self.lr
        if self.dev_state is not None:
            self.dev_state[_lib.OPT_LR] = self.lr

    def sync_lr(self):
        """pick up a learning rate a scheduler wrote into param_groups (also before a hipGraph replay: the captured
        Adam launch reads step count and lr from device memory)"""
        if self.param_groups[0]['lr'] != self.lr:
            self.set_lr(self.param_groups[0]['lr'])

    def zero_grad(self, set_to_none=False):
        if self.net.flat_grads is not None:
            self.net.flat_grads.zero_()

    def _pack_table(self):
        """device tables of msc_adam_pack: one item per trainable tensor (conv weights with their compute copies), blocks of 2048
        elements / 64x32 tiles"""
        import numpy as np
        net = self.net
        dev = net.flat_params.device
        if net._pack is None:
            net._pack = net._build_pack()
            net._programs = {}
            net._packed_version = -1
        key = (id(net._pack), net.flat_params.data_ptr())
        if self._table is not None and self._table[0] == key:
            return self._table[2]
        copies = {}
        for name, mod, kind in net._conv_list():
            if kind == 'stem':
                continue
            w, wt = net._pack['w'][name], net._pack['wt'][name]
            direct, trans = (w, wt) if kind == 'conv' else (wt, w)
            copies[id(mod.weight)] = (None if direct.data_ptr() == mod.weight.data_ptr() else direct, trans)
        base = net.flat_params.data_ptr()
        rec = np.zeros(len(net._trainable()), dtype=np.dtype({'names': ['off', 'n', 'direct', 'trans', 'A', 'T', 'B', 'r'],
                                                              'formats': ['<i8', '<i8', '<u8', '<u8', '<i4', '<i4', '<i4', '<i4'],
                                                              'offsets': [0, 8, 16, 24, 32, 36, 40, 44], 'itemsize': 48}))
        blk_item, blk_local = [], []
        for i, (name, prm) in enumerate(net._trainable()):
            off = (prm.data_ptr() - base) // 4
            n = prm.numel()
            cp = copies.get(id(prm))
            if cp is not None:
                a, b, kh, kw = prm.shape                   # torch-logical [a, b, kh, kw] over the physical [a][kh][kw][b]
                if b % 4:
                    raise _lib.MscError('HipAdam: conv weight %s has %d input channels (multiple of 4 needed)' % (name, b))
                rec[i] = (off, n, cp[0].data_ptr() if cp[0] is not None else 0, cp[1].data_ptr(), a, kh * kw, b, 0)
                nb = kh * kw * ((a + 63) // 64) * ((b + 31) // 32)
            else:
                rec[i] = (off, n, 0, 0, 0, 0, 0, 0)
                nb = (n + 2047) // 2048
            blk_item.append(np.full(nb, i, np.int32))
            blk_local.append(np.arange(nb, dtype=np.int32))
        t_items = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        t_bi = torch.from_numpy(np.concatenate(blk_item)).to(dev)
        t_bl = torch.from_numpy(np.concatenate(blk_local)).to(dev)
        args = (t_items.data_ptr(), t_bi.data_ptr(), t_bl.data_ptr(), int(t_bi.numel()), net._dt)
        self._table = (key, (t_items, t_bi, t_bl), args)
        return args

    def launches(self, grad_scale=1.0):
        """the launches of one update as (fn, args) pairs: (overflow check in the dynamic-scale mode,) step counter tick, fused
        Adam + weight packing, the stem's own compute layout"""
        p = self._ensure()
        lib = _lib.load()
        g = self.net.flat_grads
        items, bi, bl, nb, dt = self._pack_table()
        out = []
        if self.dynamic_scale:
            out.append((lib.msc_grad_check, (g.data_ptr(), g.numel(), self.dev_state.data_ptr())))
        out.append((lib.msc_adam_tick, (self.dev_state.data_ptr(),)))
        out.append((lib.msc_adam_pack, (p.data_ptr(), g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), items, bi, bl, nb, dt, self.lr,
                                        self.betas[0], self.betas[1], self.eps, self.weight_decay, 0, float(grad_scale), self.dev_state.data_ptr())))
        out += [op for op in self.net._pack['ops'] if op[0].__name__ == 'msc_stem_pack']
        return out

    def step(self, closure=None, grad_scale=1.0):
        p = self._ensure()
        self.sync_lr()
        from .unet_models import _Program, _stream_of
        _Program.run(self.launches(grad_scale), _stream_of(p.device))
        self.net.weights_changed()
        self.net._packed_version = self.net._version        # the update wrote the compute copies itself

    def state_dict(self):
        return {'m': self.m, 'v': self.v, 'steps': self.steps, 'lr': self.lr, 'loss_scale': self.current_loss_scale(),
                'dynamic_scale': self.dynamic_scale, 'growth_interval': self.growth_interval,
                'param_groups': [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups]}

    def load_state_dict(self, state):
        """restore what state_dict() returned: both moment buffers (flat, in the model's parameter order), the step count the
        bias correction uses and the learning rate -- also into the device-resident state a captured graph reads.  Works
        before the first forward pass too (the usual resume order: build model and optimizer, load both, train): the state is
        kept and applied when the flat buffers appear."""
        if self.net.flat_params is None:
            self._pending = state
            self._apply_hyper(state)
            return
        self._ensure()
        self._apply_state(state)

    def _apply_hyper(self, state):
        old = (tuple(self.betas), self.eps, self.weight_decay)
        for g, saved in zip(self.param_groups, state.get('param_groups', [])):
            g.update({k: v for k, v in saved.items() if k != 'params'})
        g0 = self.param_groups[0]
        self.betas, self.eps, self.weight_decay = tuple(g0['betas']), float(g0['eps']), float(g0['weight_decay'])
        self._steps = int(state['steps'])
        self.lr = float(state['lr'])
        self.param_groups[0]['lr'] = self.lr
        if (tuple(self.betas), self.eps, self.weight_decay) != old and self.on_hyper_change is not None:
            self.on_hyper_change()       # betas / eps / weight_decay are arguments of already captured launches

    def _apply_state(self, state):
        p = self.net.flat_params
        for key in ('m', 'v'):
            src = state[key]
            if src is None:
                getattr(self, key).zero_()
            else:
                if src.numel() != p.numel():
                    raise ValueError('HipAdam.load_state_dict: %s has %d elements, the model has %d' % (key, src.numel(), p.numel()))
                getattr(self, key).copy_(src.to(p.device, torch.float32).reshape(p.shape))
        self._apply_hyper(state)
        self.set_lr(self.lr)
        self.dev_state[_lib.OPT_STEP] = float(self._steps)
        if state.get('loss_scale') and not self._scale_pinned:
            # the (dynamic) loss scale a checkpoint carries is restored whatever order optimizer / TrainStep were built in -- unless the TrainStep
            # pinned an explicit static scale, which always wins; the clean-step counter and a pending overflow flag belong to the run that wrote
            # the checkpoint
            if 'dynamic_scale' in state:
                was = (self.dynamic_scale, self.growth_interval)
                self.dynamic_scale, self.growth_interval = bool(state['dynamic_scale']), int(state.get('growth_interval', self.growth_interval))
                if (self.dynamic_scale, self.growth_interval) != was and self.on_hyper_change is not None:
                    self.on_hyper_change()       # launches() holds msc_grad_check only in the dynamic mode: a captured step is stale
            self.loss_scale = float(state['loss_scale'])
            self.dev_state[_lib.OPT_SCALE] = self.loss_scale
            self.dev_state[_lib.OPT_GROWTH] = float(self.growth_interval if self.dynamic_scale else 0)
            self.dev_state[_lib.OPT_GOOD] = 0.0
            self.dev_state[_lib.OPT_OVERFLOW] = 0.0
            self._scale_restored = True


def DDP_PIECES():
    """pieces the backward is cut into around the gradient exchanges (MSC_DDP_PIECES, default 4; measurement switch)"""
    import os
    return max(1, int(os.environ.get('MSC_DDP_PIECES', '4')))


def DDP_ONE_GRAPH():
    """the distributed step captured as ONE graph with the RCCL calls inside (MSC_DDP_ONE_GRAPH=1; default: piecewise graphs around eager calls)"""
    import os
    return os.environ.get('MSC_DDP_ONE_GRAPH', '0') == '1'


DDP_FRACTIONS = (0.40, 0.65, 0.85)      # share of the gradient bytes that must be final before the first three exchanges


def ddp_plan(prog, flat_grads, nchunks=4, fractions=None):
    """Cut the backward launch list into `nchunks` pieces and find, after each piece, the suffix of the flat
    gradient buffer that is final (parameters are laid out in forward order, backward finishes them from the
    tail): [(launch_end, grad_lo, grad_hi) ...] with grad_lo None when nothing new completed.  The gradient
    all-reduce of a piece is issued as soon as its launches are enqueued, so RCCL overlaps the rest of backward.

    The cuts are placed by BYTES, not by launch count: piece c ends at the first launch after which `fractions[c]` of the
    gradient buffer is final (default 40 / 65 / 85 %), the last piece takes the rest -- the LAST exchange has no compute
    left to hide behind, so it gets the smallest share (<= 25 %: the stem, layer1 / layer2 and the head of layer3, whose
    weight gradients complete with the last grouped launch), and the decoder's 45 % of the bytes, final after the first
    third of backward, leaves early.  Falls back to equal launch counts when the program finishes (almost) everything
    with its last launches (all weight gradients in one grouped launch: the single-process configuration)."""
    base, n = flat_grads.data_ptr(), flat_grads.numel()
    last = {}
    for idx, ptr in prog.grad_writes:
        off = (ptr - base) // 4
        last[off] = max(last.get(off, -1), idx)
    offs = sorted(last)
    total = len(prog.bwd)

    def final_from(end):
        """start of the suffix of the buffer whose writers all lie in launches [0, end)"""
        s = n
        for off in reversed(offs):
            if last[off] < end:
                s = off
            else:
                break
        return s

    fractions = DDP_FRACTIONS if fractions is None else tuple(fractions)
    ends = []
    if nchunks == len(fractions) + 1 and offs:
        # the suffix start only moves at launches that are some parameter's last writer
        marks = sorted(set(last.values()))
        for f in fractions:
            end = next((m + 1 for m in marks if (n - final_from(m + 1)) >= f * n), total)
            ends.append(max(end, (ends[-1] + 1) if ends else 1))
        if ends[-1] >= total or (n - final_from(ends[-1])) < 0.75 * n:      # nothing much is final before the end: by launch count
            ends = []
    if not ends:
        ends = [total * (c + 1) // nchunks for c in range(nchunks - 1)]
    ends.append(total)
    plan, done_from = [], n
    for c, end in enumerate(ends):
        s = 0 if c == len(ends) - 1 else final_from(end)
        if s < done_from:
            plan.append((end, s, done_from))
            done_from = s
        else:
            plan.append((end, None, None))
    return plan


class _ShapeState:
    """everything of a TrainStep that is bound to one batch shape: staging buffers, the program, captured graphs"""
    __slots__ = ('x', 't', 'prog', 'graph', 'pieces', 'one')

    def __init__(self, x, target):
        self.x = torch.empty_like(x, dtype=torch.float32)
        self.t = torch.empty_like(target, dtype=torch.float32)
        self.prog = self.graph = self.pieces = self.one = None


class TrainStep:
    """forward -> loss -> backward -> (all-reduce) -> Adam.  use_graph: the ~1.1k launches of a step are captured so that
    they cost no host time on replay -- ONE hipGraph in a single process; with collectives, a handful of graphs (forward +
    loss sums | loss gradient + backward piece 1 | piece 2.. | Adam) with the RCCL calls issued eagerly between them, so
    the gradient all-reduce of a piece still overlaps the next pieces.  Staging buffers, program and graphs are kept per
    batch shape: the reference's DataLoader has no drop_last, so the last batch of an epoch is usually smaller, and a
    graph (or graph piece) captured for one shape must never be replayed for another."""

    def __init__(self, net, spec, optimizer, world=None, use_graph=False, force_collectives=False, loss_scale=None):
        self.net, self.spec, self.opt, self.world = net, spec, optimizer, world
        # fp16 activations cannot hold the gradients of a mean over millions of pixels (dlogits ~ 1e-7 per pixel, below
        # the smallest fp16 subnormal): a static loss scale multiplies dlogits and is divided out again inside the
        # Adam kernel, on the fp32 weight gradients.  bf16 / fp32 share fp32's exponent range and need none.
        # fp16: the scale is DYNAMIC (device-resident, HipAdam.set_loss_scale): a step whose gradients overflowed is skipped and halves
        # it; an explicit loss_scale argument pins a static one.
        fp16 = getattr(net, 'compute_dtype', '') == 'fp16'
        self.loss_scale = float(loss_scale) if loss_scale is not None else (4096.0 if fp16 else 1.0)
        if loss_scale is None and fp16 and optimizer._scale_restored and optimizer.dynamic_scale:
            self.loss_scale = optimizer.loss_scale          # resumed fp16 run: keep the scale the checkpoint reached
        else:
            optimizer.set_loss_scale(self.loss_scale, dynamic=(fp16 and loss_scale is None))
        optimizer._scale_pinned = loss_scale is not None
        optimizer.on_hyper_change = self._drop_graphs
        self.dist = world is not None and (world.size > 1 or force_collectives)     # force: exercise RCCL with one rank
        self.use_graph = use_graph
        self.keep_graph = False       # tests: keep the captured hipGraph_t (torch.cuda.CUDAGraph(keep_graph=True)) so that its nodes can be inspected
        self.shapes = {}
        self.cur = None
        self.loss = self.sums = None

    # the state of the batch shape in use (what the tests and bench.py look at)
    x = property(lambda self: self.cur.x if self.cur else None)
    t = property(lambda self: self.cur.t if self.cur else None)
    prog = property(lambda self: self.cur.prog if self.cur else None)
    graph = property(lambda self: self.cur.graph if self.cur else None)
    pieces = property(lambda self: self.cur.pieces if self.cur else None)

    def _drop_graphs(self):
        """captured launches carry betas / eps / weight_decay as arguments: re-capture after they changed"""
        for st in self.shapes.values():
            st.graph = st.pieces = st.one = None

    def _setup(self, x, target):
        key = (tuple(x.shape), tuple(target.shape), x.device)
        st = self.shapes.get(key)
        if st is None:
            st = self.shapes[key] = _ShapeState(x, target)
        self.cur = st
        if self.loss is None or self.loss.device != x.device:
            self.loss = torch.zeros(1, dtype=torch.float32, device=x.device)
            self.sums = torch.zeros(4, dtype=torch.float64, device=x.device)
        return st

    def _body(self):
        net, st = self.net, self.cur
        prog = net.train_forward(st.x)
        st.prog = prog
        self.opt._ensure()
        loss_forward_backward(prog.logits, st.t, self.spec, prog.dlogits, self.loss, self.sums, self.world, 1.0, self.opt.dev_state)
        if self.dist:
            self._backward_overlapped(prog)
        else:
            net.train_backward(prog)
        self.opt.step()

    def _backward_overlapped(self, prog):
        """backward in pieces; each piece's finished gradient range goes to RCCL (async, its own stream) while the
        next piece computes"""
        from .unet_models import _Program
        net = self.net
        flat_g = net.flat_grads
        if getattr(prog, '_ddp_plan', None) is None:
            prog._ddp_plan = ddp_plan(prog, flat_g, nchunks=DDP_PIECES())
        flat_g.zero_()
        prog.stem_dw.zero_()
        works, beg = [], 0
        for end, lo, hi in prog._ddp_plan:
            _Program.run_backward(prog.bwd[beg:end], flat_g.device)     # joins its side stream before returning
            beg = end
            if lo is not None:
                works.append(self.world.all_reduce_grad_range(flat_g, lo, hi))
        for w in works:
            w.wait()

    def __call__(self, x, target):
        st = self._setup(x, target)
        st.x.copy_(x, non_blocking=True)
        st.t.copy_(target, non_blocking=True)
        if not self.use_graph:
            self._body()
            return self.loss
        if st.graph is None and st.pieces is None and st.one is None:
            # the first step of a shape runs eagerly (builds the program, allocates, packs) and IS this call's step;
            # capturing afterwards does not execute anything, replays start with the next call of this shape
            if st.x.is_cuda:
                from .unet_models import _quiesce
                _quiesce(st.x.device)          # nothing of an earlier user of the device in flight while the step that gets captured is built
            self._body()
            if st.x.is_cuda:
                torch.cuda.synchronize()
            one_graph = self.dist and DDP_ONE_GRAPH()
            if one_graph:
                # only RCCL's collectives are stream work that a capture can record: gloo's device path synchronises on the host and a
                # capture around it never returns (seen with two gloo ranks on one GPU, tests/test_gpu_two_ranks.py) -- refuse, do not try
                import torch.distributed as dist
                if not (dist.is_initialized() and dist.get_backend(self.world.group) == 'nccl'):
                    import warnings
                    warnings.warn('MSC_DDP_ONE_GRAPH=1 needs the nccl (RCCL) backend; using the piecewise graphs')
                    one_graph = False
            if one_graph:
                # round 6: the whole distributed step as ONE graph, the RCCL calls captured as forked branches; a capture that fails falls back
                # to the piecewise graphs below (and those to eager launches)
                try:
                    self._capture_one()
                except RuntimeError as e:
                    import warnings
                    warnings.warn('one-graph capture of the distributed step failed (%s); falling back to piecewise graphs' % e)
                    if st.x.is_cuda:
                        torch.cuda.synchronize()
                    st.one = None
            if self.dist and st.one is None:
                try:
                    self._capture_pieces()
                except RuntimeError as e:          # capture next to a live communicator is the fragile part: keep training
                    import warnings
                    warnings.warn('hipGraph capture of the distributed step failed (%s); continuing with eager launches' % e)
                    if st.x.is_cuda:
                        torch.cuda.synchronize()
                    st.pieces, self.use_graph = None, False
            else:
                g = torch.cuda.CUDAGraph(keep_graph=True) if self.keep_graph else torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._body_captured()
                if self.keep_graph:
                    g.instantiate()
                st.graph = g
            return self.loss
        self.opt.sync_lr()                # a scheduler callback may have changed it: the captured Adam reads it from the device
        if self.net._packed_version != self.net._version:
            # the master weights changed between two replays (load_state_dict, load_encoder_state_dict, weights_changed()): the replay
            # reads the 16-bit compute copies, which only the Adam launch at the END of a step rewrites -- repack first
            from .unet_models import _stream_of
            self.net._refresh_weights(_stream_of(st.x.device))
        if st.one is not None:
            st.one.replay()
        elif st.pieces is not None:
            self._replay_pieces()
        else:
            st.graph.replay()
        self.opt._opt_called = True       # torch's LR schedulers check that optimizer.step() ran before scheduler.step(): the replay was that step
        self.net._version += 1            # the replay changed the master weights -- and wrote their compute copies (msc_adam_pack)
        self.net._packed_version = self.net._version
        return self.loss

    # ---- piecewise capture: graphs around the collectives ----------------------------------------------
    def _capture_pieces(self):
        from .unet_models import _Program
        net, st = self.net, self.cur
        prog, dev = st.prog, st.x.device
        N, _, H, W = prog.logits.shape
        total = float(N * H * W) * (self.world.size if self.world.size > 1 else 1)
        flat_g = net.flat_grads
        if getattr(prog, '_ddp_plan', None) is None:
            prog._ddp_plan = ddp_plan(prog, flat_g, nchunks=DDP_PIECES())

        from .unet_models import _stream_of

        def capture(fn):
            # thread_local: the RCCL watchdog thread of torch.distributed may poll events while we capture.  Everything a
            # piece does is a C-ABI launch (kernel nodes only, the fills and copies included): nothing runs at capture time
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                fn(_stream_of(dev))
            return g

        def forward(stream):
            _copy(prog.x_in, st.x, stream)
            _Program.run(prog.fwd, stream)
            loss_sums(prog.logits, st.t, self.spec, self.sums)

        def piece(beg, end, first):
            def fn(stream):
                if first:
                    loss_grad(prog.logits, st.t, self.spec, prog.dlogits, self.loss, self.sums, total, 1.0, self.opt.dev_state)
                    _zero(flat_g, stream)
                    _zero(prog.stem_dw, stream)
                _Program.run_backward(prog.bwd[beg:end], dev)
            return fn

        def adam(stream):
            _Program.run(self.opt.launches(), stream)

        pieces, beg = [], 0
        for end, lo, hi in prog._ddp_plan:
            pieces.append((capture(piece(beg, end, beg == 0)), lo, hi))
            beg = end
        st.pieces = (capture(forward), pieces, capture(adam))

    def _capture_one(self):
        """The distributed step as ONE hipGraph (MSC_DDP_ONE_GRAPH=1): forward, loss sums, their all-reduce, loss gradient, the backward pieces with
        the gradient exchange of each piece forked off as it completes (torch.distributed issues a collective on its own stream behind an event of
        the caller's stream; under capture that is a parallel branch of the graph), the joins, Adam.  No host round trip and no idle gap at the
        hand-overs between graphs (profiles/r5_run3_collectives_gaps.txt: +2.6-3.2 % for the six-graph form on one GPU).  Validated with a process
        group of ONE rank only (gpurun exposes one GPU): opt-in until it has run on a multi-GPU node."""
        from .unet_models import _Program, _stream_of
        net, st = self.net, self.cur
        prog, dev = st.prog, st.x.device
        N, _, H, W = prog.logits.shape
        total = float(N * H * W) * (self.world.size if self.world.size > 1 else 1)
        flat_g = net.flat_grads
        if getattr(prog, '_ddp_plan', None) is None:
            prog._ddp_plan = ddp_plan(prog, flat_g, nchunks=DDP_PIECES())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            stream = _stream_of(dev)
            _copy(prog.x_in, st.x, stream)
            _Program.run(prog.fwd, stream)
            loss_sums(prog.logits, st.t, self.spec, self.sums)
            self.world.all_reduce(self.sums)
            loss_grad(prog.logits, st.t, self.spec, prog.dlogits, self.loss, self.sums, total, 1.0, self.opt.dev_state)
            _zero(flat_g, stream)
            _zero(prog.stem_dw, stream)
            works, beg = [], 0
            for end, lo, hi in prog._ddp_plan:
                _Program.run_backward(prog.bwd[beg:end], dev)
                beg = end
                if lo is not None:
                    works.append(self.world.all_reduce_grad_range(flat_g, lo, hi))
            for w in works:
                w.wait()
            _Program.run(self.opt.launches(), stream)
        st.one = g

    def _replay_pieces(self):
        fwd, pieces, adam = self.cur.pieces
        flat_g = self.net.flat_grads
        fwd.replay()
        self.world.all_reduce(self.sums)
        works = []
        for g, lo, hi in pieces:
            g.replay()
            if lo is not None:
                works.append(self.world.all_reduce_grad_range(flat_g, lo, hi))
        for w in works:
            w.wait()
        adam.replay()

    def _body_captured(self):
        # same as _body; the compute copies of the weights are written by the Adam launch at the end of every replay
        net, st = self.net, self.cur
        from .unet_models import _Program, _stream_of
        stream = _stream_of(st.x.device)
        prog = st.prog
        _copy(prog.x_in, st.x, stream)
        _Program.run(prog.fwd, stream)
        loss_forward_backward(prog.logits, st.t, self.spec, prog.dlogits, self.loss, self.sums, None, 1.0, self.opt.dev_state)
        _zero(net._flat[1], stream)
        _zero(prog.stem_dw, stream)
        _Program.run_backward(prog.bwd, st.x.device)
        _Program.run(self.opt.launches(), stream)
