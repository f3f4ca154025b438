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
                 imsize=(256, 256), eps=1e-7):
        self.cfg = LossCfg()
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
                   w0=wce['w0'], sigma=wce['sigma'], imsize=tuple(wce['imsize']))


def loss_sums(logits, target, spec, sums):
    """phase 1 of the fused loss: the four f64 sums of this rank's batch"""
    import ctypes as C
    N, Cc, H, W = logits.shape
    if Cc != 2:
        raise ValueError('loss kernels implement the 2-class head')
    stream = torch.cuda.current_stream(logits.device).cuda_stream
    _lib.check(_lib.load().msc_loss_sums(logits.data_ptr(), target.data_ptr(), target.shape[1], C.byref(spec.cfg), sums.data_ptr(),
                                         N, H, W, stream), 'msc_loss_sums')


def loss_grad(logits, target, spec, dlogits, loss_out, sums, total_pixels, grad_scale=1.0):
    """phase 2: loss value and dlogits from the (all-reduced) sums; total_pixels = pixels of the GLOBAL batch"""
    import ctypes as C
    N, _, H, W = logits.shape
    stream = torch.cuda.current_stream(logits.device).cuda_stream
    _lib.check(_lib.load().msc_loss_grad(logits.data_ptr(), target.data_ptr(), target.shape[1], C.byref(spec.cfg), sums.data_ptr(),
                                         float(total_pixels), float(grad_scale), loss_out.data_ptr(), dlogits.data_ptr(), N, H, W, stream),
               'msc_loss_grad')


def loss_forward_backward(logits, target, spec, dlogits, loss_out, sums, world=None, grad_scale=1.0):
    """Fused loss: fills `dlogits` (f32 NCHW) and `loss_out` (f32[1]); `sums` f64[4] scratch.
    With `world` (a torch.distributed process group wrapper) the sums are all-reduced first."""
    N, _, H, W = logits.shape
    loss_sums(logits, target, spec, sums)
    total = float(N * H * W)
    if world is not None and world.size > 1:
        world.all_reduce(sums)
        total *= world.size
    loss_grad(logits, target, spec, dlogits, loss_out, sums, total, grad_scale)


class HipAdam:
    """torch.optim.Adam(params, lr, weight_decay) semantics over the model's flat parameter buffer."""

    def __init__(self, net, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.net, self.lr, self.betas, self.eps, self.weight_decay = net, float(lr), betas, float(eps), float(weight_decay)
        self.m = self.v = self.state = None
        self.steps = 0
        self.param_groups = [{'lr': self.lr}]     # what the reference's LR scheduler callbacks poke at

    def _ensure(self):
        p = self.net.flat_params
        if p is None:
            raise _lib.MscError('HipAdam: model parameters are not flattened yet (run a forward pass first)')
        if self.m is None or self.m.shape != p.shape or self.m.device != p.device:
            self.m, self.v = torch.zeros_like(p), torch.zeros_like(p)
            self.state = torch.tensor([float(self.steps), self.lr], dtype=torch.float32, device=p.device)
        return p

    def set_lr(self, lr):
        self.lr = float(lr)
        self.param_groups[0]['lr'] = self.lr
        if self.state is not None:
            self.state[1] = self.lr

    def zero_grad(self):
        if self.net.flat_grads is not None:
            self.net.flat_grads.zero_()

    def step(self, grad_scale=1.0):
        p = self._ensure()
        if self.param_groups[0]['lr'] != self.lr:
            self.set_lr(self.param_groups[0]['lr'])
        g = self.net.flat_grads
        stream = torch.cuda.current_stream(p.device).cuda_stream
        lib = _lib.load()
        self.steps += 1
        _lib.check(lib.msc_adam_tick(self.state.data_ptr(), stream), 'msc_adam_tick')
        _lib.check(lib.msc_adam_step(p.data_ptr(), g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), p.numel(), self.lr,
                                     self.betas[0], self.betas[1], self.eps, self.weight_decay, 0, float(grad_scale),
                                     self.state.data_ptr(), stream), 'msc_adam_step')
        self.net.weights_changed()

    def state_dict(self):
        return {'m': self.m, 'v': self.v, 'steps': self.steps, 'lr': self.lr}


def ddp_plan(prog, flat_grads, nchunks=4):
    """Cut the backward launch list into `nchunks` pieces and find, after each piece, the suffix of the flat
    gradient buffer that is final (parameters are laid out in forward order, backward finishes them from the
    tail): [(launch_end, grad_lo, grad_hi) ...] with grad_lo None when nothing new completed.  The gradient
    all-reduce of a piece is issued as soon as its launches are enqueued, so RCCL overlaps the rest of backward."""
    base, n = flat_grads.data_ptr(), flat_grads.numel()
    last = {}
    for idx, ptr in prog.grad_writes:
        off = (ptr - base) // 4
        last[off] = max(last.get(off, -1), idx)
    offs = sorted(last)
    total = len(prog.bwd)
    plan, done_from = [], n
    for c in range(nchunks):
        end = total * (c + 1) // nchunks
        s = n
        for off in reversed(offs):
            if last[off] < end:
                s = off
            else:
                break
        if c == nchunks - 1:
            s = 0
        if s < done_from:
            plan.append((end, s, done_from))
            done_from = s
        else:
            plan.append((end, None, None))
    return plan


class TrainStep:
    """forward -> loss -> backward -> (all-reduce) -> Adam for a fixed batch shape.  use_graph: the ~1.1k launches of
    a step are captured so that they cost no host time on replay -- ONE hipGraph in a single process; with
    collectives, a handful of graphs (forward + loss sums | loss gradient + backward piece 1 | piece 2.. | Adam) with the
    RCCL calls issued eagerly between them, so the gradient all-reduce of a piece still overlaps the next pieces."""

    def __init__(self, net, spec, optimizer, world=None, use_graph=False, force_collectives=False):
        self.net, self.spec, self.opt, self.world = net, spec, optimizer, world
        self.dist = world is not None and (world.size > 1 or force_collectives)     # force: exercise RCCL with one rank
        self.use_graph = use_graph
        self.graph = None
        self.pieces = None
        self.x = self.t = self.loss = self.sums = None
        self.prog = None

    def _setup(self, x, target):
        dev = x.device
        self.x = torch.empty_like(x, dtype=torch.float32)
        self.t = torch.empty_like(target, dtype=torch.float32)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sums = torch.zeros(4, dtype=torch.float64, device=dev)

    def _body(self):
        net = self.net
        prog = net.train_forward(self.x)
        self.prog = prog
        loss_forward_backward(prog.logits, self.t, self.spec, prog.dlogits, self.loss, self.sums, self.world)
        if self.dist:
            self._backward_overlapped(prog)
        else:
            net.train_backward(prog)
        self.opt.step()

    def _backward_overlapped(self, prog):
        """backward in pieces; each piece's finished gradient range goes to RCCL (async, its own stream) while the
        next piece computes"""
        import torch.distributed as dist
        from .unet_models import _Program
        net = self.net
        flat_g = net.flat_grads
        if getattr(prog, '_ddp_plan', None) is None:
            prog._ddp_plan = ddp_plan(prog, flat_g)
        flat_g.zero_()
        prog.stem_dw.zero_()
        works, beg = [], 0
        for end, lo, hi in prog._ddp_plan:
            _Program.run_backward(prog.bwd[beg:end], flat_g.device)     # joins its side stream before returning
            beg = end
            if lo is not None:
                works.append(dist.all_reduce(flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.world.group, async_op=True))
        for w in works:
            w.wait()

    def __call__(self, x, target):
        if self.x is None or self.x.shape != x.shape or self.t.shape != target.shape:
            self._setup(x, target)
            self.graph = None
        self.x.copy_(x, non_blocking=True)
        self.t.copy_(target, non_blocking=True)
        if not self.use_graph:
            self._body()
            return self.loss
        if self.graph is None and self.pieces is None:
            # the first step runs eagerly (builds the program, allocates, packs) and IS this call's step;
            # capturing afterwards does not execute anything, replays start with the next call
            self._body()
            torch.cuda.synchronize()
            if self.dist:
                try:
                    self._capture_pieces()
                except RuntimeError as e:          # capture next to a live communicator is the fragile part: keep training
                    import warnings
                    warnings.warn('hipGraph capture of the distributed step failed (%s); continuing with eager launches' % e)
                    torch.cuda.synchronize()
                    self.pieces, self.use_graph = None, False
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._body_captured()
                self.graph = g
            return self.loss
        if self.pieces is not None:
            self._replay_pieces()
        else:
            self.graph.replay()
        self.opt.steps += 1
        return self.loss

    # ---- piecewise capture: graphs around the collectives ----------------------------------------------
    def _capture_pieces(self):
        from .unet_models import _Program
        net, prog, dev = self.net, self.prog, self.x.device
        N, _, H, W = prog.logits.shape
        total = float(N * H * W) * (self.world.size if self.world.size > 1 else 1)
        flat_g = net.flat_grads
        if getattr(prog, '_ddp_plan', None) is None:
            prog._ddp_plan = ddp_plan(prog, flat_g)

        def capture(fn):
            # thread_local: the RCCL watchdog thread of torch.distributed may poll events while we capture
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                fn(torch.cuda.current_stream(dev).cuda_stream)
            return g

        def forward(stream):
            _Program.run(net._pack['ops'], stream)
            prog.x_in.copy_(self.x)
            _Program.run(prog.fwd, stream)
            loss_sums(prog.logits, self.t, self.spec, self.sums)

        def piece(beg, end, first):
            def fn(stream):
                if first:
                    loss_grad(prog.logits, self.t, self.spec, prog.dlogits, self.loss, self.sums, total)
                    flat_g.zero_()
                    prog.stem_dw.zero_()
                _Program.run_backward(prog.bwd[beg:end], dev)
            return fn

        def adam(stream):
            o, lib = self.opt, _lib.load()
            p = net.flat_params
            _lib.check(lib.msc_adam_tick(o.state.data_ptr(), stream), 'msc_adam_tick')
            _lib.check(lib.msc_adam_step(p.data_ptr(), flat_g.data_ptr(), o.m.data_ptr(), o.v.data_ptr(), p.numel(), o.lr, o.betas[0],
                                         o.betas[1], o.eps, o.weight_decay, 0, 1.0, o.state.data_ptr(), stream), 'msc_adam_step')

        pieces, beg = [], 0
        for end, lo, hi in prog._ddp_plan:
            pieces.append((capture(piece(beg, end, beg == 0)), lo, hi))
            beg = end
        self.pieces = (capture(forward), pieces, capture(adam))
        net._packed_version = -1

    def _replay_pieces(self):
        import torch.distributed as dist
        fwd, pieces, adam = self.pieces
        flat_g = self.net.flat_grads
        fwd.replay()
        self.world.all_reduce(self.sums)
        works = []
        for g, lo, hi in pieces:
            g.replay()
            if lo is not None:
                works.append(dist.all_reduce(flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.world.group, async_op=True))
        for w in works:
            w.wait()
        adam.replay()

    def _body_captured(self):
        # same as _body, but the weight repack after Adam is part of the graph so replays stay consistent
        net = self.net
        stream = torch.cuda.current_stream(self.x.device).cuda_stream
        from .unet_models import _Program
        _Program.run(net._pack['ops'], stream)
        prog = self.prog
        prog.x_in.copy_(self.x)
        _Program.run(prog.fwd, stream)
        loss_forward_backward(prog.logits, self.t, self.spec, prog.dlogits, self.loss, self.sums, None)
        net._flat[1].zero_()
        prog.stem_dw.zero_()
        _Program.run_backward(prog.bwd, self.x.device)
        lib = _lib.load()
        o = self.opt
        p, gr = net.flat_params, net.flat_grads
        _lib.check(lib.msc_adam_tick(o.state.data_ptr(), stream), 'msc_adam_tick')
        _lib.check(lib.msc_adam_step(p.data_ptr(), gr.data_ptr(), o.m.data_ptr(), o.v.data_ptr(), p.numel(), o.lr, o.betas[0],
                                     o.betas[1], o.eps, o.weight_decay, 0, 1.0, o.state.data_ptr(), stream), 'msc_adam_step')
        net._packed_version = -1      # host bookkeeping: packed copies refreshed at the head of every replay
