"""GPU: single-kernel parity through the C ABI against torch-CPU fp32 references of the same op.
fp32 mode (v_mfma_f32_16x16x4_f32) is held to 1e-4-class tolerances; bf16 mode to bf16 rounding."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16, torch.float16]


def tol(dtype):
    # one rounding of the stored result to the dtype (bf16: 2^-8, fp16: 2^-11 relative) on top of fp32 accumulation
    return {torch.float32: dict(rtol=2e-5, atol=2e-5), torch.bfloat16: dict(rtol=3e-2, atol=3e-2), torch.float16: dict(rtol=4e-3, atol=4e-3)}[dtype]


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).float()      # values representable in dtype


def nhwc(t, dtype):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def to_nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cin,cout,k,stride,hw,n', [(64, 64, 3, 1, 12, 2), (64, 128, 1, 1, 9, 3), (128, 64, 3, 2, 16, 2),
                                                    (64, 256, 1, 2, 14, 2), (32, 32, 3, 1, 20, 1), (320, 128, 3, 1, 8, 2),
                                                    (64, 64, 3, 1, 40, 4), (256, 512, 3, 1, 33, 2)])
def test_conv_forward_epilogues(dtype, cin, cout, k, stride, hw, n):
    import hip_ops as ops
    pad = k // 2
    x = rnd((n, cin, hw, hw), dtype, 1)
    w = rnd((cout, cin, k, k), dtype, 2, (2.0 / (cin * k * k)) ** 0.5)
    ho = (hw + 2 * pad - k) // stride + 1
    res = rnd((n, cout, ho, ho), dtype, 3)
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1
    ref = F.conv2d(x, w, stride=stride, padding=pad)
    wk = w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    out = torch.empty((n, ho, ho, cout), dtype=dtype, device='cuda')
    ops.conv_igemm(nhwc(x, dtype), wk, out, stride=stride, pad=pad)
    assert torch.allclose(to_nchw(out), ref, **tol(dtype))
    ops.conv_igemm(nhwc(x, dtype), wk, out, stride=stride, pad=pad, relu=True, scale=scale.cuda(), shift=shift.cuda(),
                   res=nhwc(res, dtype))
    ref2 = torch.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    assert torch.allclose(to_nchw(out), ref2, **tol(dtype))


@pytest.mark.parametrize('dtype', DT)
def test_conv_channel_slices_and_stats(dtype):
    """input / output / residual as channel slices of wider buffers; BN partial statistics"""
    import hip_ops as ops
    n, hw, cin, cout = 2, 10, 64, 64
    x = rnd((n, cin, hw, hw), dtype, 1)
    w = rnd((cout, cin, 3, 3), dtype, 2, 0.06)
    xbuf = torch.zeros((n, hw, hw, 192), dtype=dtype, device='cuda')
    xbuf[..., 64:128] = nhwc(x, dtype)
    obuf = torch.full((n, hw, hw, 160), 7.0, dtype=dtype, device='cuda')
    xin, out = xbuf[..., 64:128], obuf[..., 32:96]
    slices = ops.conv_stats_slices(xin, w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda(), out, 1, 1)
    stats = torch.zeros((slices, cout, 2), dtype=torch.float64, device='cuda')      # layout [XCD slot][Cout][2], accumulated
    ops.conv_igemm(xin, w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda(), out, stride=1, pad=1, stats=stats)
    ref = F.conv2d(x, w, padding=1)
    assert torch.allclose(to_nchw(out), ref, **tol(dtype))
    assert (obuf[..., :32] == 7).all() and (obuf[..., 96:] == 7).all()          # neighbours untouched
    s = stats.sum(0).float().cpu()
    assert torch.allclose(s[:, 0], ref.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    assert torch.allclose(s[:, 1], (ref * ref).sum((0, 2, 3)), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cin,cout,hw', [(64, 64, 6), (512, 256, 4), (128, 32, 16)])
def test_conv_transpose_forward(dtype, cin, cout, hw):
    import hip_ops as ops
    n = 2
    x = rnd((n, cin, hw, hw), dtype, 1)
    wt = rnd((cin, cout, 4, 4), dtype, 2, (2.0 / (cin * 4)) ** 0.5)
    bias = torch.randn(cout) * 0.1
    ref = torch.relu(F.conv_transpose2d(x, wt, bias, stride=2, padding=1))
    master = wt.permute(0, 2, 3, 1).contiguous()                          # [Cin][kh][kw][Cout]
    wk = ops.pack_transpose(master.view(cin, 16, cout).cuda(), dtype).view(cout, 4, 4, cin)
    out = torch.empty((n, 2 * hw, 2 * hw, cout), dtype=dtype, device='cuda')
    ops.conv_igemm(nhwc(x, dtype), wk, out, stride=2, pad=1, mode=1, relu=True, shift=bias.cuda())
    assert torch.allclose(to_nchw(out), ref, **tol(dtype))


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cin,cout,k,stride,hw', [(64, 64, 3, 1, 10), (64, 128, 1, 1, 7), (64, 128, 3, 2, 12),
                                                  (64, 256, 1, 2, 12), (128, 64, 4, 2, 16)])
def test_data_gradient(dtype, cin, cout, k, stride, hw):
    """dgrad of conv (stride 1: flipped gather; stride 2: transposed mode) and of ConvTranspose2d (k4: gather s2)"""
    import hip_ops as ops
    n = 2
    deconv = k == 4
    if deconv:           # forward: x[cin, hw] -> out[cout, 2hw]; dgrad wrt x
        x = rnd((n, cin, hw, hw), dtype, 1).requires_grad_(True)
        wt = rnd((cin, cout, 4, 4), dtype, 2, 0.05)
        y = F.conv_transpose2d(x, wt, stride=2, padding=1)
        dy = rnd(tuple(y.shape), dtype, 3)
        y.backward(dy)
        wk = wt.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()          # [Cin][kh][kw][Cout]: used directly
        gx = torch.empty((n, hw, hw, cin), dtype=dtype, device='cuda')
        ops.conv_igemm(nhwc(dy, dtype), wk, gx, stride=2, pad=1)
    else:
        pad = k // 2
        x = rnd((n, cin, hw, hw), dtype, 1).requires_grad_(True)
        w = rnd((cout, cin, k, k), dtype, 2, 0.05)
        y = F.conv2d(x, w, stride=stride, padding=pad)
        dy = rnd(tuple(y.shape), dtype, 3)
        y.backward(dy)
        master = w.permute(0, 2, 3, 1).contiguous()                         # [Cout][kh][kw][Cin]
        wk = ops.pack_transpose(master.view(cout, k * k, cin).cuda(), dtype).view(cin, k, k, cout)
        gx = torch.empty((n, hw, hw, cin), dtype=dtype, device='cuda')
        if stride == 1:
            ops.conv_igemm(nhwc(dy, dtype), wk, gx, stride=1, pad=pad, flip=1)
        else:
            ops.conv_igemm(nhwc(dy, dtype), wk, gx, stride=2, pad=pad, mode=1)
    assert torch.allclose(to_nchw(gx), x.grad, **tol(dtype))
    # accumulate form: res aliases out
    if deconv:
        ops.conv_igemm(nhwc(dy, dtype), wk, gx, stride=2, pad=1, res=gx)
    elif stride == 1:
        ops.conv_igemm(nhwc(dy, dtype), wk, gx, stride=1, pad=pad, flip=1, res=gx)
    else:
        ops.conv_igemm(nhwc(dy, dtype), wk, gx, stride=2, pad=pad, mode=1, res=gx)
    t = tol(dtype)
    assert torch.allclose(to_nchw(gx), 2 * x.grad, rtol=2 * t['rtol'], atol=2 * t['atol'])


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cin,cout,k,stride,hw,n', [(64, 64, 3, 1, 12, 2), (64, 128, 1, 1, 9, 3), (128, 64, 3, 2, 16, 2),
                                                    (32, 32, 3, 1, 24, 2), (256, 128, 3, 1, 8, 4), (64, 32, 4, 2, 10, 2), (128, 32, 4, 2, 24, 3),
                                                    # ConvTranspose2d shapes whose four kw taps share one staged fine-row segment (wgrad3_dma_body<NT=4, QS=2>):
                                                    # map widths 16 / 8 / 32 / 64, the 128x128 and the 64x64 tile
                                                    (128, 128, 4, 2, 16, 2), (64, 64, 4, 2, 8, 2), (256, 64, 4, 2, 32, 1), (128, 128, 4, 2, 64, 1)])
def test_weight_gradient(dtype, cin, cout, k, stride, hw, n):
    import hip_ops as ops
    if k == 4:           # ConvTranspose2d: dW[cin][kh][kw][cout]
        x = rnd((n, cin, hw, hw), dtype, 1)
        wt = rnd((cin, cout, 4, 4), dtype, 2, 0.05).requires_grad_(True)
        y = F.conv_transpose2d(x, wt, stride=2, padding=1)
        dy = rnd(tuple(y.shape), dtype, 3)
        y.backward(dy)
        dw = torch.zeros((cin, 4, 4, cout), dtype=torch.float32, device='cuda')
        ops.conv_wgrad(nhwc(x, dtype), nhwc(dy, dtype), dw, 4, 4, stride=2, pad=1)
        ref = wt.grad.permute(0, 2, 3, 1)
    else:
        pad = k // 2
        x = rnd((n, cin, hw, hw), dtype, 1)
        w = rnd((cout, cin, k, k), dtype, 2, 0.05).requires_grad_(True)
        y = F.conv2d(x, w, stride=stride, padding=pad)
        dy = rnd(tuple(y.shape), dtype, 3)
        y.backward(dy)
        dw = torch.zeros((cout, k, k, cin), dtype=torch.float32, device='cuda')
        ops.conv_wgrad(nhwc(dy, dtype), nhwc(x, dtype), dw, k, k, stride=stride, pad=pad)
        ref = w.grad.permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    assert (dw.cpu() - ref).abs().max().item() / scale < (2e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cin,cout,hw,n', [(128, 128, 32, 2), (128, 256, 16, 3), (64, 64, 64, 1), (64, 64, 8, 4), (64, 32, 16, 2), (32, 64, 32, 1),
                                           (32, 32, 16, 2), (320, 128, 8, 2), (128, 128, 96, 1)])
def test_weight_gradient_three_taps_per_block(dtype, cin, cout, hw, n):
    """3x3 / stride 1 / pad 1 in a 16-bit type on image rows of 8, 16 or a multiple of 32 pixels: a block covers the three taps of
    a kernel row from ONE staged segment with a one-pixel halo (wgrad3_dma_body) -- every tile shape, every split-K policy, alone
    and grouped, channel slices of wider buffers, against torch; MSC_WGRAD_KW3=0 (single-tap blocks) must agree bit for bit in
    the structure of the result (same tolerance)"""
    from mapping_challenge_amd import _lib
    import hip_ops as ops
    x = rnd((n, cin, hw, hw), dtype, 1)
    w = rnd((cout, cin, 3, 3), dtype, 2, 0.05).requires_grad_(True)
    y = F.conv2d(x, w, padding=1)
    dy = rnd(tuple(y.shape), dtype, 3)
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    # operands as channel slices of wider buffers (the concat halves the decoder reads in place)
    xb = torch.zeros((n, hw, hw, cin + 32), dtype=dtype, device='cuda'); xb[..., 16:16 + cin] = nhwc(x, dtype)
    db = torch.zeros((n, hw, hw, cout + 64), dtype=dtype, device='cuda'); db[..., 32:32 + cout] = nhwc(dy, dtype)
    xs, ds = xb[..., 16:16 + cin], db[..., 32:32 + cout]
    ncfg = _lib.load().msc_conv_wgrad_num_cfgs()
    for c in [0] + list(range(1 if (cin % 128 == 0 and cout % 128 == 0) else 6, ncfg + 1)):
        dw = torch.zeros((cout, 3, 3, cin), dtype=torch.float32, device='cuda')
        ops.conv_wgrad(ds, xs, dw, 3, 3, stride=1, pad=1, cfg=c)
        assert (dw.cpu() - ref).abs().max().item() / scale < 1e-2, c
    for steps, cap in [(64, 128), (4, 64), (1, 32)]:
        dw = torch.zeros((cout, 3, 3, cin), dtype=torch.float32, device='cuda')
        dw1 = torch.zeros((cin, 1, 1, cin), dtype=torch.float32, device='cuda')      # a 1x1 problem in the same group (other bucket)
        ops.conv_wgrad_group([(ds, xs, dw, 3, 3, 1, 1), (xs, xs, dw1, 1, 1, 1, 0)], steps, cap)
        assert (dw.cpu() - ref).abs().max().item() / scale < 1e-2, (steps, cap)
        g1 = torch.einsum('nhwa,nhwb->ab', xs.float(), xs.float()).cpu()
        assert (dw1.cpu().view(cin, cin) - g1).abs().max().item() / g1.abs().max().item() < 1e-2


@pytest.mark.parametrize('dtype', DT)
def test_maxpool_forward_backward(dtype):
    import hip_ops as ops
    x = rnd((2, 64, 12, 16), dtype, 1).requires_grad_(True)
    y = F.max_pool2d(x, 2, 2)
    dy = rnd(tuple(y.shape), dtype, 2)
    y.backward(dy)
    xd = nhwc(x.detach(), dtype)
    assert torch.equal(to_nchw(ops.maxpool2_fwd(xd)), y.detach())
    assert torch.equal(to_nchw(ops.maxpool2_bwd(nhwc(dy, dtype), xd)), x.grad)


def test_adam_matches_torch_optim():
    import hip_ops as ops
    torch.manual_seed(0)
    p0 = torch.randn(10007)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pr], lr=5e-4, weight_decay=1e-4)
    n = (10007 + 3) // 4 * 4
    p = torch.zeros(n, device='cuda'); p[:10007] = p0.cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        g = torch.randn(10007)
        pr.grad = g.clone()
        opt.step()
        gd = torch.zeros(n, device='cuda'); gd[:10007] = g.cuda()
        ops.adam_step(p, gd, m, v, 5e-4, 0.9, 0.999, 1e-8, 1e-4, step)
        assert torch.allclose(p[:10007].cpu(), pr.data, atol=1e-6)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_adam_pack_updates_like_torch_and_writes_the_compute_copies(dtype):
    """msc_adam_pack (ABI v7): one launch over a table of tensors = torch.optim.Adam on each of them (src/models.py:57,287-292), plus
    the conv weights' compute copies: `direct` = the updated master rounded to the compute dtype, `trans` = its [B][T][A] transpose;
    ragged channel counts (a tile's edge) and a plain tensor whose length is not a multiple of 4 (a bias).  A step whose gradients
    hold an inf is skipped on the device: parameters, moments, step count untouched, loss scale halved (msc_grad_check + msc_adam_tick)."""
    from mapping_challenge_amd import _lib
    lib = _lib.load()
    dt = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}[dtype]
    torch.manual_seed(3)
    shapes = [(96, 9, 40), None, (32, 16, 128), None]           # [A][T][B] conv weights and plain tensors
    plain_n = [1003, 2]
    sizes, offs, total = [], [], 0
    pi = iter(plain_n)
    for sh in shapes:
        n = sh[0] * sh[1] * sh[2] if sh else next(pi)
        sizes.append(n); offs.append(total); total += (n + 3) // 4 * 4
    p0 = torch.randn(total)
    refs = [torch.nn.Parameter(p0[o:o + n].clone()) for o, n in zip(offs, sizes)]
    opt = torch.optim.Adam(refs, lr=5e-4, weight_decay=1e-4)
    p = p0.cuda()
    m, v, g = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
    direct = [torch.zeros(n, dtype=dtype, device='cuda') if sh else None for sh, n in zip(shapes, sizes)]
    trans = [torch.zeros(n, dtype=dtype, device='cuda') if sh else None for sh, n in zip(shapes, sizes)]
    rec = np.zeros(len(shapes), dtype=np.dtype({'names': ['off', 'n', 'direct', 'trans', 'A', 'T', 'B', 'r'],
                                                'formats': ['<i8', '<i8', '<u8', '<u8', '<i4', '<i4', '<i4', '<i4'],
                                                'offsets': [0, 8, 16, 24, 32, 36, 40, 44], 'itemsize': 48}))
    bi, bl = [], []
    for i, (sh, n, o) in enumerate(zip(shapes, sizes, offs)):
        if sh:
            use_direct = dtype != torch.float32 or i == 0          # fp32 mode: the master is its own direct copy (exercise both)
            rec[i] = (o, n, direct[i].data_ptr() if use_direct else 0, trans[i].data_ptr(), sh[0], sh[1], sh[2], 0)
            nb = sh[1] * ((sh[0] + 63) // 64) * ((sh[2] + 31) // 32)
        else:
            rec[i] = (o, n, 0, 0, 0, 0, 0, 0)
            nb = (n + 2047) // 2048
        bi.append(np.full(nb, i, np.int32)); bl.append(np.arange(nb, dtype=np.int32))
    t_items = torch.from_numpy(rec.view(np.uint8).copy()).cuda()
    t_bi, t_bl = torch.from_numpy(np.concatenate(bi)).cuda(), torch.from_numpy(np.concatenate(bl)).cuda()
    state = torch.zeros(_lib.OPT_STATE, device='cuda')
    state[_lib.OPT_LR], state[_lib.OPT_SCALE], state[_lib.OPT_GROWTH] = 5e-4, 8.0, 3.0
    stream = torch.cuda.current_stream().cuda_stream

    def update():
        _lib.check(lib.msc_grad_check(g.data_ptr(), g.numel(), state.data_ptr(), stream), 'check')
        _lib.check(lib.msc_adam_tick(state.data_ptr(), stream), 'tick')
        _lib.check(lib.msc_adam_pack(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), t_items.data_ptr(), t_bi.data_ptr(), t_bl.data_ptr(),
                                     int(t_bi.numel()), dt, 9.0, 0.9, 0.999, 1e-8, 1e-4, 0, 1.0, state.data_ptr(), stream), 'adam_pack')

    for step in range(1, 4):
        scale = float(state[_lib.OPT_SCALE])
        for r, o, n in zip(refs, offs, sizes):
            gr = torch.randn(n)
            r.grad = gr.clone()
            g[o:o + n] = (gr * scale).cuda()                       # the backward ran on scaled dlogits
        opt.step()
        update()
        for i, (r, o, n, sh) in enumerate(zip(refs, offs, sizes, shapes)):
            assert torch.allclose(p[o:o + n].cpu(), r.data, atol=2e-6), (step, i)
            if sh:
                want = p[o:o + n].view(sh)
                if rec[i]['direct']:
                    assert torch.equal(direct[i].view(sh), want.to(dtype)), (step, i)
                assert torch.equal(trans[i].view(sh[2], sh[1], sh[0]), want.permute(2, 1, 0).to(dtype)), (step, i)
    assert float(state[_lib.OPT_STEP]) == 3.0 and float(state[_lib.OPT_SCALE]) == 16.0       # 3 clean steps: the scale doubled
    # an overflowed gradient: nothing moves, the scale halves, the next clean step continues with step count 4
    keep = (p.clone(), m.clone(), v.clone())
    g[offs[2] + 17] = float('inf')
    update()
    assert torch.equal(p, keep[0]) and torch.equal(m, keep[1]) and torch.equal(v, keep[2])
    assert float(state[_lib.OPT_STEP]) == 3.0 and float(state[_lib.OPT_SCALE]) == 8.0 and float(state[_lib.OPT_SKIPPED]) == 1.0
    g[offs[2] + 17] = float('nan')
    update()
    assert torch.equal(p, keep[0]) and float(state[_lib.OPT_SCALE]) == 4.0
    g[offs[2] + 17] = 0.0
    update()
    assert float(state[_lib.OPT_STEP]) == 4.0 and float(state[_lib.OPT_SKIP]) == 0.0 and not torch.equal(p, keep[0])


def test_loss_kernels_match_oracle_and_golden(golden_dir):
    import os
    from mapping_challenge_amd.trainer import LossSpec, loss_forward_backward
    from oracle import losses_ref
    g = np.load(os.path.join(golden_dir, 'loss.npz'))
    rng = np.random.default_rng(1234)
    logits = torch.from_numpy(rng.standard_normal((2, 2, 64, 64)).astype(np.float32) * 3)
    tgt = losses_ref.synthetic_target(2, 64, 64, seed=7)
    arch = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
            'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
    arch_s = dict(arch, dice={'smooth': 1, 'dice_activation': 'sigmoid'})      # src/models.py:440-441 (round 5: was silently softmax)
    assert LossSpec.mixed(arch_s).cfg.dice_sigmoid == 1 and LossSpec.mixed(arch).cfg.dice_sigmoid == 0
    with pytest.raises(NotImplementedError):
        LossSpec.mixed(dict(arch, dice={'smooth': 1, 'dice_activation': 'tanh'}))
    for name, spec, t in (('ce', LossSpec.plain_ce(), tgt[:, :1].contiguous()), ('mixed', LossSpec.mixed(arch), tgt),
                          ('mixed_sigmoid', LossSpec.mixed(arch_s), tgt)):
        d = torch.empty((2, 2, 64, 64), device='cuda')
        loss = torch.zeros(1, device='cuda')
        sums = torch.zeros(4, dtype=torch.float64, device='cuda')
        loss_forward_backward(logits.cuda(), t.cuda(), spec, d, loss, sums)
        assert abs(loss.item() - float(g['loss_' + name])) < 2e-5
        assert np.allclose(d.cpu().numpy(), g['dlogits_' + name], atol=2e-8, rtol=1e-4)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cin,cout,k,stride,hw,n', [(128, 128, 3, 1, 20, 3), (64, 256, 1, 1, 16, 4), (32, 32, 3, 1, 24, 2), (64, 64, 3, 2, 18, 2)])
def test_every_kernel_configuration_is_correct(dtype, cin, cout, k, stride, hw, n):
    """the per-layer autotuner may pick any valid configuration: each must give the same convolution (incl. BN partial
    statistics and the transposed mode), and each wgrad configuration the same weight gradient"""
    import hip_ops as ops
    pad = k // 2
    x = rnd((n, cin, hw, hw), dtype, 1)
    w = rnd((cout, cin, k, k), dtype, 2, (2.0 / (cin * k * k)) ** 0.5)
    ref = F.conv2d(x, w, stride=stride, padding=pad)
    ho = ref.shape[2]
    xd = nhwc(x, dtype)
    wk = w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    out = torch.empty((n, ho, ho, cout), dtype=dtype, device='cuda')
    cfgs = ops.conv_valid_cfgs(xd, wk, out, stride, pad)
    assert cfgs
    for c in cfgs:
        out.zero_()
        slices = ops.conv_stats_slices(xd, wk, out, stride, pad, cfg=c)
        stats = torch.zeros((slices, cout, 2), dtype=torch.float64, device='cuda')
        ops.conv_igemm(xd, wk, out, stride=stride, pad=pad, stats=stats, cfg=c)
        assert torch.allclose(to_nchw(out), ref, **tol(dtype)), c
        assert torch.allclose(stats.sum(0)[:, 0].float().cpu(), ref.sum((0, 2, 3)), rtol=2e-3, atol=5e-2), c
    if stride == 1 and k == 3:      # transposed mode with the same operands: ConvTranspose2d(k3,s2,p1,output_padding=1)
        wt = rnd((cin, cout, 4, 4), dtype, 3, 0.05)
        reft = F.conv_transpose2d(x, wt, stride=2, padding=1)
        wkt = ops.pack_transpose(wt.permute(0, 2, 3, 1).contiguous().view(cin, 16, cout).cuda(), dtype).view(cout, 4, 4, cin)
        outt = torch.empty((n, 2 * hw, 2 * hw, cout), dtype=dtype, device='cuda')
        for c in ops.conv_valid_cfgs(xd, wkt, outt, 2, 1, mode=1):
            outt.zero_()
            ops.conv_igemm(xd, wkt, outt, stride=2, pad=1, mode=1, cfg=c)
            assert torch.allclose(to_nchw(outt), reft, **tol(dtype)), c
    dy = rnd(tuple(ref.shape), dtype, 4)
    wv = w.clone().requires_grad_(True)
    F.conv2d(x, wv, stride=stride, padding=pad).backward(dy)
    gref = wv.grad.permute(0, 2, 3, 1)
    from mapping_challenge_amd import _lib
    for c in range(1 if (cin % 128 == 0 and cout % 128 == 0) else 6, _lib.load().msc_conv_wgrad_num_cfgs() + 1):
        dw = torch.zeros((cout, k, k, cin), dtype=torch.float32, device='cuda')
        ops.conv_wgrad(nhwc(dy, dtype), xd, dw, k, k, stride=stride, pad=pad, cfg=c)
        assert (dw.cpu() - gref).abs().max().item() / gref.abs().max().item() < (2e-5 if dtype == torch.float32 else 1e-2), c


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cin,cout,hw,n,cfg,k', [(256, 1024, 16, 4, 0, 1), (256, 1024, 16, 4, 1, 1), (256, 1024, 16, 4, 33, 1), (64, 256, 24, 3, 1, 1),
                                                 (128, 512, 20, 2, 33, 1), (512, 2048, 8, 5, 0, 1), (128, 128, 9, 3, 1, 1),
                                                 (256, 256, 16, 3, 0, 3), (256, 256, 16, 3, 51, 3), (256, 256, 16, 3, 53, 3), (64, 128, 32, 2, 42, 3),
                                                 (128, 128, 32, 2, 51, 3), (64, 128, 48, 1, 53, 3)])
def test_conv_applies_batchnorm_and_relu_to_its_input_on_load(dtype, cin, cout, hw, n, cfg, k):
    """msc_conv_desc.in_bn (ABI v9): the 1x1 conv reads the RAW output y of a training-mode BatchNorm'd conv, finalises that layer's
    coefficients from its statistics slots and applies relu(scale*y + shift) to the operand in LDS -- against torch batch_norm + relu + conv2d,
    with the coefficients / saved statistics / running statistics msc_bn_apply would publish and the activation stored for the weight gradient;
    ragged pixel counts, pixel tiles that end inside a block, every tile that has the path; 3x3 (halo-tile kernel): the zero padding is of
    the activation, the halo of a patch is transformed once for the nine taps"""
    import ctypes as C
    from mapping_challenge_amd import _lib
    import hip_ops as ops
    lib = _lib.load()
    y = (rnd((n, cin, hw, hw), dtype, 1) * 1.5 + 0.25).to(dtype).float()
    w = rnd((cout, cin, k, k), dtype, 2, (2.0 / (cin * k * k)) ** 0.5)
    gamma, beta = rnd((cin,), torch.float32, 3) * 0.3 + 1.0, rnd((cin,), torch.float32, 4) * 0.2
    rm0, rv0 = rnd((cin,), torch.float32, 5) * 0.1, rnd((cin,), torch.float32, 6).abs() + 0.5
    rm, rv = rm0.clone(), rv0.clone()
    a_ref = torch.relu(F.batch_norm(y, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5))      # updates rm / rv
    a16 = a_ref.to(dtype).float()
    ref = F.conv2d(a16, w, padding=k // 2)
    yd = nhwc(y, dtype)
    yf = yd.float().double()
    count = n * hw * hw
    slots = torch.zeros((8, cin, 2), dtype=torch.float64, device='cuda')
    parts = yf.reshape(-1, cin).chunk(8, 0)                          # the sums spread over the per-XCD slots, as the producing conv leaves them
    for i, pt in enumerate(parts):
        slots[i, :, 0], slots[i, :, 1] = pt.sum(0), (pt * pt).sum(0)
    bi = _lib.BnInput()
    outs = {k: torch.full((cin,), float('nan'), device='cuda') for k in ('scale', 'shift', 'mean', 'invstd')}
    gd, bd, rmd, rvd = gamma.cuda(), beta.cuda(), rm0.cuda(), rv0.cuda()
    act = torch.full((n, hw, hw, cin), 7.0, dtype=dtype, device='cuda')
    bi.slots, bi.count, bi.gamma, bi.beta, bi.eps, bi.momentum = slots.data_ptr(), count, gd.data_ptr(), bd.data_ptr(), 1e-5, 0.1
    bi.running_mean, bi.running_var = rmd.data_ptr(), rvd.data_ptr()
    bi.scale, bi.shift, bi.save_mean, bi.save_invstd = [outs[k].data_ptr() for k in ('scale', 'shift', 'mean', 'invstd')]
    bi.out, bi.out_ld = act.data_ptr(), cin
    wd = w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    out = torch.zeros((n, hw, hw, cout), dtype=dtype, device='cuda')
    st = torch.zeros((8, cout, 2), dtype=torch.float64, device='cuda')      # the consumer's own BatchNorm statistics ride in its epilogue as usual
    ops.conv_igemm(yd, wd, out, pad=k // 2, cfg=cfg, in_bn=bi, stats=st)
    torch.cuda.synchronize()
    # ... and see only the real pixels (tile rows past the last pixel must stay zero through the on-load pass)
    assert torch.allclose(st.sum(0)[:, 0].cpu().float(), ref.sum((0, 2, 3)), rtol=2e-2, atol=2e-2 * ref.abs().sum((0, 2, 3)).max().item())
    assert torch.allclose(st.sum(0)[:, 1].cpu().float(), (ref * ref).sum((0, 2, 3)), rtol=3e-2)
    # the coefficients (the 16-bit y the kernel saw defines the statistics)
    mean = yf.mean((0, 1, 2)); var = yf.var((0, 1, 2), unbiased=False)
    inv = 1.0 / torch.sqrt(var + 1e-5)
    assert torch.allclose(outs['mean'].cpu().double(), mean.cpu(), atol=1e-5)
    assert torch.allclose(outs['invstd'].cpu().double(), inv.cpu(), rtol=1e-5)
    assert torch.allclose(outs['scale'].cpu().double(), (gamma.double() * inv.cpu()), rtol=1e-5)
    assert torch.allclose(rmd.cpu(), rm, atol=2e-3) and torch.allclose(rvd.cpu(), rv, rtol=5e-3)
    # the stored activation = what msc_bn_apply would have written, and the conv of it
    a_dev = torch.relu(yd.float() * outs['scale'] + outs['shift']).to(dtype)
    assert (act.float() - a_dev.float()).abs().max().item() <= 2 * (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * a_dev.float().abs().max().item()
    assert torch.allclose(to_nchw(act), a_ref, **tol(dtype))
    assert torch.allclose(to_nchw(out), ref, **tol(dtype))
    # not for the shapes / options it does not take
    bad = _lib.ConvDesc()
    bad.in_, bad.wt, bad.out, bad.in_ld, bad.out_ld, bad.dtype = yd.data_ptr(), wd.data_ptr(), out.data_ptr(), cin, cout, ops._dt(yd)
    bad.N, bad.Hi, bad.Wi, bad.Cin, bad.Ho, bad.Wo, bad.Cout, bad.KH, bad.KW, bad.stride, bad.pad = n, hw, hw, cin, hw, hw, cout, k, k, 1, k // 2
    bad.in_bn = C.addressof(bi)
    valid = [c for c in range(1, lib.msc_conv_num_cfgs() + 1) if lib.msc_conv_cfg_ok(C.byref(bad), c)]
    if k == 1:
        assert valid == ([1] if cout % 256 else [1, 33])
    else:
        assert valid and set(valid) <= {42, 51, 53} and (cfg == 0 or cfg in valid)
    bad.stride = 2
    assert not any(lib.msc_conv_cfg_ok(C.byref(bad), c) for c in range(1, lib.msc_conv_num_cfgs() + 1))


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('steps,cap,ordered', [(64, 128, 0), (4, 64, 0), (1, 32, 0), (0, 128, 0), (4, 64, 1), (1, 128, 1), (64, 128, 1)])
def test_grouped_weight_gradients_equal_separate_ones(dtype, steps, cap, ordered):
    """msc_wgrad_group_*: layers of different shapes (several tile buckets, 1x1 / 3x3 / strided, ragged pixel counts) in one
    launch per bucket; each gradient equals torch's, also when the group is run twice into the same buffers (+=).
    ordered (MSC_WGRAD_ORDERED): the split planes summed in a fixed order -- the same, and the same BITS from a second group"""
    import hip_ops as ops
    if ordered and steps == 0:
        pytest.skip('per-descriptor policy: covered unordered')
    shapes = [(128, 128, 3, 1, 20, 3), (64, 256, 1, 1, 16, 4), (32, 32, 3, 1, 24, 2), (64, 64, 3, 2, 18, 2), (256, 128, 1, 1, 9, 3),
              (128, 256, 3, 1, 7, 1), (32, 96, 1, 2, 10, 2),
              (512, 256, 1, 1, 11, 2), (256, 256, 1, 2, 12, 2)]      # channel counts that take the 256x256 tile (16-bit, grouped; ragged pixels)
    problems, refs = [], []
    for i, (cin, cout, k, stride, hw, n) in enumerate(shapes):
        pad = k // 2
        x = rnd((n, cin, hw, hw), dtype, 10 + i)
        wv = rnd((cout, cin, k, k), dtype, 30 + i, 0.05).requires_grad_(True)
        y = F.conv2d(x, wv, stride=stride, padding=pad)
        dy = rnd(tuple(y.shape), dtype, 50 + i)
        y.backward(dy)
        refs.append(wv.grad.permute(0, 2, 3, 1))
        dw = torch.zeros((cout, k, k, cin), dtype=torch.float32, device='cuda')
        problems.append((nhwc(dy, dtype), nhwc(x, dtype), dw, k, k, stride, pad))
    # ... and two ConvTranspose2d(k4, s2, p1) layers (four-tap blocks; P = the coarse input, Q = the fine output gradient)
    for i, (cin, cout, hw, n) in enumerate([(128, 128, 16, 2), (64, 64, 32, 1)]):
        x = rnd((n, cin, hw, hw), dtype, 70 + i)
        wt = rnd((cin, cout, 4, 4), dtype, 80 + i, 0.05).requires_grad_(True)
        y = F.conv_transpose2d(x, wt, stride=2, padding=1)
        dy = rnd(tuple(y.shape), dtype, 90 + i)
        y.backward(dy)
        refs.append(wt.grad.permute(0, 2, 3, 1))
        dw = torch.zeros((cin, 4, 4, cout), dtype=torch.float32, device='cuda')
        problems.append((nhwc(x, dtype), nhwc(dy, dtype), dw, 4, 4, 2, 1))
    launches = ops.conv_wgrad_group(problems, steps, cap, runs=2, flags=ordered)
    assert 1 <= launches <= 8 + ordered
    for pr, gref in zip(problems, refs):
        err = (pr[2].cpu() / 2 - gref).abs().max().item() / gref.abs().max().item()
        assert err < (2e-5 if dtype == torch.float32 else 1e-2), (pr[2].shape, err)
    if ordered:      # a second group on fresh buffers: bit for bit what the first run of the first group added
        again = [pr[:2] + (torch.zeros_like(pr[2]),) + pr[3:] for pr in problems]
        ops.conv_wgrad_group(again, steps, cap, runs=1, flags=1)
        third = [pr[:2] + (torch.zeros_like(pr[2]),) + pr[3:] for pr in problems]
        ops.conv_wgrad_group(third, steps, cap, runs=1, flags=1)
        for a, b in zip(again, third):
            assert torch.equal(a[2], b[2])


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cin,cout,k,hw,n,masked', [(128, 64, 1, 16, 3, True), (64, 64, 3, 20, 2, True), (256, 128, 1, 9, 2, False)])
def test_conv_epilogue_batchnorm_backward_sums(dtype, cin, cout, k, hw, n, masked):
    """stats_kind 1: a (data-gradient) conv also reduces (sum dh, sum dh*y), dh = out * [scale*y + shift > 0], per channel --
    what msc_bn_bwd_reduce would compute from the stored tensors; every configuration"""
    import ctypes as C
    from mapping_challenge_amd import _lib
    import hip_ops as ops
    lib = _lib.load()
    pad = k // 2
    x = rnd((n, cin, hw, hw), dtype, 1)
    w = rnd((cout, cin, k, k), dtype, 2, (2.0 / (cin * k * k)) ** 0.5)
    y = rnd((n, cout, hw, hw), dtype, 3)
    scale, shift = rnd((cout,), torch.float32, 4), rnd((cout,), torch.float32, 5) * 0.3
    ref = F.conv2d(x, w, padding=pad)
    m = ((y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0).float() if masked else torch.ones_like(y)
    e1, e2 = (ref * m).sum((0, 2, 3)), (ref * m * y).sum((0, 2, 3))
    xd, yd = nhwc(x, dtype), nhwc(y, dtype)
    wk = w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    out = torch.empty((n, hw, hw, cout), dtype=dtype, device='cuda')
    sc, sh = scale.cuda(), shift.cuda()
    for c in ops.conv_valid_cfgs(xd, wk, out, 1, pad):
        d = ops.ConvDesc()
        d.in_, d.wt, d.out = xd.data_ptr(), wk.data_ptr(), out.data_ptr()
        d.in_ld, d.out_ld, d.dtype, d.mode = cin, cout, ops._dt(xd), 0
        d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.cfg = n, hw, hw, cin, hw, hw, cout, k, k, 1, pad, c
        d.stats_kind, d.stats_y, d.stats_y_ld = 1, yd.data_ptr(), cout
        if masked:
            d.scale, d.shift = sc.data_ptr(), sh.data_ptr()
        d.stats = 1
        slices = lib.msc_conv_stats_slices(C.byref(d))
        stats = torch.zeros((slices, cout, 2), dtype=torch.float64, device='cuda')
        d.stats = stats.data_ptr()
        out.zero_()
        _lib.check(lib.msc_conv_igemm(C.byref(d), torch.cuda.current_stream().cuda_stream), 'conv')
        assert torch.allclose(to_nchw(out), ref, **tol(dtype)), c          # the coefficients do not touch the output
        s = stats.sum(0).float().cpu()
        t = dict(rtol=2e-3, atol=5e-2) if dtype == torch.float32 else dict(rtol=2e-2, atol=0.5)
        assert torch.allclose(s[:, 0], e1, **t) and torch.allclose(s[:, 1], e2, **t), c


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cin,cout,k,stride,hw,n,flip', [(128, 64, 3, 1, 16, 2, 1), (32, 128, 4, 2, 32, 2, 0), (32, 32, 3, 1, 32, 3, 1),
                                                         (128, 320, 3, 1, 12, 2, 1), (32, 64, 4, 2, 26, 3, 0),
                                                         (32, 128, 4, 2, 256, 8, 0)])      # 1024 patches: four per persistent block of configuration 59
def test_conv_epilogue_relu_backward_and_bias_sums(dtype, cin, cout, k, stride, hw, n, flip):
    """stats_kind 2: a data-gradient conv stores its result masked by [act > 0] (ReLU backward of the layer whose activation it
    is given) and reduces the per-channel sums of what it stored (that layer's bias gradient, folded by
    msc_bias_slots_finalize) -- the pass msc_relu_bias_grad would make over the tensors; every valid configuration, incl. the
    halo-tile kernel of the 32-channel layers and the stride-2 gather form of the ConvTranspose2d data gradient"""
    import ctypes as C
    from mapping_challenge_amd import _lib
    import hip_ops as ops
    lib = _lib.load()
    pad = 1
    ho = (hw + 2 * pad - k) // stride + 1
    x = rnd((n, cin, hw, hw), dtype, 1)
    w = rnd((cout, cin, k, k), dtype, 2, (2.0 / (cin * k * k)) ** 0.5)
    act = torch.relu(rnd((n, cout, ho, ho), dtype, 3))
    ref = F.conv2d(x, w.flip(2, 3) if flip else w, stride=stride, padding=pad) * (act > 0)
    xd, ad = nhwc(x, dtype), nhwc(act, dtype)
    wk = w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    out = torch.empty((n, ho, ho, cout), dtype=dtype, device='cuda')
    tried = 0
    for c in range(1, lib.msc_conv_num_cfgs() + 1):
        d = ops.ConvDesc()
        d.in_, d.wt, d.out = xd.data_ptr(), wk.data_ptr(), out.data_ptr()
        d.in_ld, d.out_ld, d.dtype, d.mode, d.flip = cin, cout, ops._dt(xd), 0, flip
        d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.cfg = n, hw, hw, cin, ho, ho, cout, k, k, stride, pad, c
        d.stats_kind, d.stats_y, d.stats_y_ld = 2, ad.data_ptr(), cout
        stats = torch.zeros((_lib.BN_SLOTS, cout, 2), dtype=torch.float64, device='cuda')
        d.stats = stats.data_ptr()
        if not lib.msc_conv_cfg_ok(C.byref(d), c):
            continue
        tried += 1
        out.fill_(7.0)
        _lib.check(lib.msc_conv_igemm(C.byref(d), torch.cuda.current_stream().cuda_stream), 'conv')
        assert torch.allclose(to_nchw(out), ref, **tol(dtype)), c
        # the sums are taken before the store rounds to the dtype; against a pass over the stored tensor that is rounding noise
        stored = to_nchw(out).double().sum((0, 2, 3))
        db = torch.zeros(cout + 3, device='cuda')
        _lib.check(lib.msc_bias_slots_finalize(stats.data_ptr(), cout, db.data_ptr(), cout, torch.cuda.current_stream().cuda_stream), 'fin')
        assert (stats[:, :, 1] == 0).all(), c
        # (the rounding noise of the stored tensor grows with the square root of the pixels summed: 512 in the first shapes, 131 072 in the last one)
        noise = max(1.0, (n * ho * ho / 512.0) ** 0.5)
        assert torch.allclose(db[:cout].cpu().double(), stored, rtol=2e-3, atol=(5e-2 if dtype == torch.float32 else 0.5) * noise), c
        assert torch.allclose(db[:cout].cpu().double(), ref.double().sum((0, 2, 3)), rtol=2e-3, atol=(5e-2 if dtype == torch.float32 else 0.5)), c      # against the unrounded result
        assert (db[cout:] == 0).all()
        # several layers in one launch (what the program emits after the decoder's backward)
        items = (_lib.BiasSlotsItem * 2)()
        dbm = torch.zeros(2, cout, device='cuda')
        for it, row, cc in ((items[0], dbm[0], cout), (items[1], dbm[1], 32)):
            it.slots, it.db, it.Cs, it.C = stats.data_ptr(), row.data_ptr(), cout, cc
        _lib.check(lib.msc_bias_slots_finalize_multi(items, 2, torch.cuda.current_stream().cuda_stream), 'fin multi')
        assert torch.equal(dbm[0], db[:cout]) and torch.equal(dbm[1, :32], db[:32]) and (dbm[1, 32:] == 0).all(), c
        assert lib.msc_bias_slots_finalize_multi(items, _lib.BIAS_SLOTS_MAX + 1, torch.cuda.current_stream().cuda_stream) != 0
        # a channel sub-range of wider slots (the decoder half of a concat buffer)
        if cout >= 64:
            db2 = torch.ones(32, device='cuda')
            _lib.check(lib.msc_bias_slots_finalize(stats.data_ptr() + 16 * 32, cout, db2.data_ptr(), 32, torch.cuda.current_stream().cuda_stream), 'fin')
            assert torch.allclose(db2.cpu().double() - 1, stored[32:64], rtol=2e-3, atol=(5e-2 if dtype == torch.float32 else 0.5) * noise), c
    assert tried >= 2
    if cin == 32 and cout == 32 and dtype != torch.float32:
        d.cfg = _lib.CFG_HALO
        assert lib.msc_conv_cfg_ok(C.byref(d), _lib.CFG_HALO)          # the halo-tile kernel carries the mask and the sums too
    # no residual / scale / ReLU with this kind
    d.cfg, d.relu = 0, 1
    assert lib.msc_conv_igemm(C.byref(d), torch.cuda.current_stream().cuda_stream) != 0


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('n,hw,c,with_bias_in,ordered', [(2, 24, 32, True, 0), (3, 17, 32, False, 0), (1, 64, 64, True, 0), (2, 9, 16, True, 0),
                                                         (2, 24, 32, True, 1), (3, 65, 32, False, 1), (1, 64, 64, True, 1)])
def test_final_1x1_backward(dtype, n, hw, c, with_bias_in, ordered):
    """msc_final_bwd against torch autograd of ReLU -> Conv2d(C, 2, 1): the input gradient masked by the producer's ReLU, the
    1x1 weight / bias gradients, and the producer's bias gradient summed from the stored (rounded) input gradient.
    ordered: the per-block sums through ordered_ws, added in block order -- the same values, and the same bits twice"""
    from mapping_challenge_amd import _lib
    import hip_ops as ops
    lib = _lib.load()
    if c * (4 if dtype == torch.float32 else 2) % 16:
        pytest.skip('C must keep 16-byte channel vectors')
    a = torch.relu(rnd((n, c, hw, hw), dtype, 1))
    w = rnd((2, c), torch.float32, 2, 0.3)
    g = rnd((n, 2, hw, hw), torch.float32, 3)
    pre = a.clone().requires_grad_(True)
    wt = w.clone().requires_grad_(True)
    b = torch.zeros(2, requires_grad=True)
    y = F.conv2d(pre, wt.view(2, c, 1, 1), b)
    y.backward(g)
    din_ref = pre.grad * (a > 0)
    ad = nhwc(a, dtype)
    din = torch.full((n, hw, hw, c), 5.0, dtype=dtype, device='cuda')
    dw = torch.ones(2, c, device='cuda')
    db = torch.ones(2, device='cuda')
    dbin = torch.ones(c, device='cuda') if with_bias_in else None
    st = torch.cuda.current_stream().cuda_stream
    gd, wd = g.cuda(), w.cuda()          # kept alive: a temporary's memory is recycled by the next allocation
    ws = torch.full((_lib.FINAL_BWD_WS_ROWS * (3 * c + 2),), float('nan'), device='cuda') if ordered else None
    _lib.check(lib.msc_final_bwd(gd.data_ptr(), ad.data_ptr(), c, wd.data_ptr(), din.data_ptr(), c, dw.data_ptr(), db.data_ptr(),
                                 dbin.data_ptr() if with_bias_in else None, ws.data_ptr() if ordered else None, ops._dt(ad), n, hw, hw, c, st), 'final_bwd')
    torch.cuda.synchronize()
    if ordered:
        dw2, db2 = torch.ones(2, c, device='cuda'), torch.ones(2, device='cuda')
        _lib.check(lib.msc_final_bwd(gd.data_ptr(), ad.data_ptr(), c, wd.data_ptr(), din.data_ptr(), c, dw2.data_ptr(), db2.data_ptr(),
                                     None, ws.data_ptr(), ops._dt(ad), n, hw, hw, c, st), 'final_bwd')
        torch.cuda.synchronize()
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
    assert torch.allclose(to_nchw(din), din_ref, **tol(dtype))
    t = dict(rtol=1e-4, atol=1e-3)
    assert torch.allclose(dw.cpu() - 1, wt.grad, **t)
    assert torch.allclose(db.cpu() - 1, b.grad, **t)
    if with_bias_in:
        assert torch.allclose(dbin.cpu() - 1, to_nchw(din).sum((0, 2, 3)), rtol=1e-4, atol=2e-2)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cin,cout,hw,n', [(64, 64, 16, 2), (128, 128, 32, 2), (320, 128, 16, 3), (256, 256, 16, 2), (128, 320, 32, 1),
                                           (64, 128, 48, 1)])
def test_halo_tile_kernel_for_3x3_convs_of_any_width(dtype, cin, cout, hw, n):
    """configurations 42..46 (conv3x3_halo_dma_kernel: the halo of a 64-channel chunk is fetched once for the nine taps): plain,
    flipped taps + accumulate (data gradient), folded BN + ReLU, channel slices of wider buffers, and the three kinds of
    epilogue statistics, against torch"""
    import ctypes as C
    from mapping_challenge_amd import _lib
    import hip_ops as ops
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    x = rnd((n, cin, hw, hw), dtype, 1)
    w = rnd((cout, cin, 3, 3), dtype, 2, (2.0 / (cin * 9)) ** 0.5)
    prev = rnd((n, cout, hw, hw), dtype, 3)
    act = torch.relu(rnd((n, cout, hw, hw), dtype, 4))
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1
    ref = F.conv2d(x, w, padding=1)
    ref_flip = F.conv2d(x, w.flip(2, 3), padding=1)
    # operands as channel slices: input channels [8, 8+cin) of a wider buffer, output channels [16, 16+cout)
    xin = torch.zeros((n, hw, hw, cin + 24), dtype=dtype, device='cuda')
    xin[..., 8:8 + cin] = nhwc(x, dtype)
    xs = xin[..., 8:8 + cin]
    wk = w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    obuf = torch.zeros((n, hw, hw, cout + 32), dtype=dtype, device='cuda')
    resd, actd = nhwc(prev, dtype), nhwc(act, dtype)
    sc, sh = scale.cuda(), shift.cuda()

    def desc(cfg):
        d = ops.ConvDesc()
        d.in_, d.wt, d.out = xs.data_ptr(), wk.data_ptr(), obuf.data_ptr() + 16 * obuf.element_size()
        d.in_ld, d.out_ld, d.dtype, d.mode = cin + 24, cout + 32, ops._dt(xin), 0
        d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.cfg = n, hw, hw, cin, hw, hw, cout, 3, 3, 1, 1, cfg
        return d

    def result():
        assert (obuf[..., :16] == 0).all() and (obuf[..., 16 + cout:] == 0).all()      # nothing outside the slice
        return to_nchw(obuf[..., 16:16 + cout])

    tried = 0
    for cfg in list(range(42, 47)) + list(range(51, 57)):
        d = desc(cfg)
        if not lib.msc_conv_cfg_ok(C.byref(d), cfg):
            continue
        tried += 1
        obuf.zero_()
        _lib.check(lib.msc_conv_igemm(C.byref(d), st), 'plain')
        assert torch.allclose(result(), ref, **tol(dtype)), cfg
        # data-gradient form: flipped taps, accumulated onto what the output already holds
        d = desc(cfg)
        obuf[..., 16:16 + cout] = resd
        d.flip, d.res, d.res_ld = 1, d.out, cout + 32
        _lib.check(lib.msc_conv_igemm(C.byref(d), st), 'flip+res')
        assert torch.allclose(result(), ref_flip + prev, **tol(dtype)), cfg
        # folded BatchNorm + ReLU, with the BN partial statistics of the raw accumulators
        d = desc(cfg)
        obuf.zero_()
        stats = torch.zeros((_lib.BN_SLOTS, cout, 2), dtype=torch.float64, device='cuda')
        d.scale, d.shift, d.relu, d.stats = sc.data_ptr(), sh.data_ptr(), 1, stats.data_ptr()
        _lib.check(lib.msc_conv_igemm(C.byref(d), st), 'bn')
        assert torch.allclose(result(), torch.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)), **tol(dtype)), cfg
        ssum = stats.sum(0).float().cpu()
        assert torch.allclose(ssum[:, 0], ref.sum((0, 2, 3)), rtol=2e-2, atol=0.5), cfg
        assert torch.allclose(ssum[:, 1], (ref * ref).sum((0, 2, 3)), rtol=2e-2, atol=0.5), cfg
        # stats_kind 1 (BatchNorm-backward sums against y = act, masked by scale*y+shift > 0) and 2 (ReLU backward + bias sums)
        d = desc(cfg)
        obuf.zero_(); stats.zero_()
        d.flip, d.stats, d.stats_kind, d.stats_y, d.stats_y_ld = 1, stats.data_ptr(), 1, actd.data_ptr(), cout
        d.scale, d.shift = sc.data_ptr(), sh.data_ptr()
        _lib.check(lib.msc_conv_igemm(C.byref(d), st), 'kind1')
        assert torch.allclose(result(), ref_flip, **tol(dtype)), cfg
        m1 = ((act * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0).float()
        ssum = stats.sum(0).float().cpu()
        assert torch.allclose(ssum[:, 0], (ref_flip * m1).sum((0, 2, 3)), rtol=2e-2, atol=0.5), cfg
        assert torch.allclose(ssum[:, 1], (ref_flip * m1 * act).sum((0, 2, 3)), rtol=2e-2, atol=0.5), cfg
        d = desc(cfg)
        obuf.zero_(); stats.zero_()
        d.flip, d.stats, d.stats_kind, d.stats_y, d.stats_y_ld = 1, stats.data_ptr(), 2, actd.data_ptr(), cout
        _lib.check(lib.msc_conv_igemm(C.byref(d), st), 'kind2')
        assert torch.allclose(result(), ref_flip * (act > 0), **tol(dtype)), cfg
        assert torch.allclose(stats.sum(0)[:, 0].float().cpu(), (ref_flip * (act > 0)).sum((0, 2, 3)), rtol=2e-2, atol=0.5), cfg
    assert tried >= 2, tried


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('hw,n,flip,with_res,relu', [(32, 3, False, False, True), (48, 2, True, True, False), (16, 1, True, False, False)])
def test_halo_tile_kernel_for_32_channel_3x3(hw, n, flip, with_res, relu, dtype):
    """the halo-tile configuration (last one) of the 32-channel 3x3 layers: bias / scale, ReLU, residual accumulate and the
    flipped-tap form used by the data gradient, against torch"""
    from mapping_challenge_amd import _lib
    import hip_ops as ops
    lib = _lib.load()
    halo = _lib.CFG_HALO
    x = rnd((n, 32, hw, hw), dtype, 1)
    w = rnd((32, 32, 3, 3), dtype, 2, 0.08)
    bias, scale = rnd((32,), torch.float32, 3), rnd((32,), torch.float32, 4) * 0.2 + 1.0
    prev = rnd((n, 32, hw, hw), dtype, 5)
    wref = w.flip(2, 3) if flip else w
    ref = F.conv2d(x, wref, padding=1) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    if with_res:
        ref = ref + prev
    if relu:
        ref = torch.relu(ref)
    xd = nhwc(x, dtype)
    wk = w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    out = nhwc(prev, dtype) if with_res else torch.empty((n, hw, hw, 32), dtype=dtype, device='cuda')
    assert halo in ops.conv_valid_cfgs(xd, wk, out, 1, 1)
    ops.conv_igemm(xd, wk, out, stride=1, pad=1, flip=flip, relu=relu, scale=scale.cuda(), shift=bias.cuda(),
                   res=out if with_res else None, cfg=halo)
    assert torch.allclose(to_nchw(out), ref, **tol(dtype))
    # a channel slice of a wider buffer as input (ld > C)
    wide = torch.zeros((n, hw, hw, 96), dtype=dtype, device='cuda')
    wide[..., 32:64] = xd
    out2 = torch.empty((n, hw, hw, 32), dtype=dtype, device='cuda')
    ops.conv_igemm(wide[..., 32:64], wk, out2, stride=1, pad=1, flip=flip, cfg=halo)
    assert torch.allclose(to_nchw(out2), F.conv2d(x, wref, padding=1), **tol(dtype))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('hw,n,relu,with_res,cout', [(16, 2, True, False, 32), (32, 1, False, True, 32), (8, 3, True, False, 32),
                                                     (16, 2, True, False, 128), (8, 3, False, True, 128), (32, 2, True, False, 64)])
def test_halo_tile_kernel_for_the_128_to_32_transposed_conv(hw, n, relu, with_res, cout, dtype):
    """ConvTranspose2d(128, 32 G, k4, s2, p1) + bias (+ReLU) through the halo-tile configuration, against torch: dec1's 128 -> 32 and, since
    round 5, groups of 32 output channels per block (dec2's 128 -> 128: G = 4), written into a channel slice of a wider buffer"""
    from mapping_challenge_amd import _lib
    import hip_ops as ops
    cfg = _lib.CFG_HALO_T
    x = rnd((n, 128, hw, 2 * hw), dtype, 1)                     # non-square: 8 | H, 16 | W
    wt = rnd((128, cout, 4, 4), dtype, 2, 0.05)
    bias = rnd((cout,), torch.float32, 3)
    prev = rnd((n, cout, 2 * hw, 4 * hw), dtype, 4)
    ref = F.conv_transpose2d(x, wt, stride=2, padding=1) + bias.view(1, -1, 1, 1)
    if with_res:
        ref = ref + prev
    if relu:
        ref = torch.relu(ref)
    xd = nhwc(x, dtype)
    wk = ops.pack_transpose(wt.permute(0, 2, 3, 1).contiguous().view(128, 16, cout).cuda(), dtype).view(cout, 4, 4, 128)
    out = nhwc(prev, dtype) if with_res else torch.empty((n, 2 * hw, 4 * hw, cout), dtype=dtype, device='cuda')
    assert cfg in ops.conv_valid_cfgs(xd, wk, out, 2, 1, mode=1)
    ops.conv_igemm(xd, wk, out, stride=2, pad=1, mode=1, relu=relu, shift=bias.cuda(), res=out if with_res else None, cfg=cfg)
    assert torch.allclose(to_nchw(out), ref, **tol(dtype))
    # the same layer through the implicit-GEMM configurations gives the same result (the tuner may pick either)
    other = [c for c in ops.conv_valid_cfgs(xd, wk, out, 2, 1, mode=1) if c != cfg][:2]
    for c in other:
        out2 = nhwc(prev, dtype) if with_res else torch.empty_like(out)
        ops.conv_igemm(xd, wk, out2, stride=2, pad=1, mode=1, relu=relu, shift=bias.cuda(), res=out2 if with_res else None, cfg=c)
        assert torch.allclose(to_nchw(out2), ref, **tol(dtype)), c


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('c,hw,n,relu,with_res', [(64, 12, 3, True, False), (256, 8, 4, True, True), (128, 10, 2, False, False), (1024, 4, 8, True, True)])
def test_batchnorm_training_forward_and_backward_with_prologue_finalize(dtype, c, hw, n, relu, with_res):
    """BatchNorm2d (+residual)(+ReLU) in training mode and its backward against torch autograd: statistics accumulated per XCD
    slot (here by msc_bn_bwd_reduce and, for the forward sums, scattered by hand over the slots) and finalised in the
    prologues of msc_bn_apply / msc_bn_bwd_apply; running statistics, dgamma, dbeta, the residual gradient"""
    from mapping_challenge_amd import _lib
    lib = _lib.load()
    dt = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}[dtype]
    st = torch.cuda.current_stream().cuda_stream
    y = rnd((n, c, hw, hw), dtype, 1, 1.5) + 0.3
    y = y.to(dtype).float()
    res = rnd((n, c, hw, hw), dtype, 2) if with_res else None
    gamma, beta = torch.rand(c) + 0.5, torch.randn(c) * 0.2
    bn = torch.nn.BatchNorm2d(c)
    bn.weight.data, bn.bias.data = gamma.clone(), beta.clone()
    bn.train()
    yr = y.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if with_res else None
    z = bn(yr) + (rr if with_res else 0)
    o = torch.relu(z) if relu else z
    dout = rnd(tuple(o.shape), dtype, 3)
    o.backward(dout)
    # device side
    pixels = n * hw * hw
    yd, outd = nhwc(y, dtype), torch.empty((n, hw, hw, c), dtype=dtype, device='cuda')
    resd = nhwc(res, dtype) if with_res else None
    flat = y.permute(0, 2, 3, 1).reshape(-1, c).double()
    slots = torch.zeros((_lib.BN_SLOTS, c, 2), dtype=torch.float64, device='cuda')
    # the forward sums as the conv epilogue leaves them: spread over the slots (here: pixel p into slot p % 8)
    for x_ in range(_lib.BN_SLOTS):
        part = flat[x_::_lib.BN_SLOTS]
        slots[x_, :, 0] = part.sum(0).cuda()
        slots[x_, :, 1] = (part * part).sum(0).cuda()
    g_d, b_d = gamma.cuda(), beta.cuda()
    rm, rv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
    scale, shift, mean, invstd = (torch.empty(c, device='cuda') for _ in range(4))
    # residual joins: msc_bn_apply also leaves the ReLU mask as bytes (ABI v7), which the backward reads instead of `out` (relu mode 3)
    use_mask, ce = bool(relu and with_res), 16 // yd.element_size()
    maskd = torch.zeros((n, hw, hw, c // ce), dtype=torch.uint8, device='cuda')
    _lib.check(lib.msc_bn_apply(yd.data_ptr(), c, resd.data_ptr() if with_res else None, c if with_res else 0, outd.data_ptr(), c, slots.data_ptr(),
                                pixels, g_d.data_ptr(), b_d.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                mean.data_ptr(), invstd.data_ptr(), maskd.data_ptr() if use_mask else None, c // ce if use_mask else 0, int(relu), dt, pixels, c, st), 'msc_bn_apply')
    assert torch.allclose(to_nchw(outd), o.detach(), **tol(dtype))
    assert torch.allclose(rm.cpu(), bn.running_mean, atol=1e-5) and torch.allclose(rv.cpu(), bn.running_var, rtol=1e-5, atol=1e-6)
    # backward: reduce -> apply; mask from `out` (relu 1) with a residual, recomputed from y (relu 2) without
    mask = 0 if not relu else (1 if with_res else 2)
    if use_mask:
        bits = (outd.float().view(n, hw, hw, c // ce, ce) > 0).to(torch.int32) << torch.arange(ce, device='cuda', dtype=torch.int32)
        assert torch.equal(maskd.to(torch.int32), bits.sum(-1))
    doutd = nhwc(dout, dtype)
    bslots = torch.zeros((_lib.BN_SLOTS, c, 2), dtype=torch.float64, device='cuda')
    _lib.check(lib.msc_bn_bwd_reduce(doutd.data_ptr(), c, outd.data_ptr(), c, yd.data_ptr(), c, mask, scale.data_ptr(), shift.data_ptr(), bslots.data_ptr(),
                                     dt, pixels, c, st), 'msc_bn_bwd_reduce')
    dgamma, dbeta = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
    dyd = torch.empty_like(yd)
    dresd = torch.empty_like(yd) if with_res else None
    # with a residual: also the BatchNorm-backward sums of the layer that produced it (res_y / res_slots), against a tensor ry
    ry = rnd((n, c, hw, hw), dtype, 9)
    ryd, rslots = nhwc(ry, dtype), torch.zeros((_lib.BN_SLOTS, c, 2), dtype=torch.float64, device='cuda')
    _lib.check(lib.msc_bn_bwd_apply(doutd.data_ptr(), c, maskd.data_ptr() if use_mask else outd.data_ptr(), c // ce if use_mask else c, yd.data_ptr(), c,
                                    3 if use_mask else mask, scale.data_ptr(), shift.data_ptr(), bslots.data_ptr(),
                                    pixels, g_d.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dyd.data_ptr(), c,
                                    dresd.data_ptr() if with_res else None, c if with_res else 0, 0, ryd.data_ptr() if with_res else None, c if with_res else 0,
                                    rslots.data_ptr() if with_res else None, dt, pixels, c, st), 'msc_bn_bwd_apply')
    t = tol(dtype)
    # the stored `out` is rounded to the dtype: a 16-bit out of exactly 0 vs a tiny positive reference value may flip a mask
    # element; gradients are compared with that allowance (a handful of elements) in the 16-bit modes
    bad = (~torch.isclose(to_nchw(dyd), yr.grad, **t)).float().mean().item()
    assert bad <= (0.0 if dtype == torch.float32 else 2e-3), bad
    scale_g = max(1.0, bn.weight.grad.abs().max().item())
    assert (dgamma.cpu() - bn.weight.grad).abs().max().item() < (2e-4 if dtype == torch.float32 else 5e-2) * scale_g
    assert (dbeta.cpu() - bn.bias.grad).abs().max().item() < (2e-4 if dtype == torch.float32 else 5e-2) * max(1.0, bn.bias.grad.abs().max().item())
    if with_res:
        assert (~torch.isclose(to_nchw(dresd), rr.grad, **t)).float().mean().item() <= (0.0 if dtype == torch.float32 else 2e-3)
        dh = to_nchw(dresd).double()
        rs = rslots.sum(0).cpu()
        e1, e2 = dh.sum((0, 2, 3)), (dh * ry.double()).sum((0, 2, 3))
        assert torch.allclose(rs[:, 0], e1, rtol=1e-3, atol=1e-2) and torch.allclose(rs[:, 1], e2, rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('c,hw,n', [(64, 12, 3), (32, 6, 2), (128, 10, 1)])
def test_stem_batchnorm_relu_maxpool_fused_forward_and_backward(dtype, c, hw, n):
    """msc_bn_apply_pool / msc_bn_pool_bwd_reduce / msc_bn_pool_bwd_apply (ABI v7): BatchNorm2d (training) + ReLU + MaxPool2d(2,2) of the stem
    (src/unet_models.py:360-363) against torch autograd -- pooled output, running statistics, the gradient w.r.t. the conv output
    (MaxPool routing to the first maximum + ReLU mask + BatchNorm backward), dgamma, dbeta; windows that are entirely non-positive and
    windows with ties are in the data"""
    from mapping_challenge_amd import _lib
    lib = _lib.load()
    dt = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}[dtype]
    st = torch.cuda.current_stream().cuda_stream
    y = rnd((n, c, hw, hw), dtype, 1, 1.5) + 0.1
    y[:, :, 0:2, 0:2] = -3.0                      # a window the ReLU zeroes completely: its gradient stops
    y[:, :, 2:4, 2:4] = y[:, :, 2:3, 2:3]         # a window of four equal values: the first one gets the gradient
    y = y.to(dtype).float()
    gamma, beta = torch.rand(c) + 0.5, torch.randn(c) * 0.2
    bn = torch.nn.BatchNorm2d(c)
    bn.weight.data, bn.bias.data = gamma.clone(), beta.clone()
    bn.train()
    yr = y.clone().requires_grad_(True)
    o = F.max_pool2d(torch.relu(bn(yr)), 2, 2)
    dout = rnd(tuple(o.shape), dtype, 3)
    o.backward(dout)
    pixels, ho = n * hw * hw, hw // 2
    yd, outd = nhwc(y, dtype), torch.empty((n, ho, ho, c), dtype=dtype, device='cuda')
    flat = y.permute(0, 2, 3, 1).reshape(-1, c).double()
    slots = torch.zeros((_lib.BN_SLOTS, c, 2), dtype=torch.float64, device='cuda')
    for x_ in range(_lib.BN_SLOTS):
        part = flat[x_::_lib.BN_SLOTS]
        slots[x_, :, 0] = part.sum(0).cuda()
        slots[x_, :, 1] = (part * part).sum(0).cuda()
    g_d, b_d = gamma.cuda(), beta.cuda()
    rm, rv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
    scale, shift, mean, invstd = (torch.empty(c, device='cuda') for _ in range(4))
    _lib.check(lib.msc_bn_apply_pool(yd.data_ptr(), c, outd.data_ptr(), c, slots.data_ptr(), pixels, g_d.data_ptr(), b_d.data_ptr(), 1e-5, 0.1,
                                     rm.data_ptr(), rv.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dt, n, ho, ho,
                                     c, st), 'msc_bn_apply_pool')
    assert torch.allclose(to_nchw(outd), o.detach(), **tol(dtype))
    assert torch.allclose(rm.cpu(), bn.running_mean, atol=1e-5) and torch.allclose(rv.cpu(), bn.running_var, rtol=1e-5, atol=1e-6)
    doutd = nhwc(dout, dtype)
    bslots = torch.zeros((_lib.BN_SLOTS, c, 2), dtype=torch.float64, device='cuda')
    _lib.check(lib.msc_bn_pool_bwd_reduce(doutd.data_ptr(), c, yd.data_ptr(), c, scale.data_ptr(), shift.data_ptr(), bslots.data_ptr(), dt, n, ho, ho, c,
                                          st), 'msc_bn_pool_bwd_reduce')
    dgamma, dbeta = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
    _lib.check(lib.msc_bn_pool_bwd_apply(doutd.data_ptr(), c, yd.data_ptr(), c, scale.data_ptr(), shift.data_ptr(), bslots.data_ptr(), pixels,
                                         g_d.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dt, n, ho, ho, c, st),
               'msc_bn_pool_bwd_apply')
    t = tol(dtype)
    # the device takes the maximum over fp32 activations computed from ITS coefficients: a near-tie inside a window may route the gradient
    # to the other element than torch's fp32 chain does (16-bit inputs make near-ties common) -- allowed for a handful of elements
    bad = (~torch.isclose(to_nchw(yd), yr.grad, **t)).float().mean().item()
    assert bad <= (1e-4 if dtype == torch.float32 else 4e-3), bad
    assert (dgamma.cpu() - bn.weight.grad).abs().max().item() < (2e-4 if dtype == torch.float32 else 5e-2) * max(1.0, bn.weight.grad.abs().max().item())
    assert (dbeta.cpu() - bn.bias.grad).abs().max().item() < (2e-4 if dtype == torch.float32 else 5e-2) * max(1.0, bn.bias.grad.abs().max().item())
    # the all-negative window passes no gradient through the pool; the BatchNorm backward's mean terms still reach it
    assert yr.grad[:, :, 0:2, 0:2].abs().max().item() < 1.0


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cin,cout,k,hw,n,join', [(128, 64, 3, 8, 2, False), (256, 128, 1, 6, 3, True), (64, 64, 3, 10, 1, True)])
def test_transposed_mode_epilogue_carries_the_batchnorm_backward_sums(dtype, cin, cout, k, hw, n, join):
    """round 4: stats_kind 1 in TRANSPOSED mode (the stride-2 data gradients of a stage's first block: 3x3 and the 1x1 downsample) -- the
    plain form (mask from scale*y + shift) and the residual-join form (out = acc + res in place, mask from stats_z), every valid
    configuration, against the stride-2 conv's data gradient computed by torch"""
    import ctypes as C
    from mapping_challenge_amd import _lib
    import hip_ops as ops
    lib = _lib.load()
    pad = k // 2
    # forward conv: fine [n, cout_f = `cout`... here the data gradient maps dy [n, cin, hw, hw] (coarse) -> dx [n, cout, 2hw, 2hw] (fine)
    xf = rnd((n, cout, 2 * hw, 2 * hw), dtype, 1).requires_grad_(True)
    w = rnd((cin, cout, k, k), dtype, 2, 0.05)                              # forward weight [Cout_fwd = cin][Cin_fwd = cout][k][k]
    yf = F.conv2d(xf, w, stride=2, padding=pad)
    assert yf.shape[2] == hw
    dy = rnd(tuple(yf.shape), dtype, 3)
    yf.backward(dy)
    ref = xf.grad                                                           # [n, cout, 2hw, 2hw]
    master = w.permute(0, 2, 3, 1).contiguous()                             # [cin][kh][kw][cout]
    wk = ops.pack_transpose(master.view(cin, k * k, cout).cuda(), dtype).view(cout, k, k, cin)
    ysave = rnd((n, cout, 2 * hw, 2 * hw), dtype, 4)
    scale, shift = rnd((cout,), torch.float32, 5), rnd((cout,), torch.float32, 6) * 0.3
    z = torch.relu(rnd((n, cout, 2 * hw, 2 * hw), dtype, 7))
    g0 = rnd((n, cout, 2 * hw, 2 * hw), dtype, 8)
    if join:
        tot = ref + g0
        m = (z > 0).float()
    else:
        tot = ref
        m = ((ysave * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0).float()
    e1, e2 = (tot * m).sum((0, 2, 3)), (tot * m * ysave).sum((0, 2, 3))
    dyd, yd, zd = nhwc(dy, dtype), nhwc(ysave, dtype), nhwc(z, dtype)
    out = torch.empty((n, 2 * hw, 2 * hw, cout), dtype=dtype, device='cuda')
    sc, sh = scale.cuda(), shift.cuda()
    stream = torch.cuda.current_stream().cuda_stream
    tried = 0
    for c in [0] + ops.conv_valid_cfgs(dyd, wk, out, 2, pad, mode=1):
        d = ops.ConvDesc()
        d.in_, d.wt, d.out = dyd.data_ptr(), wk.data_ptr(), out.data_ptr()
        d.in_ld, d.out_ld, d.dtype, d.mode = cin, cout, ops._dt(dyd), 1
        d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.cfg = n, hw, hw, cin, 2 * hw, 2 * hw, cout, k, k, 2, pad, c
        d.stats_kind, d.stats_y, d.stats_y_ld = 1, yd.data_ptr(), cout
        if join:
            d.stats_z, d.stats_z_ld = zd.data_ptr(), cout
            out.copy_(nhwc(g0, dtype))
            d.res, d.res_ld = out.data_ptr(), cout
        else:
            d.scale, d.shift = sc.data_ptr(), sh.data_ptr()
            out.zero_()
        stats = torch.zeros((_lib.BN_SLOTS, cout, 2), dtype=torch.float64, device='cuda')
        d.stats = stats.data_ptr()
        if c and not lib.msc_conv_cfg_ok(C.byref(d), c):
            continue
        _lib.check(lib.msc_conv_igemm(C.byref(d), stream), 'conv')
        tried += 1
        assert torch.allclose(to_nchw(out), tot.detach(), **tol(dtype)), c
        s_ = stats.sum(0).float().cpu()
        t = dict(rtol=2e-3, atol=5e-2) if dtype == torch.float32 else dict(rtol=2e-2, atol=0.5)
        assert torch.allclose(s_[:, 0], e1.detach(), **t) and torch.allclose(s_[:, 1], e2.detach(), **t), c
    assert tried >= 1


def test_conv_and_wgrad_beyond_2gib_run_as_image_ranges():
    """the kernels address their operands with 31-bit byte offsets (buffer descriptors): a 2.4 GB input (9 images of
    512 x 512 x 512 channels in bf16) is run as consecutive image ranges -- conv with a statistics epilogue (accumulated
    across the ranges), and the weight gradient (a sum over images) -- against torch on the device"""
    import hip_ops as ops
    n, hw, cin, cout = 9, 512, 512, 32
    g = torch.Generator(device='cuda').manual_seed(3)
    x = (torch.randn((n, hw, hw, cin), generator=g, device='cuda') * 0.5).to(torch.bfloat16)      # 2.42 GB > 2 GiB
    assert x.numel() * 2 > 2 ** 31
    w = (torch.randn((cout, 1, 1, cin), generator=g, device='cuda') * (2.0 / cin) ** 0.5).to(torch.bfloat16)
    out = torch.empty((n, hw, hw, cout), dtype=torch.bfloat16, device='cuda')
    stats = torch.zeros(8 * cout * 2, dtype=torch.float64, device='cuda')
    ops.conv_igemm(x, w, out, stride=1, pad=0, stats=stats)
    ref = x.view(-1, cin).float() @ w.view(cout, cin).float().t()
    assert torch.allclose(out.view(-1, cout).float(), ref, rtol=2e-2, atol=2e-2)
    assert out[-1].float().abs().max().item() > 0        # the last image range was written
    s = stats.view(8, cout, 2).sum(0)
    assert torch.allclose(s[:, 0], ref.double().sum(0), rtol=1e-3, atol=1e-1) and torch.allclose(s[:, 1], (ref.double() ** 2).sum(0), rtol=1e-3)
    # weight gradient dW[a][b] = sum_m dy[m][a] * x[m][b]: q (= x) is the operand beyond 2 GiB
    dy = (torch.randn((n, hw, hw, cout), generator=g, device='cuda') * 0.05).to(torch.bfloat16)
    dw = torch.zeros((cout, 1, 1, cin), dtype=torch.float32, device='cuda')
    ops.conv_wgrad(dy, x, dw, 1, 1, stride=1, pad=0)
    dref = dy.view(-1, cout).float().t() @ x.view(-1, cin).float()
    assert (dw.view(cout, cin) - dref).abs().max().item() <= 2e-3 * dref.abs().max().item()
    # ... also through a grouped launch (the table gets one problem per image range)
    dw2 = torch.zeros_like(dw)
    ops.conv_wgrad_group([(dy, x, dw2, 1, 1, 1, 0)], 128, 128)
    assert (dw2.view(cout, cin) - dref).abs().max().item() <= 2e-3 * dref.abs().max().item()


def _bottleneck_reference(x, w1, w2, w3, co, dtype):
    """torch-CPU fp32 reference of the eval-mode identity Bottleneck with the kernel's storage points: both intermediates are
    rounded to the 16-bit dtype (as the three-launch path stores them in HBM), accumulation in fp32"""
    def store(t):
        return t.to(dtype).float()
    s1, b1, s2, b2, s3, b3 = [c.view(1, -1, 1, 1) for c in co]
    y1 = store(torch.relu(F.conv2d(x, w1) * s1 + b1))
    y2 = store(torch.relu(F.conv2d(y1, w2, padding=1) * s2 + b2))
    return torch.relu(F.conv2d(y2, w3) * s3 + b3 + x)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cmid,n,h,w,cfg,pad_ld', [(64, 3, 16, 32, 0, 0), (64, 1, 8, 16, 0, 64), (64, 2, 12, 16, 4, 0), (64, 1, 16, 16, 4, 0), (128, 2, 16, 16, 0, 0), (128, 1, 24, 32, 0, 256),
                                                   (256, 2, 8, 16, 2, 0), (256, 2, 8, 16, 4, 0), (256, 1, 16, 32, 4, 128), (256, 1, 6, 16, 2, 0)])
def test_fused_bottleneck_matches_the_three_convolutions(dtype, cmid, n, h, w, cfg, pad_ld):
    """msc_bottleneck_fused (conv1x1-bn-relu, conv3x3-bn-relu, conv1x1-bn, + x, relu in one launch, intermediates in LDS) against
    the composition of the three convolutions, for every kernel instantiation (Cmid 64: patch rows 8 and 4, 128: 8, 256: 2 and 4), several
    patches per image in both directions (halo rows AND columns cross patch borders), image borders (zero padding of the 3x3
    applies to conv1's OUTPUT: not ReLU(shift)), and input / output as channel slices of wider buffers"""
    import ctypes as C
    from mapping_challenge_amd import _lib
    lib = _lib.load()
    c4 = 4 * cmid
    x = rnd((n, c4, h, w), dtype, 1)
    w1 = rnd((cmid, c4, 1, 1), dtype, 2, (2.0 / c4) ** 0.5)
    w2 = rnd((cmid, cmid, 3, 3), dtype, 3, (2.0 / (9 * cmid)) ** 0.5)
    w3 = rnd((c4, cmid, 1, 1), dtype, 4, (2.0 / cmid) ** 0.5)
    g = torch.Generator().manual_seed(5)
    co = [torch.rand(cmid, generator=g) + 0.5, torch.randn(cmid, generator=g) * 0.2 + 0.1,       # positive shifts: a wrong padding value shows
          torch.rand(cmid, generator=g) + 0.5, torch.randn(cmid, generator=g) * 0.2 + 0.1,
          torch.rand(c4, generator=g) * 0.5 + 0.25, torch.randn(c4, generator=g) * 0.2]
    ref = _bottleneck_reference(x, w1, w2, w3, co, dtype)
    ld = c4 + pad_ld
    xb = torch.full((n, h, w, ld), 7.0, dtype=dtype, device='cuda')
    ob = torch.full((n, h, w, ld), -3.0, dtype=dtype, device='cuda')
    xb[..., pad_ld:] = nhwc(x, dtype)
    dt = {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}[dtype]
    wk = [t.permute(0, 2, 3, 1).contiguous().to(dtype).cuda() for t in (w1, w2, w3)]
    wpk = torch.empty(int(lib.msc_bottleneck_pack_bytes(cmid)), dtype=torch.uint8, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.msc_bottleneck_pack(wk[0].data_ptr(), wk[1].data_ptr(), wk[2].data_ptr(), wpk.data_ptr(), cmid, dt, stream), 'pack')
    cod = [c.cuda() for c in co]
    d = _lib.BneckDesc()
    d.x, d.out, d.wpk = xb.data_ptr() + pad_ld * 2, ob.data_ptr() + pad_ld * 2, wpk.data_ptr()
    d.scale1, d.shift1, d.scale2, d.shift2, d.scale3, d.shift3 = [c.data_ptr() for c in cod]
    d.x_ld = d.out_ld = ld
    d.dtype, d.N, d.H, d.W, d.Cmid, d.cfg = dt, n, h, w, cmid, cfg
    assert lib.msc_bottleneck_ok(C.byref(d)) == 1
    _lib.check(lib.msc_bottleneck_fused(C.byref(d), stream), 'msc_bottleneck_fused')
    torch.cuda.synchronize()
    got = to_nchw(ob[..., pad_ld:])
    assert torch.allclose(got, ref, **tol(dtype)), (got - ref).abs().max().item()
    if pad_ld:
        assert (ob[..., :pad_ld].float() == -3.0).all()          # nothing written outside the channel slice
    # shapes the kernel does not take are refused, not mis-run
    d.W = w + 8
    assert lib.msc_bottleneck_ok(C.byref(d)) == 0 and lib.msc_bottleneck_fused(C.byref(d), stream) != 0
    d.W, d.Cmid = w, 512
    assert lib.msc_bottleneck_ok(C.byref(d)) == 0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_dec0_conv_with_fused_final_1x1_and_softmax(dtype):
    """eval: ConvRelu(32, 32) + Conv2d(32, 2, 1) + channel softmax in one launch (msc_conv_desc.final_*; src/unet_models.py:401-403,
    src/models.py:88-92) against conv -> ReLU -> 16-bit rounding -> 1x1 -> softmax; `out` is left untouched with final_skip_store"""
    import ctypes as C
    from mapping_challenge_amd import _lib
    lib = _lib.load()
    n, hw = 2, 48
    x = rnd((n, 32, hw, hw), dtype, 1)
    w = rnd((32, 32, 3, 3), dtype, 2, (2.0 / 288) ** 0.5)
    bias = torch.randn(32) * 0.1
    fw, fb = torch.randn(2, 32) * 0.3, torch.randn(2) * 0.1
    act = torch.relu(F.conv2d(x, w, bias, padding=1)).to(dtype).float()
    logits = torch.einsum('nchw,kc->nkhw', act, fw) + fb.view(1, 2, 1, 1)
    xd, wk = nhwc(x, dtype), w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    out = torch.full((n, hw, hw, 32), 5.0, dtype=dtype, device='cuda')
    lg = torch.empty((n, 2, hw, hw), dtype=torch.float32, device='cuda')
    pr = torch.empty_like(lg)
    bd, fwd_, fbd = bias.cuda(), fw.contiguous().cuda(), fb.cuda()
    d = _lib.ConvDesc()
    d.in_, d.wt, d.out, d.shift = xd.data_ptr(), wk.data_ptr(), out.data_ptr(), bd.data_ptr()
    d.in_ld = d.out_ld = 32
    d.dtype, d.mode = {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}[dtype], 0
    d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = n, hw, hw, 32, hw, hw, 32
    d.KH, d.KW, d.stride, d.pad, d.relu = 3, 3, 1, 1, 1
    d.final_w, d.final_b, d.final_logits, d.final_probs, d.final_skip_store = fwd_.data_ptr(), fbd.data_ptr(), lg.data_ptr(), pr.data_ptr(), 1
    assert lib.msc_conv_cfg_ok(C.byref(d), _lib.CFG_HALO) == 1 and lib.msc_conv_cfg_ok(C.byref(d), 10) == 0
    _lib.check(lib.msc_conv_igemm(C.byref(d), torch.cuda.current_stream().cuda_stream), 'msc_conv_igemm')
    torch.cuda.synchronize()
    assert (lg.cpu() - logits).abs().max().item() < 2e-2 * max(1.0, logits.abs().max().item())
    assert (pr.cpu() - torch.softmax(lg.cpu(), 1)).abs().max().item() < 1e-6 and (pr.sum(1) - 1).abs().max().item() < 1e-6
    assert (out.float() == 5.0).all()                    # final_skip_store: dec0's activation is not written
    d.final_skip_store = 0
    _lib.check(lib.msc_conv_igemm(C.byref(d), torch.cuda.current_stream().cuda_stream), 'msc_conv_igemm')
    assert torch.allclose(to_nchw(out), act, **tol(dtype))
    # the logits are computed from the ROUNDED activation: comparable with the two-launch path up to the order of 32 fp32 additions
    lg2 = torch.empty_like(lg)
    _lib.call('msc_final_fwd', out.data_ptr(), 32, fwd_.data_ptr(), fbd.data_ptr(), lg2.data_ptr(), None, d.dtype, n, hw, hw, 32,
              torch.cuda.current_stream().cuda_stream)
    assert (lg - lg2).abs().max().item() < 1e-5 * max(1.0, lg2.abs().max().item())


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cin,cout,hw,n,split', [(512, 128, 4, 2, 4), (256, 256, 8, 1, 3), (192, 64, 6, 3, 16), (64, 128, 4, 1, 2)])
def test_split_k_convolution(dtype, cin, cout, hw, n, split):
    """msc_conv_desc.splitk: the reduction of a 3x3 convolution on a tiny map in `split` slices (partial tiles stored into the slice's plane of an
    fp32 workspace, a finishing pass that adds the planes and applies the epilogue) equals the single-pass launch and torch; slices that start inside a filter tap,
    a last slice shorter than the others (or empty), residual + scale / shift + ReLU in the finishing pass"""
    import ctypes as C
    from mapping_challenge_amd import _lib
    lib = _lib.load()
    x = rnd((n, cin, hw, hw), dtype, 1)
    w = rnd((cout, cin, 3, 3), dtype, 2, (2.0 / (9 * cin)) ** 0.5)
    res = rnd((n, cout, hw, hw), dtype, 3)
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1
    ref = torch.relu(F.conv2d(x, w, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    xd, wk, rd = nhwc(x, dtype), w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda(), nhwc(res, dtype)
    out = torch.empty((n, hw, hw, cout), dtype=dtype, device='cuda')
    ws = torch.full((split * n * hw * hw * cout,), float('nan'), dtype=torch.float32, device='cuda')      # scratch: every plane is written before it is read
    sc, sh = scale.cuda(), shift.cuda()
    d = _lib.ConvDesc()
    d.in_, d.wt, d.out, d.res, d.scale, d.shift = xd.data_ptr(), wk.data_ptr(), out.data_ptr(), rd.data_ptr(), sc.data_ptr(), sh.data_ptr()
    d.in_ld, d.out_ld, d.res_ld = cin, cout, cout
    d.dtype, d.mode = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}[dtype], 0
    d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = n, hw, hw, cin, hw, hw, cout
    d.KH, d.KW, d.stride, d.pad, d.relu = 3, 3, 1, 1, 1
    d.splitk, d.splitk_ws = split, ws.data_ptr()
    stream = torch.cuda.current_stream().cuda_stream
    cfgs = [c for c in range(1, lib.msc_conv_num_cfgs() + 1) if lib.msc_conv_cfg_ok(C.byref(d), c)]
    assert cfgs and not any(42 <= c <= 46 or 51 <= c <= 56 or c in (27, 28) for c in cfgs)       # the halo-tile kernels do not split
    for cfg in [0] + cfgs[:6]:
        d.cfg = cfg
        out.fill_(7.0)
        _lib.check(lib.msc_conv_igemm(C.byref(d), stream), 'msc_conv_igemm')
        torch.cuda.synchronize()
        assert torch.allclose(to_nchw(out), ref, **tol(dtype)), (cfg, (to_nchw(out) - ref).abs().max().item())
    # statistics / transposed mode are refused with split-K
    d.mode = 1
    assert lib.msc_conv_igemm(C.byref(d), stream) != 0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cin,cout', [(64, 256), (64, 64), (128, 512), (128, 128), (256, 128), (256, 64), (512, 128), (256, 1024)])
def test_streaming_1x1_kernel(dtype, cin, cout):
    """conv1x1_stream_kernel (configuration 57): persistent blocks with the weights as register fragments and the pixel tiles streamed
    through two LDS buffers -- more tiles than blocks (the carried loop), channel slices of wider buffers on both sides, folded-BN
    coefficients + residual + ReLU, the three statistics epilogues (sums carried across the tiles of a block), against torch"""
    import ctypes as C
    from mapping_challenge_amd import _lib
    lib = _lib.load()
    n, hw = 5, 128                                    # 81 920 pixels: 2.5 tiles of 64 per block at 512 blocks
    if cin * cout >= 256 * 512:
        n = 3
    x = rnd((n, cin, hw, hw), dtype, 1)
    w = rnd((cout, cin, 1, 1), dtype, 2, (2.0 / cin) ** 0.5)
    res = rnd((n, cout, hw, hw), dtype, 3)
    y = rnd((n, cout, hw, hw), dtype, 4)
    scale, shift = torch.rand(cout, generator=torch.Generator().manual_seed(5)) + 0.5, rnd((cout,), torch.float32, 6) * 0.3
    raw = F.conv2d(x, w)
    wide_in = torch.zeros((n, hw, hw, cin + 64), dtype=dtype, device='cuda')
    wide_in[..., 32:32 + cin] = nhwc(x, dtype)
    xd = wide_in[..., 32:32 + cin]
    wide_out = torch.zeros((n, hw, hw, cout + 32), dtype=dtype, device='cuda')
    od = wide_out[..., 16:16 + cout]
    wk = w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    rd, yd = nhwc(res, dtype), nhwc(y, dtype)
    sc, sh = scale.cuda(), shift.cuda()
    stream = torch.cuda.current_stream().cuda_stream
    dt = {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}[dtype]

    def desc(cfg):
        d = _lib.ConvDesc()
        d.in_, d.wt, d.out = xd.data_ptr(), wk.data_ptr(), od.data_ptr()
        d.in_ld, d.out_ld, d.dtype, d.mode = cin + 64, cout + 32, dt, 0
        d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.cfg = n, hw, hw, cin, hw, hw, cout, 1, 1, 1, 0, cfg
        return d

    tried = 0
    t2 = dict(rtol=2e-3, atol=5e-1) if dtype == torch.float16 else dict(rtol=2e-2, atol=2.0)
    for cfg in (_lib.CFG_STREAM,):
        if not lib.msc_conv_cfg_ok(C.byref(desc(cfg)), cfg):
            continue
        tried += 1
        # (a) raw output + forward statistics
        d = desc(cfg)
        stats = torch.zeros((_lib.BN_SLOTS, cout, 2), dtype=torch.float64, device='cuda')
        d.stats = stats.data_ptr()
        wide_out.fill_(3.0)
        _lib.check(lib.msc_conv_igemm(C.byref(d), stream), 'conv')
        assert torch.allclose(to_nchw(od), raw, **tol(dtype)), cfg
        assert (wide_out[..., :16] == 3).all() and (wide_out[..., 16 + cout:] == 3).all()
        s = stats.sum(0).float().cpu()
        assert torch.allclose(s[:, 0], raw.sum((0, 2, 3)), **t2) and torch.allclose(s[:, 1], (raw * raw).sum((0, 2, 3)), rtol=2e-2, atol=2.0), cfg
        # (b) folded BatchNorm + residual + ReLU
        d = desc(cfg)
        d.scale, d.shift, d.res, d.res_ld, d.relu = sc.data_ptr(), sh.data_ptr(), rd.data_ptr(), cout, 1
        _lib.check(lib.msc_conv_igemm(C.byref(d), stream), 'conv')
        ref = torch.relu(raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
        assert torch.allclose(to_nchw(od), ref, **tol(dtype)), cfg
        # (c) BatchNorm-backward sums against y under the ReLU mask of (scale, shift)
        d = desc(cfg)
        stats.zero_()
        d.stats, d.stats_kind, d.stats_y, d.stats_y_ld, d.scale, d.shift = stats.data_ptr(), 1, yd.data_ptr(), cout, sc.data_ptr(), sh.data_ptr()
        _lib.check(lib.msc_conv_igemm(C.byref(d), stream), 'conv')
        m = ((y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0).float()
        s = stats.sum(0).float().cpu()
        assert torch.allclose(to_nchw(od), raw, **tol(dtype)), cfg
        assert torch.allclose(s[:, 0], (raw * m).sum((0, 2, 3)), **t2) and torch.allclose(s[:, 1], (raw * m * y).sum((0, 2, 3)), **t2), cfg
        # (d) ReLU backward + bias-gradient sums
        d = desc(cfg)
        stats.zero_()
        d.stats, d.stats_kind, d.stats_y, d.stats_y_ld = stats.data_ptr(), 2, yd.data_ptr(), cout
        _lib.check(lib.msc_conv_igemm(C.byref(d), stream), 'conv')
        assert torch.allclose(to_nchw(od), raw * (y > 0), **tol(dtype)), cfg
        assert torch.allclose(stats.sum(0)[:, 0].float().cpu(), (raw * (y > 0)).sum((0, 2, 3)), **t2), cfg
    assert tried >= 1
    # not for other filter sizes / strides / ragged pixel counts
    d = desc(_lib.CFG_STREAM)
    d.stride, d.Ho, d.Wo = 2, hw // 2, hw // 2
    assert not lib.msc_conv_cfg_ok(C.byref(d), _lib.CFG_STREAM)
    d = desc(_lib.CFG_STREAM)
    d.Hi = d.Ho = 3
    d.Wi = d.Wo = 5
    assert not lib.msc_conv_cfg_ok(C.byref(d), _lib.CFG_STREAM)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,hw', [(3, 64), (2, 256), (12, 256)])      # (12, 256): 1536 patches = three per persistent block (all three halo buffers)
def test_stem_halo_kernel(dtype, n, hw):
    """stem7_halo_kernel (configuration 58): the 7x7 / stride 2 / pad 3 stem on the prepared 4-channel input equals torch's conv and the
    implicit-GEMM configurations of the same descriptor: raw output + BatchNorm statistics (training), folded coefficients + ReLU (eval)"""
    import ctypes as C
    from mapping_challenge_amd import _lib
    lib = _lib.load()
    x = rnd((n, 3, hw, hw), dtype, 1)
    w = rnd((64, 3, 7, 7), dtype, 2, (2.0 / 147) ** 0.5)
    raw = F.conv2d(x, w, stride=2, padding=3)
    ho = hw // 2
    dt = {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}[dtype]
    stream = torch.cuda.current_stream().cuda_stream
    xd, wd = x.cuda().contiguous(), w.cuda().contiguous()
    xp = torch.empty((n, hw + 6, hw + 8, 4), dtype=dtype, device='cuda')
    wp = torch.empty((64, 7, 32), dtype=dtype, device='cuda')
    _lib.check(lib.msc_stem_prepare(xd.data_ptr(), xp.data_ptr(), dt, n, hw, hw, stream), 'prepare')
    _lib.check(lib.msc_stem_pack(wd.data_ptr(), wp.data_ptr(), dt, 64, stream), 'pack')
    out = torch.empty((n, ho, ho, 64), dtype=dtype, device='cuda')
    scale, shift = torch.rand(64, generator=torch.Generator().manual_seed(3)) + 0.5, rnd((64,), torch.float32, 4) * 0.3
    sc, sh = scale.cuda(), shift.cuda()

    def desc(cfg):
        d = _lib.ConvDesc()
        d.in_, d.wt, d.out = xp.data_ptr(), wp.data_ptr(), out.data_ptr()
        d.in_ld, d.out_ld, d.dtype, d.mode = 4, 64, dt, 0
        d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.cfg = n, hw + 6, hw + 8, 32, ho, ho, 64, 7, 1, 2, 0, cfg
        return d

    assert lib.msc_conv_cfg_ok(C.byref(desc(58)), 58)
    cfgs = [c for c in range(1, lib.msc_conv_num_cfgs() + 1) if lib.msc_conv_cfg_ok(C.byref(desc(c)), c)]
    assert 58 in cfgs and len(cfgs) > 1
    for cfg in (58, cfgs[0]):
        d = desc(cfg)
        stats = torch.zeros((_lib.BN_SLOTS, 64, 2), dtype=torch.float64, device='cuda')
        d.stats = stats.data_ptr()
        out.fill_(5.0)
        _lib.check(lib.msc_conv_igemm(C.byref(d), stream), 'conv')
        assert torch.allclose(to_nchw(out), raw, **tol(dtype)), cfg
        s = stats.sum(0).float().cpu()
        assert torch.allclose(s[:, 0], raw.sum((0, 2, 3)), rtol=2e-2, atol=1.0) and torch.allclose(s[:, 1], (raw * raw).sum((0, 2, 3)), rtol=2e-2, atol=1.0), cfg
        d = desc(cfg)
        d.scale, d.shift, d.relu = sc.data_ptr(), sh.data_ptr(), 1
        _lib.check(lib.msc_conv_igemm(C.byref(d), stream), 'conv')
        assert torch.allclose(to_nchw(out), torch.relu(raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)), **tol(dtype)), cfg
    # other geometries do not qualify
    d = desc(58)
    d.in_ld = 8
    assert not lib.msc_conv_cfg_ok(C.byref(d), 58)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('bits', [False, True])
@pytest.mark.parametrize('cin,cout,k,hw,n,with_res', [(256, 1024, 1, 16, 4, True), (64, 64, 3, 20, 2, True), (128, 64, 1, 16, 3, False), (64, 256, 1, 32, 2, True)])
def test_conv_epilogue_batchnorm_backward_sums_of_a_residual_join(dtype, cin, cout, k, hw, n, with_res, bits):
    """stats_kind 1 with stats_z (ABI v6): the data-gradient conv that accumulates the last addend of a residual join's gradient stores
    out = acc + res and reduces (sum dh, sum dh*y), dh = out * [z > 0] with z the block's (post-add, post-ReLU) output -- what
    msc_bn_bwd_reduce (mask mode 1) would compute from the stored tensors; in place (res == out), every valid configuration.
    bits (ABI v7): stats_z is the byte mask msc_bn_apply leaves (one byte per 16-byte channel vector) instead of the activation"""
    import ctypes as C
    from mapping_challenge_amd import _lib
    import hip_ops as ops
    lib = _lib.load()
    pad = k // 2
    x = rnd((n, cin, hw, hw), dtype, 1)
    w = rnd((cout, cin, k, k), dtype, 2, (2.0 / (cin * k * k)) ** 0.5)
    y = rnd((n, cout, hw, hw), dtype, 3)
    z = torch.relu(rnd((n, cout, hw, hw), dtype, 4))
    g0 = rnd((n, cout, hw, hw), dtype, 5)
    tot = F.conv2d(x, w, padding=pad) + (g0 if with_res else 0)
    m = (z > 0).float()
    e1, e2 = (tot * m).sum((0, 2, 3)), (tot * m * y).sum((0, 2, 3))
    xd, yd, zd = nhwc(x, dtype), nhwc(y, dtype), nhwc(z, dtype)
    wk = w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    out = torch.empty((n, hw, hw, cout), dtype=dtype, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream
    tried = 0
    for c in [0] + ops.conv_valid_cfgs(xd, wk, out, 1, pad):
        d = ops.ConvDesc()
        d.in_, d.wt, d.out = xd.data_ptr(), wk.data_ptr(), out.data_ptr()
        d.in_ld, d.out_ld, d.dtype, d.mode = cin, cout, ops._dt(xd), 0
        d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.cfg = n, hw, hw, cin, hw, hw, cout, k, k, 1, pad, c
        d.stats_kind, d.stats_y, d.stats_y_ld, d.stats_z, d.stats_z_ld = 1, yd.data_ptr(), cout, zd.data_ptr(), cout
        if bits:
            ce = 16 // zd.element_size()
            zb = ((zd.float().view(n, hw, hw, cout // ce, ce) > 0).to(torch.int32) << torch.arange(ce, device='cuda', dtype=torch.int32)).sum(-1).to(torch.uint8)
            d.stats_z, d.stats_z_ld, d.stats_z_bits = zb.data_ptr(), cout // ce, 1
        stats = torch.zeros((_lib.BN_SLOTS, cout, 2), dtype=torch.float64, device='cuda')
        d.stats = stats.data_ptr()
        if c and not lib.msc_conv_cfg_ok(C.byref(d), c):
            continue
        tried += 1
        out.copy_(nhwc(g0, dtype))
        if with_res:
            d.res, d.res_ld = out.data_ptr(), cout          # accumulate in place, as the engine does
        _lib.check(lib.msc_conv_igemm(C.byref(d), stream), 'conv')
        assert torch.allclose(to_nchw(out), tot, **tol(dtype)), c
        s = stats.sum(0).float().cpu()
        t = dict(rtol=2e-3, atol=5e-2) if dtype == torch.float32 else dict(rtol=2e-2, atol=0.7)
        assert torch.allclose(s[:, 0], e1, **t) and torch.allclose(s[:, 1], e2, **t), c
    assert tried >= 2
    # a residual without stats_z, or stats_z together with coefficients, is refused
    d.stats_z = None
    d.res, d.res_ld = out.data_ptr(), cout
    assert lib.msc_conv_igemm(C.byref(d), stream) != 0


def test_stream_ordered_fill_and_copy_are_exact_at_any_alignment_and_size():
    """msc_memset_zero / msc_copy are kernels since round 5 (a captured step must hold kernel nodes only): every byte of the range and
    no byte outside it, for unaligned starts, sizes around the 16-byte vector width and a size that wraps the grid; the copy also with source and destination
    in different phases of 16 bytes"""
    from mapping_challenge_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(5)
    for n in (1, 15, 16, 17, 31, 255, 1000, 4097, (1 << 22) + 3, 4096 * 256 * 16 * 2 + 48):
        for off in (0, 1, 4, 8, 15):
            buf = torch.randint(1, 256, (n + 64,), dtype=torch.uint8, generator=gen).cuda()
            keep = buf.clone()
            _lib.check(lib.msc_memset_zero(buf.data_ptr() + off, n, st), 'msc_memset_zero')
            torch.cuda.synchronize()
            assert int(buf[off:off + n].max()) == 0, (n, off)
            assert torch.equal(buf[:off], keep[:off]) and torch.equal(buf[off + n:], keep[off + n:]), (n, off)
            for soff in (off, (off + 3) % 16):
                src = torch.randint(0, 256, (n + 64,), dtype=torch.uint8, generator=gen).cuda()
                dst = keep.clone()
                _lib.check(lib.msc_copy(dst.data_ptr() + off, src.data_ptr() + soff, n, st), 'msc_copy')
                torch.cuda.synchronize()
                assert torch.equal(dst[off:off + n], src[soff:soff + n]), (n, off, soff)
                assert torch.equal(dst[:off], keep[:off]) and torch.equal(dst[off + n:], keep[off + n:]), (n, off, soff)
    assert lib.msc_memset_zero(None, 16, st) != 0 and lib.msc_copy(None, None, 16, st) != 0       # null pointers are refused
