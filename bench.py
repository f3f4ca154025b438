"""Benchmark of the U-Net hot path on MI355X (contract: see the task statement / DESIGN.md section 5).

    python bench.py --gpus N --steps K --warmup W [--workload train|infer|tta|post|annot] [--encoder 101] [--dtype bf16|fp16|fp32]
                    [--size 256|320|512] [--batch B]

One "step" = one pass of the hot path over one synthetic batch resident in HBM:
  train (default, BASELINE.json metric / configs[2]): ResNet101-U-Net forward + mixed weighted-CE/Dice loss + backward +
        Adam(+L2), batch 32 per GPU, 300x300 tiles resized to the 256x256 network input (the reference's default
        loader_mode, neptune.yaml:23,27-28), bf16 compute / fp32 accumulate & master weights
  infer (configs[1]): ResNet34-U-Net eval forward + fused softmax, batch 32 (--encoder 101: the north-star forward line)
  tta   (configs[4]): ResNet152-U-Net fp16, batch 64, 512x512 tiles, test-time augmentation x4 (identity, two flips, both
        -- the reference's elif chain, src/loaders.py:478-481), aggregated on the device
  post  (configs[3], the chain alone): resize 256->300, threshold, 4-connected labelling, 2x2 label dilation, scoring, batch 64 masks
  e2e   (configs[3] as written): ResNet101-U-Net inference -> post-processing on the device -> COCO annotations, with the variants
        + watershed (extension), + dense CRF, + both; value = the FULL chain (morphology + watershed + dense CRF)
--size: network input edge; 256 = the reference's default loader (300x300 tiles resized, neptune.yaml:23,27-28), 320 = its
crop_and_pad loader (tiles replicate-padded by 10 px, neptune.yaml:77-79).
N > 1: one rank per GPU, batch sharded (weak scaling), loss sums and gradients all-reduced over RCCL.  Started without a
torchrun environment, `--gpus N` re-executes itself under torch.distributed.run.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# forward GFLOP/img (2 x MACs of Conv2d + ConvTranspose2d), train = 3x
PEAK_BF16, PEAK_F32, PEAK_HBM = 2.5e15, 157.3e12, 8.0e12
ARCH = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
        'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}


def conv_flops(d):
    """algorithmic FLOPs (2 x MACs) of one msc_conv_igemm launch"""
    macs = d.N * d.Ho * d.Wo * d.Cout * d.Cin * d.KH * d.KW
    if d.mode == 1:
        macs /= 4.0          # each output parity phase uses a quarter of the taps
    return 2.0 * macs


def bneck_flops(d):
    """algorithmic FLOPs of one fused Bottleneck launch: its three convolutions (1x1 4C->C, 3x3 C->C, 1x1 C->4C), halo recompute not counted"""
    return 2.0 * d.N * d.H * d.W * 17.0 * d.Cmid * d.Cmid


def wgrad_flops(d):
    return 2.0 * d.N * d.Hp * d.Wp * d.A * d.B * d.KH * d.KW


def dump_launches(launches, stream, path, repeats=3):
    """per-launch HIP-event timing of the conv / wgrad launches (shape, us, TFLOP/s) -> JSON, for tuning"""
    rows = []
    for fn, args in launches:
        name = fn.__name__
        if name not in ('msc_conv_igemm', 'msc_conv_wgrad', 'msc_bottleneck_fused'):
            continue
        best = 1e30
        for _ in range(repeats):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            assert fn(*args, stream) == 0
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        d = args[0]._obj
        if name == 'msc_bottleneck_fused':
            rows.append({'k': 'bottleneck', 'N': d.N, 'H': d.H, 'W': d.W, 'Cmid': d.Cmid, 'us': 1e3 * best, 'tflops': bneck_flops(d) / (best * 1e-3) / 1e12})
        elif name == 'msc_conv_igemm':
            rows.append({'k': 'conv', 'mode': d.mode, 'flip': d.flip, 'N': d.N, 'Hi': d.Hi, 'Wi': d.Wi, 'Cin': d.Cin, 'Ho': d.Ho, 'Wo': d.Wo,
                         'Cout': d.Cout, 'KH': d.KH, 'KW': d.KW, 'stride': d.stride, 'stats': bool(d.stats), 'res': bool(d.res),
                         'us': 1e3 * best, 'tflops': conv_flops(d) / (best * 1e-3) / 1e12})
        else:
            rows.append({'k': 'wgrad', 'N': d.N, 'Hp': d.Hp, 'Wp': d.Wp, 'A': d.A, 'Hq': d.Hq, 'Wq': d.Wq, 'B': d.B, 'KH': d.KH, 'KW': d.KW,
                         'stride': d.stride, 'us': 1e3 * best, 'tflops': wgrad_flops(d) / (best * 1e-3) / 1e12})
    with open(path, 'w') as f:
        json.dump(rows, f)


def family_times(launches, stream, repeats=2):
    """HIP-event time per kernel family over a launch list, on the stream the kernels run on."""
    fam = {}
    for _ in range(repeats):
        evs = []
        for fn, args in launches:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = fn(*args, stream)
            assert rc == 0, fn.__name__
            b.record()
            evs.append((fn.__name__, args, a, b))
        torch.cuda.synchronize()
        for name, args, a, b in evs:
            f = fam.setdefault(name, {'ms': 0.0, 'launches': 0, 'flops': 0.0})
            f['ms'] += a.elapsed_time(b) / repeats
            f['launches'] += 1.0 / repeats
            if name == 'msc_conv_igemm':
                f['flops'] += conv_flops(args[0]._obj) / repeats
            elif name == 'msc_bottleneck_fused':
                f['flops'] += bneck_flops(args[0]._obj) / repeats
            elif name == 'msc_conv_wgrad':
                f['flops'] += wgrad_flops(args[0]._obj) / repeats
            elif name == 'msc_wgrad_group_run':
                f['flops'] += sum(wgrad_flops(d) for d in args[0].descs) / repeats
                f['launches'] += (len(args[0].descs) - 1.0) / repeats      # counted as layers, not kernel launches
            elif name in BN_TENSORS:
                f['bytes'] = f.get('bytes', 0.0) + BN_TENSORS[name](args) / repeats
    return fam


def _es(dtype):
    from mapping_challenge_amd._lib import BF16
    return 2 if dtype == BF16 else 4


# algorithmic HBM bytes of the BatchNorm elementwise launches: tensors of pixels*C elements read or written
# (argument positions: include/msc.h)
BN_TENSORS = {
    'msc_bn_apply': lambda a: a[22] * a[23] * _es(a[21]) * (2 + (1 if a[2] else 0)),
    'msc_bn_bwd_reduce': lambda a: a[11] * a[12] * _es(a[10]) * (2 + (1 if a[6] == 1 else 0)),
    'msc_bn_apply_pool': lambda a: a[17] * a[18] * a[19] * a[20] * _es(a[16]) * 5,
    'msc_bn_pool_bwd_reduce': lambda a: a[8] * a[9] * a[10] * a[11] * _es(a[7]) * 5,
    'msc_bn_pool_bwd_apply': lambda a: a[14] * a[15] * a[16] * a[17] * _es(a[13]) * 9,
    'msc_bn_bwd_apply': lambda a: a[25] * a[26] * _es(a[24]) * (3 + (1 if a[6] == 1 else (1.0 / 16 if a[6] == 3 else 0)) + ((1 + (1 if a[20] else 0)) if a[18] else 0) + (1 if a[21] else 0)),
}


def cpu_threads():
    """torch-CPU intra-op threads for the baseline: all host cores up to 32 (beyond that the small convolutions of
    this network get slower, not faster, from oversubscription); reported as `cores`."""
    return max(1, min(32, os.cpu_count() or 1))


def cpu_baseline_train(encoder, hw, budget_s=25.0):
    """The oracle (torch-CPU fp32 restatement of the reference modules + losses + torch Adam) timed on the host
    cores on a bounded sample of the same workload: batch 2, as many steps as fit the budget (>= 1)."""
    from oracle import losses_ref, unet_ref
    torch.set_num_threads(cpu_threads())
    net = unet_ref.UNetResNetRef(encoder)
    net.load_state_dict(unet_ref.seeded_state_dict(net))
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=5e-4, weight_decay=1e-4)
    n = 2
    x, t = unet_ref.synthetic_batch(n, hw, hw), losses_ref.synthetic_target(n, hw, hw)

    def step():
        opt.zero_grad()
        losses_ref.mixed_dice_ce(net(x), t).backward()
        opt.step()
    t0 = time.time()
    step()
    warm = time.time() - t0
    t0, k = time.time(), 0
    if warm > budget_s:           # one step already exceeds the budget: the (cold) step is the sample
        k, t0 = 1, t0 - warm
    while k < 1 or (time.time() - t0) < budget_s and k < 8:
        step()
        k += 1
    dt = time.time() - t0
    return {'value': n * k / dt, 'unit': 'img/s', 'cores': cpu_threads(), 'kind': 'port',
            'sample': 'oracle port UNetResNetRef(%d) (torch.equal to src.unet_models.UNetResNet, tests/test_oracle.py) fp32 torch-CPU train step (fwd+mixed loss+bwd+Adam), batch %d at %dx%d, %d steps'
                      % (encoder, n, hw, hw, k)}


def cpu_baseline_infer(encoder, hw, budget_s=20.0, n=4):
    from oracle import unet_ref
    torch.set_num_threads(cpu_threads())
    net = unet_ref.UNetResNetRef(encoder)
    net.load_state_dict(unet_ref.seeded_state_dict(net))
    net.eval()
    x = unet_ref.synthetic_batch(n, hw, hw)
    with torch.no_grad():
        net(x)
        t0, k = time.time(), 0
        while k < 1 or (time.time() - t0) < budget_s and k < 10:
            torch.softmax(net(x), 1)
            k += 1
    dt = time.time() - t0
    return {'value': n * k / dt, 'unit': 'img/s', 'cores': cpu_threads(), 'kind': 'port',
            'sample': 'oracle UNetResNetRef(%d) fp32 torch-CPU eval forward + softmax, batch %d at %dx%d, %d passes' % (encoder, n, hw, hw, k)}


def cpu_baseline_post(probs, target, dilate, budget_s=15.0):
    """oracle/post_ref.c: the chain restated in plain C, one thread (the reference runs its Python/scipy loop on one
    core too, src/utils.py:352-354; the C port is ~2.5x faster than that scipy path on the same core)."""
    from oracle import post_ref_c
    post_ref_c.load()
    t0, k = time.time(), 0
    while time.time() - t0 < budget_s and k < 4096:
        post_ref_c.postprocess(probs[k % len(probs)], target, dilate)
        k += 1
    dt = time.time() - t0
    return {'value': k / dt, 'unit': 'img/s', 'cores': 1, 'kind': 'port',
            'sample': 'oracle/post_ref.c (plain C, -O2, single thread) resize+threshold+label+dilate+score on %d masks' % k}


def cpu_baseline_annot(layers, budget_s=15.0):
    """the reference's loop (decompose -> encode -> toBbox per instance, src/utils.py:61-127) on the numpy restatement of
    the pycocotools arithmetic, one thread, image by image until the budget is spent"""
    from oracle import annot_ref
    t0, k = time.time(), 0
    while time.time() - t0 < budget_s and k < len(layers) // 2:
        pair = layers[2 * k:2 * k + 2]
        scores = [[1.0] * int(l.max()) for l in pair]
        annot_ref.create_annotations([k], [(pair, scores)], [100, 100], [1, 1])
        k += 1
    dt = time.time() - t0
    return {'value': k / dt, 'unit': 'img/s', 'cores': 1, 'kind': 'port',
            'sample': 'oracle/annot_ref.py (numpy restatement of src/utils.py:61-127 + maskApi.c) on %d images of 2 layers' % k}


def bench_e2e(args, world, dev, stream, timed):
    """BASELINE.json configs[3] as written: inference + full HIP post-processing (morphology + watershed + dense CRF) on 256x256
    masks, ending in the COCO annotations (src/pipelines.py:248-304 -> src/utils.py:76-115).  The network is first trained
    for 40 steps on inputs that carry their target (untimed), so that its masks are building-like blobs -- the cost of labelling,
    flooding and encoding depends on the number and shape of the instances, and random weights give 0.5-noise."""
    from mapping_challenge_amd import postprocessing as post, utils
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    from mapping_challenge_amd.unet_models import UNetResNet
    import synthetic_inputs as losses_ref
    unet_ref = losses_ref
    enc = args.encoder or 101
    batch = args.batch or 32
    hw = args.size
    net = UNetResNet(enc, 2, num_filters=32, dropout_2d=0.0, is_deconv=True, compute_dtype=args.dtype)
    net.load_state_dict(unet_ref.seeded_state_dict(net))
    net.flatten_parameters(dev)
    tgt = losses_ref.synthetic_target(min(batch, 8), hw, hw, seed=31 + world.rank)
    tgt = tgt.repeat((batch + tgt.shape[0] - 1) // tgt.shape[0], 1, 1, 1)[:batch].contiguous()
    x = (unet_ref.synthetic_batch(batch, hw, hw, seed=1234 + world.rank) * 0.5 + 2.0 * tgt[:, :1]).to(dev)
    net.train()
    step = TrainStep(net, LossSpec.mixed(ARCH), HipAdam(net, lr=5e-4, weight_decay=1e-4), use_graph=False)
    for _ in range(40):
        step(x, tgt.to(dev))
    del step
    net.eval()
    # the de-normalised RGB tiles the CRF compares colours on (src/postprocessing.py:203-204, src/utils.py:324-325)
    mean = torch.tensor(post.MEAN, device=dev).view(1, 3, 1, 1)
    std = torch.tensor(post.STD, device=dev).view(1, 3, 1, 1)
    rgb = ((x * std + mean) * 255.0).clamp(0, 255).permute(0, 2, 3, 1).contiguous().to(torch.uint8)
    # The tail (device chain -> annotations) has four host synchronisations, ~60 launches and the Python that builds the annotation
    # dicts: a cost per CALL that one network batch does not amortise (4.5 ms per call of 32 images, 0.5 ms of it GPU time).  The
    # chain is batched and device-resident, so it takes the probabilities of TB network batches per call (TB = 4: 128 images).
    TB = int(os.environ.get('MSC_E2E_TAIL_BATCHES', '4'))
    ids = list(range(batch * TB))
    cat_ids, layers = [None, 100], [1, 1]            # src/pipeline_config.py:17-18
    rgb_t = rgb.repeat(TB, 1, 1, 1)

    def tail(probs, ws=0, crf=False, as_list=False):
        # the product of the tail is the JSON document of submission.json (src/utils.py:105-110), written natively from the encoder's
        # table; as_list: the same annotations as Python dicts (what create_annotations returns when it does not save)
        fn = utils.annotations_from_probabilities if as_list else utils.annotations_json_from_probabilities
        return fn(ids, probs, cat_ids, layers, (300, 300), 0, 2, watershed_selem_size=ws, crf_images=rgb_t if crf else None)
    probs_t = torch.empty((batch * TB, 2, hw, hw), dtype=torch.float32, device=dev)

    def forward_all():
        for k in range(TB):
            probs_t[k * batch:(k + 1) * batch].copy_(net.predict_proba(x))
        return probs_t
    probs0 = forward_all().clone()
    fg = float((probs0[:, 1] > 0.5).float().mean().item())
    ann = tail(probs0, as_list=True)
    import json as _json
    assert _json.loads(tail(probs0)) == ann
    variants = {'plain': dict(ws=0, crf=False), 'watershed': dict(ws=5, crf=False), 'crf': dict(ws=0, crf=True), 'full': dict(ws=5, crf=True)}
    out = {}
    imgs = batch * TB * world.size * args.steps
    dt_net = timed(lambda: net.predict_proba(x))
    out['inference_only_img_s'] = batch * world.size * args.steps / dt_net
    out['images_per_tail_call'] = batch * TB
    dt_list = timed(lambda: tail(probs0, as_list=True))
    out['plain_as_python_dicts_post_only_img_s'] = imgs / dt_list      # the same tail building one dict per instance in Python
    from mapping_challenge_amd.pipelines import OverlappedAnnotator

    def overlapped(n_groups, kw):
        """n_groups complete steps through pipelines.OverlappedAnnotator: the network batches of step g+1 are enqueued on their own stream
        before the host drives the tail of step g on another (pipeline fill and drain are inside the timed region)"""
        ann_ = OverlappedAnnotator(net, cat_ids, layers, (300, 300), 0, 2, watershed_selem_size=kw['ws'])
        docs = list(ann_.annotate((ids, [x] * TB, rgb_t if kw['crf'] else None) for _ in range(n_groups)))
        return docs

    def timed_overlapped(kw):
        overlapped(max(args.warmup, 1), kw)
        world.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        docs = overlapped(args.steps, kw)
        torch.cuda.synchronize()
        world.barrier()
        assert len(docs) == args.steps
        return time.perf_counter() - t0
    assert overlapped(2, variants['plain'])[1] == tail(probs0)          # the two-stream pipeline returns the one-stream document
    sequential = os.environ.get('MSC_E2E_SEQUENTIAL') == '1'          # A/B: network and tail alternating on one stream (round 3)
    for name, kw in variants.items():
        dt_tail = timed(lambda: tail(probs0, **kw))                                  # the post-processing + annotation part alone
        dt_seq = timed(lambda: tail(forward_all(), **kw))                            # TB network batches + one tail call, on one stream
        dt_all = dt_seq if sequential else timed_overlapped(kw)
        out[name] = {'post_only_img_s': imgs / dt_tail, 'post_only_ms_per_img': 1e3 * dt_tail / (batch * TB * args.steps),
                     'end_to_end_img_s': imgs / dt_all, 'end_to_end_ms_per_step': 1e3 * dt_all / args.steps,
                     'end_to_end_one_stream_img_s': imgs / dt_seq,
                     'end_to_end_vs_slower_half': (imgs / dt_all) / min(imgs / dt_tail, out['inference_only_img_s'] * 1.0),
                     'post_not_slower_than_network': imgs / dt_tail >= out['inference_only_img_s']}
    # dense CRF alone: HIP events around the launches of one call on the launch stream
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    post.dense_crf_batch(probs0[:batch], rgb)
    torch.cuda.synchronize()
    a.record()
    for _ in range(5):
        post.dense_crf_batch(probs0[:batch], rgb)
    b.record()
    torch.cuda.synchronize()
    crf_ms = a.elapsed_time(b) / 5
    # algorithmic work of the exact windowed mean field (oracle/crf_ref.py): per pixel a normalisation pass and 5 iterations over the
    # 11x11 window; Gaussian tap = 2 FMA (4 flop), bilateral tap = 3 sub + 3 FMA + scale + exp + weight + 2 FMA (16 flop)
    taps = 121
    crf_flop = batch * hw * hw * (taps * (1 + 12) + 5 * taps * (4 + 16))
    full = out['full']
    prog = net._program(batch, hw, hw, False, dev)
    fam = family_times(list(prog.fwd), stream)
    conv = {k: fam.get('msc_conv_igemm', {}).get(k, 0.0) + fam.get('msc_bottleneck_fused', {}).get(k, 0.0) for k in ('ms', 'launches', 'flops')}
    ach = conv['flops'] / (conv['ms'] * 1e-3)
    res = {'metric': 'images/sec (inference + full HIP post-processing: morphology + watershed + dense CRF -> annotations) ResNet%d-U-Net 256x256' % enc,
           'unit': 'img/s', 'value': full['end_to_end_img_s'], 'ms_per_step': full['end_to_end_ms_per_step'],
           'config': {'workload': 'ResNet%d-U-Net eval forward (batch %d/GPU, %s, %s) -> dense CRF (5 iterations, sxy 1, srgb 50) -> resize 256->300, threshold, '
                                  'erosion-marker watershed (extension, k=5), 2x2 label dilation, scoring -> COCO RLE + bbox annotations; weights '
                                  'trained 40 steps on synthetic blobs (foreground %.2f, %d annotations per tail call of %d images in the plain variant); '
                                  'one step = %d network batches + one tail call; network (step g+1) and tail (step g) on two streams (pipelines.OverlappedAnnotator)'
                                  % (enc, batch, tile_text(hw), args.dtype, fg, len(ann), batch * TB, TB),
                      'global_batch': batch * world.size, 'parallelism': 'dp%d' % world.size, 'images_per_step': batch * TB, 'variants': out},
           'roofline': {'kernel': 'conv family of the forward (dominant: %.2f of %.2f ms per step are the network)' % (1e3 * TB * dt_net / args.steps, full['end_to_end_ms_per_step']),
                        'bound': 'mfma', 'achieved': ach / 1e12, 'peak': PEAK_BF16 / 1e12, 'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16, 'traffic': None,
                        'dense_crf': {'kernel': 'crf_norm / crf_iter (exact 11x11 windowed mean field, 5 iterations)', 'bound': 'valu-fp32',
                                      'ms_per_batch': crf_ms, 'us_per_img': 1e3 * crf_ms / batch, 'achieved': crf_flop / (crf_ms * 1e-3) / 1e12,
                                      'peak': PEAK_F32 / 1e12, 'unit': 'TFLOP/s', 'frac': crf_flop / (crf_ms * 1e-3) / PEAK_F32,
                                      'algorithmic_gflop_per_batch': crf_flop / 1e9}}}
    if world.rank == 0 and world.size == 1 and not args.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline_post(probs0[:batch].cpu().numpy(), (300, 300), 2)
        res['cpu_baseline']['sample'] += ' (the plain chain without network, watershed, CRF and annotation encoding: the part of this workload the reference runs on the CPU in every shipped pipeline)'
    return res


def north_star_block(net, x, batch, hw, enc, dtype, dev, stream):
    """The two figures BASELINE.json's north_star / metric name next to the train rate, measured in the SAME driver-run process:
    (1) the ResNet-U-Net eval FORWARD at the train batch (target: >= 0.40 of the MFMA peak) and (2) post-processing ms/img of the
    configs[3] chain -- plain (resize, threshold, label, dilate, score) and full (+ dense CRF + watershed extension), with and without
    the annotation encoding.  Inputs resident in HBM; ~3 s of GPU time."""
    from mapping_challenge_amd import postprocessing as post, utils
    import synthetic_inputs as post_ref                           # synthetic blob-like probability maps (inputs; nothing of oracle/ on a measured leg)

    def per_call(fn, iters, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters
    was = net.training
    net.eval()
    try:
        net.predict_proba(x)                                      # builds (and tunes) the eval program, folds the BatchNorms
        fwd_s = per_call(lambda: net.predict_proba(x), 200, warm=5)
        prog = net._program(batch, hw, hw, False, dev)
        fam = family_times(list(prog.fwd), stream)
    finally:
        net.train(was)
    conv = {k: fam.get('msc_conv_igemm', {}).get(k, 0.0) + fam.get('msc_bottleneck_fused', {}).get(k, 0.0) for k in ('ms', 'launches', 'flops')}
    peak = PEAK_F32 if dtype == 'fp32' else PEAK_BF16
    out = {'forward': {'workload': 'ResNet%d-U-Net eval forward + fused softmax, batch %d, %dx%d network input, %s' % (enc, batch, hw, hw, dtype),
                       'img_s': batch / fwd_s, 'ms_per_batch': 1e3 * fwd_s, 'launches': len(prog.fwd),
                       'algorithmic_gflop_per_img': conv['flops'] / batch / 1e9,
                       'frac_of_mfma_peak': conv['flops'] / fwd_s / peak,                 # wall clock of the whole forward
                       'conv_family_frac_by_events': conv['flops'] / (conv['ms'] * 1e-3) / peak if conv['ms'] else None,
                       'target_frac': 0.40}}
    nb = 128                                                      # images per tail call (four network batches, as --workload e2e)
    probs = torch.from_numpy(post_ref.synthetic_probs(nb, hw, hw, seed=1234)).to(dev)
    gen = torch.Generator().manual_seed(1234)
    rgb = torch.randint(0, 256, (nb, hw, hw, 3), dtype=torch.uint8, generator=gen).to(dev)
    ids, cat_ids, layers = list(range(nb)), [None, 100], [1, 1]
    chain = {'plain': per_call(lambda: post.postprocess_device(probs, (300, 300), 0, 2), 10),
             'full': per_call(lambda: post.postprocess_device(post.dense_crf_batch(probs, rgb), (300, 300), 0, 2, watershed_selem_size=5), 5)}
    annot = {'plain': per_call(lambda: utils.annotations_json_from_probabilities(ids, probs, cat_ids, layers, (300, 300), 0, 2), 10),
             'full': per_call(lambda: utils.annotations_json_from_probabilities(ids, probs, cat_ids, layers, (300, 300), 0, 2, watershed_selem_size=5,
                                                                              crf_images=rgb), 5)}
    out['post_ms_per_img'] = {'plain_chain': 1e3 * chain['plain'] / nb, 'full_chain_crf_watershed': 1e3 * chain['full'] / nb,
                              'plain_to_annotations': 1e3 * annot['plain'] / nb, 'full_to_annotations': 1e3 * annot['full'] / nb,
                              'images_per_call': nb,
                              'note': 'synthetic blob maps %dx%d -> 300x300, labels stay in HBM; plain = resize, threshold, 4-connected labelling, 2x2 label '
                                      'dilation, scoring (the shipped pipeline); full = dense CRF (5 iterations) in front and the erosion-marker watershed '
                                      '(extension) in place of plain labelling; to_annotations adds COCO RLE + bbox + the JSON text' % (hw, hw)}
    full_img_s = nb / annot['full']
    out['post_not_the_bottleneck'] = {'forward_img_s': batch / fwd_s, 'full_tail_img_s': full_img_s, 'plain_tail_img_s': nb / annot['plain'],
                                      'holds_for_full_chain': full_img_s >= batch / fwd_s}
    return out


DEFAULT_STEPS = {'train': 200, 'infer': 200, 'tta': 20, 'post': 100, 'annot': 50, 'e2e': 30}     # seconds of GPU work, not milliseconds


def tile_text(hw):
    return {256: '300x300 tiles as 256x256 network input (reference loader_mode resize)',
            320: '300x300 tiles replicate-padded to 320x320 network input (reference loader_mode crop_and_pad)'}.get(hw, '%dx%d tiles' % (hw, hw))


def reexec_argv(gpus, argv, port=None):
    """argv that turns `python bench.py --gpus N ...` into N ranks on this node: the launch line the driver itself uses
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`),
    with a free local port when none is given; the original flags follow unchanged"""
    if port is None:
        import socket
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus),
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='train', choices=['train', 'infer', 'tta', 'post', 'annot', 'e2e'])
    ap.add_argument('--encoder', type=int, default=None)
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--size', type=int, default=None)
    ap.add_argument('--dtype', default=None, choices=['bf16', 'fp16', 'fp32'])
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-breakdown', action='store_true')
    ap.add_argument('--no-north-star', action='store_true', help='skip the forward / post-processing block of the default train line')
    ap.add_argument('--dump-launches', default=None, help='write per-launch conv/wgrad timings to this JSON file')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = DEFAULT_STEPS[args.workload]
    if args.dtype is None:
        args.dtype = 'fp16' if args.workload == 'tta' else 'bf16'
    if args.size is None:
        args.size = 512 if args.workload == 'tta' else 256

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` on its own: become N ranks (one per GPU) under torch.distributed.run
        os.execv(sys.executable, reexec_argv(args.gpus, sys.argv[1:]))

    from mapping_challenge_amd.distributed import World
    world = World.from_env()
    if args.gpus != world.size:
        raise SystemExit('--gpus %d but WORLD_SIZE is %d' % (args.gpus, world.size))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('MSC_DIST_ONE_DEVICE') == '1':      # validation only (with MSC_DIST_BACKEND=gloo): every rank on cuda:0 of a one-GPU box
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    stream = torch.cuda.current_stream(dev).cuda_stream
    hw = args.size
    result = {'n_gpus': world.size, 'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': True, 'scaling': 'weak',
              'vs_baseline': None, 'data': 'synthetic', 'dtype': args.dtype,
              'rccl_ranks': world.size if world.size > 1 else 0}

    def timed(step_fn):
        for _ in range(args.warmup):
            step_fn()
        world.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        torch.cuda.synchronize()
        world.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world.size > 1:
            import torch.distributed as dist
            world.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item())

    if args.workload in ('train', 'infer', 'tta'):
        from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
        from mapping_challenge_amd.unet_models import UNetResNet
        import synthetic_inputs as losses_ref       # synthetic data / seeded weights (inputs; oracle/ is imported by the cpu_baseline leg only)
        unet_ref = losses_ref
        enc = args.encoder or {'train': 101, 'infer': 34, 'tta': 152}[args.workload]
        batch = args.batch or (64 if args.workload == 'tta' else 32)
        net = UNetResNet(enc, 2, num_filters=32, dropout_2d=0.0, is_deconv=True, compute_dtype=args.dtype)
        net.load_state_dict(unet_ref.seeded_state_dict(net))
        net.flatten_parameters(dev)
        world.sync_model(net)
        # gradient exchange: fp32 ring all-reduce, the reference's reduce-add precision (nn.DataParallel, src/models.py:65); the 16-bit
        # all-to-all wire (half the bytes, each rank's partial rounded before the sum) is opt-in: MSC_GRAD_WIRE=bf16
        world.grad_wire = os.environ.get('MSC_GRAD_WIRE', 'fp32')
        x = unet_ref.synthetic_batch(batch, hw, hw, seed=1234 + world.rank).to(dev)
        fwd_gf = None
        if args.workload == 'train':
            tgt = losses_ref.synthetic_target(min(batch, 4), hw, hw, seed=world.rank)
            tgt = tgt.repeat((batch + tgt.shape[0] - 1) // tgt.shape[0], 1, 1, 1)[:batch].contiguous().to(dev)
            net.train()
            # MSC_FORCE_COLLECTIVES=1: run the multi-GPU step -- piecewise graphs + RCCL calls on a process group of one
            # rank -- on a single GPU, to measure what it costs over the one-graph step
            force = os.environ.get('MSC_FORCE_COLLECTIVES') == '1' and world.size == 1
            if force:
                import torch.distributed as dist
                if not dist.is_initialized():
                    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                    os.environ.setdefault('MASTER_PORT', '29533')
                    dist.init_process_group('nccl', rank=0, world_size=1)
            step = TrainStep(net, LossSpec.mixed(ARCH), HipAdam(net, lr=5e-4, weight_decay=1e-4), world=world,
                             use_graph=not args.no_graph,
                             force_collectives=force)
            dt = timed(lambda: step(x, tgt))
            prog = step.prog
            loss = float(step.loss.item())
            result.update(metric='images/sec (train fwd+bwd) ResNet%d-U-Net' % enc, unit='img/s',
                          config={'workload': 'ResNet%d-U-Net train step (fwd + weighted-CE/Dice loss + bwd + Adam+L2), batch %d/GPU, '
                                              '%s, %s compute, fp32 accumulate / master weights; random-init seeded weights'
                                              % (enc, batch, tile_text(hw), args.dtype),
                                  'global_batch': batch * world.size, 'parallelism': 'dp%d' % world.size,
                                  'hipgraph': step.graph is not None or step.pieces is not None,
                                  'grad_wire': world.grad_wire if world.size > 1 else None, 'final_loss': loss})
        elif args.workload == 'tta':
            from mapping_challenge_amd import tta
            net.eval()
            specs = tta.tta_specs(flip_ud=True, flip_lr=True)        # identity + 3: the reference generator with rotation off
            dt = timed(lambda: tta.predict_tta(net, x, specs, 'gmean'))
            prog = net._program(batch, hw, hw, False, dev)
            result.update(metric='images/sec (inference forward, test-time augmentation x%d) ResNet%d-U-Net' % (len(specs), enc), unit='img/s',
                          config={'workload': 'ResNet%d-U-Net eval forward + fused softmax over %d TTA variants (flips), geometric-mean '
                                              'aggregation on the device, batch %d tiles/GPU, %s, %s compute'
                                              % (enc, len(specs), batch, tile_text(hw), args.dtype),
                                  'global_batch': batch * world.size, 'parallelism': 'dp%d' % world.size, 'tta_variants': len(specs),
                                  'forward_passes_per_sec': batch * world.size * len(specs) * args.steps / dt})
        else:
            net.eval()
            dt = timed(lambda: net.predict_proba(x))
            prog = net._program(batch, hw, hw, False, dev)
            result.update(metric='images/sec (inference forward) ResNet%d-U-Net' % enc, unit='img/s',
                          config={'workload': 'ResNet%d-U-Net eval forward + fused softmax, batch %d/GPU, %s, %s compute'
                                              % (enc, batch, tile_text(hw), args.dtype),
                                  'global_batch': batch * world.size, 'parallelism': 'dp%d' % world.size})
        value = batch * world.size * args.steps / dt
        result.update(value=value, ms_per_step=1e3 * dt / args.steps)
        if world.rank == 0 and not args.no_breakdown:
            launches = list(prog.fwd) + (list(prog.bwd) + list(step.opt.launches()) if args.workload == 'train' else [])
            if args.workload == 'train':
                net._flat[1].zero_()
            fam = family_times(launches, stream)
            if args.dump_launches:
                dump_launches(launches, stream, args.dump_launches)
            conv = {k: fam.get('msc_conv_igemm', {}).get(k, 0.0) + fam.get('msc_bottleneck_fused', {}).get(k, 0.0) for k in ('ms', 'launches', 'flops')}
            # the same family enqueued back to back between ONE pair of events (the per-launch events above put a host round trip
            # between two kernels; this is the figure that matches the rocprofv3 kernel-time sum of the family)
            conv_only = [(fn, a) for fn, a in launches if fn.__name__ in ('msc_conv_igemm', 'msc_bottleneck_fused')]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for fn, a in conv_only:
                assert fn(*a, stream) == 0
            e1.record()
            torch.cuda.synchronize()
            conv_b2b_ms = e0.elapsed_time(e1)
            fused = fam.get('msc_bottleneck_fused')
            wg = {'ms': 0.0, 'launches': 0.0, 'flops': 0.0}
            for name in ('msc_conv_wgrad', 'msc_wgrad_group_run'):
                for key in wg:
                    wg[key] += fam.get(name, {}).get(key, 0.0)
            total_ms = sum(f['ms'] for f in fam.values())
            dom = max(fam.items(), key=lambda kv: kv[1]['ms'])
            peak = PEAK_F32 if args.dtype == 'fp32' else PEAK_BF16        # bf16 and fp16 MFMA share the dense peak
            ach = conv['flops'] / (conv['ms'] * 1e-3) if conv['ms'] else 0.0
            # HBM bytes per launch from separate rocprofv3 --pmc passes (run_gpu_round.sh pmc -> tools/pmc_summary.py).  The file is
            # stamped with a hash of the kernel sources it was measured on: a number measured on other kernels is not reported
            traffic, traffic_note = None, 'no PMC pass on file'
            pmc_file = os.path.join(ROOT, 'profiles', 'pmc_traffic_%s_r%d.json' % (args.workload, enc))
            if os.path.exists(pmc_file):
                sys.path.insert(0, os.path.join(ROOT, 'tools'))
                from pmc_summary import csrc_stamp
                pmc = json.load(open(pmc_file))
                if pmc.get('csrc_sha16') == csrc_stamp():
                    traffic, traffic_note = pmc.get('hbm_bytes_per_launch'), 'rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per launch, measured on these kernel sources (csrc sha %s)' % pmc['csrc_sha16']
                else:
                    traffic_note = 'stale: %s was measured on kernel sources %s, the run uses %s' % (os.path.basename(pmc_file), pmc.get('csrc_sha16', 'unstamped'), csrc_stamp())
            result['roofline'] = {
                'kernel': ('msc_conv_igemm family: conv3x3_halo_dma_kernel (3x3 stride-1 layers) + conv_igemm_dma_kernel (1x1, strided, transposed) '
                           '+ the 32-channel / stem halo kernels + conv1x1_stream_kernel%s; conv / dgrad / deconv, %d launches per step, per-layer autotuned configuration; '
                           'the time includes the BatchNorm statistics, residual-join reductions and ReLU-backward / bias sums their epilogues carry'
                           % ((' + bottleneck_fused_kernel (%d eval-mode identity Bottlenecks, one launch each: %.3f ms, %.0f TFLOP/s)'
                               % (round(fused['launches']), fused['ms'], fused['flops'] / (fused['ms'] * 1e-3) / 1e12)) if fused else '',
                              round(conv['launches']))),
                'bound': 'mfma', 'achieved': ach / 1e12, 'peak': peak / 1e12, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': traffic,
                'traffic_note': traffic_note,
                'back_to_back': {'ms_per_step': conv_b2b_ms, 'achieved': conv['flops'] / (conv_b2b_ms * 1e-3) / 1e12, 'frac': conv['flops'] / (conv_b2b_ms * 1e-3) / peak},
                'avg_launch_us': 1e3 * conv['ms'] / max(conv['launches'], 1),
                'algorithmic_gflop_per_step': conv['flops'] / 1e9,
                'wgrad': {'achieved': (wg['flops'] / (wg['ms'] * 1e-3) / 1e12) if wg['ms'] else None, 'ms_per_step': wg['ms'],
                          'layers': round(wg['launches'])},
                'hbm_family': (lambda bn: {'kernel': 'bn_apply / bn_bwd_reduce / bn_bwd_apply (BatchNorm elementwise passes)', 'bound': 'hbm',
                                           'achieved': bn[0] / (bn[1] * 1e-3) / 1e9 if bn[1] else None, 'peak': PEAK_HBM / 1e9, 'unit': 'GB/s',
                                           'frac': bn[0] / (bn[1] * 1e-3) / PEAK_HBM if bn[1] else None, 'ms_per_step': bn[1],
                                           'algorithmic_gb_per_step': bn[0] / 1e9})(
                    (sum(fam.get(k, {}).get('bytes', 0.0) for k in BN_TENSORS), sum(fam.get(k, {}).get('ms', 0.0) for k in BN_TENSORS))),
                'dominant_family': dom[0], 'family_ms_per_step': {k: round(v['ms'], 3) for k, v in sorted(fam.items())},
                'sum_kernel_ms_per_step': total_ms,
                'whole_step_frac_of_mfma_peak': (conv['flops'] + wg['flops']) * result['config'].get('tta_variants', 1) * (args.steps / dt) / peak}
        if args.workload == 'train' and world.rank == 0 and world.size == 1 and not args.no_north_star:
            # the north-star figures (forward fraction of the MFMA peak, post-processing ms/img) in the driver-run line itself
            result['north_star'] = north_star_block(net, x, batch, hw, enc, args.dtype, dev, stream)
        if world.rank == 0 and world.size == 1 and not args.no_cpu_baseline:      # reported at N=1 only
            result['cpu_baseline'] = cpu_baseline_train(enc, hw) if args.workload == 'train' else cpu_baseline_infer(enc, hw, n=4 if hw <= 320 else 1)
    elif args.workload == 'e2e':
        result.update(bench_e2e(args, world, dev, stream, timed))
    elif args.workload == 'annot':
        # labelled 300x300 layers (2 per image, on the device) -> COCO RLE strings + boxes on the host
        from mapping_challenge_amd import postprocessing as post, utils
        import synthetic_inputs as post_ref
        batch = args.batch or 64
        probs_h = post_ref.synthetic_probs(batch, 256, 256, seed=1234 + world.rank)
        res = post.postprocess_batch(torch.from_numpy(probs_h).to(dev), (300, 300), 0, 2)
        layers_h = np.concatenate([labels for labels, _ in res]).astype(np.int32)
        layers = torch.from_numpy(layers_h).to(dev)
        enc = utils.encode_labels(layers)
        n_inst = sum(len(e) for e in enc)
        dt = timed(lambda: utils.encode_labels(layers))
        value = batch * world.size * args.steps / dt
        bytes_per_img = 2 * 300 * 300 * 4 * 4.0      # label read + transpose write/read + scan flags, per image (2 layers)
        result.update(metric='annotation encoding images/sec (instances of 2 layers of 300x300 labels -> COCO RLE + bbox)', unit='img/s',
                      value=value, ms_per_step=1e3 * dt / args.steps, dtype='i32/u8',
                      config={'workload': 'RLE + bbox encoding of %d images x 2 label layers (300x300), %d instances per step' % (batch, n_inst),
                              'ms_per_img': 1e3 / value * world.size},
                      roofline={'bound': 'hbm', 'achieved': value * bytes_per_img / 1e9 / world.size, 'peak': PEAK_HBM / 1e9, 'unit': 'GB/s',
                                'frac': value * bytes_per_img / world.size / PEAK_HBM, 'traffic': None,
                                'note': 'two synchronous calls (sizes return to the host) + D2H of table and strings; latency bound'})
        if world.rank == 0 and world.size == 1 and not args.no_cpu_baseline:      # reported at N=1 only
            result['cpu_baseline'] = cpu_baseline_annot(layers_h)
    else:
        from mapping_challenge_amd import postprocessing as post
        import synthetic_inputs as post_ref
        batch = args.batch or 64
        probs_h = post_ref.synthetic_probs(batch, 256, 256, seed=1234 + world.rank)
        probs = torch.from_numpy(probs_h).to(dev)
        dt = timed(lambda: post.postprocess_batch(probs, (300, 300), 0, 2))
        value = batch * world.size * args.steps / dt
        # the same chain with the label images left in HBM (what utils.annotations_from_probabilities consumes) and with the
        # watershed extension in place of plain labelling (WATERSHED.md): secondary figures, not `value`
        dt_dev = timed(lambda: post.postprocess_device(probs, (300, 300), 0, 2))
        dt_ws = timed(lambda: post.postprocess_device(probs, (300, 300), 0, 2, watershed_selem_size=5))
        bytes_per_img = 3.4e6
        result.update(metric='post-processing images/sec (resize 256->300, threshold, label, dilate k=2, score)', unit='img/s',
                      value=value, ms_per_step=1e3 * dt / args.steps, dtype='u8/i32/f32',
                      config={'workload': 'mask post-processing of %d 256x256 2-class probability maps per step' % batch,
                              'ms_per_img': 1e3 / value * world.size,
                              'ms_per_img_labels_stay_on_device': 1e3 * dt_dev / (batch * args.steps),
                              'ms_per_img_with_watershed_extension_on_device': 1e3 * dt_ws / (batch * args.steps)},
                      roofline={'bound': 'hbm', 'achieved': value * bytes_per_img / 1e9 / world.size, 'peak': PEAK_HBM / 1e9, 'unit': 'GB/s',
                                'frac': value * bytes_per_img / world.size / PEAK_HBM, 'traffic': None,
                                'note': 'whole chain incl. the final D2H of labels; latency/launch bound'})
        if world.rank == 0 and world.size == 1 and not args.no_cpu_baseline:      # reported at N=1 only
            result['cpu_baseline'] = cpu_baseline_post(probs_h, (300, 300), 2)
    if world.rank == 0:
        print(json.dumps(result))
    world.barrier()


if __name__ == '__main__':
    main()
