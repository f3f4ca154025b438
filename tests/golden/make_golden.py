"""Generates tests/golden/*.npz by executing the REFERENCE's own code (unmodified, through
oracle/ref_import.py's shims) on seeded synthetic inputs.  Run in the build container only
(/root/reference does not exist on the GPU box):   python tests/golden/make_golden.py
Inputs are regenerated from seeds by the tests (oracle.unet_ref.seeded_state_dict / synthetic_batch,
oracle.post_ref.synthetic_probs, oracle.losses_ref.synthetic_target), so only outputs are stored.
"""
import os
import sys
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_import, unet_ref, post_ref, losses_ref  # noqa: E402


def ref_loss(rm, weighted, dice_activation='softmax'):
    if not weighted:
        return ref_import.ref('steps.pytorch.validation').multiclass_segmentation_loss
    wf = partial(rm.get_weights, w0=50, sigma=10, imsize=(256, 256))
    return partial(rm.mixed_dice_cross_entropy_loss, dice_weight=0.2, cross_entropy_weight=1.0,
                   dice_loss=partial(rm.multiclass_dice_loss, excluded_classes=[0]),
                   cross_entropy_loss=partial(rm.multiclass_weighted_cross_entropy, weights_function=wf),
                   smooth=1, dice_activation=dice_activation)


def main():
    um, pp, rm = ref_import.ref('unet_models'), ref_import.ref('postprocessing'), ref_import.ref('models')
    torch.manual_seed(1234)
    # ---- network: reference UNetResNet, eval logits + one training backward (mixed loss)
    for depth, hw, n in ((34, 64, 2), (101, 64, 2)):
        net = um.UNetResNet(depth, 2, num_filters=32, dropout_2d=0.0, pretrained=True, is_deconv=True)
        net.load_state_dict(unet_ref.seeded_state_dict(net))
        x = unet_ref.synthetic_batch(n, hw, hw)
        net.eval()
        with torch.no_grad():
            logits_eval = net(x).numpy()
        net.train()
        tgt = losses_ref.synthetic_target(n, hw, hw)
        out = net(x)
        loss = ref_loss(rm, True)(out, tgt)
        loss.backward()
        g = dict(net.named_parameters())
        np.savez_compressed(os.path.join(HERE, 'unet_r%d_%d.npz' % (depth, hw)),
                            logits_eval=logits_eval, logits_train=out.detach().numpy(), loss=np.float32(loss.item()),
                            g_final_w=g['final.weight'].grad.numpy(), g_final_b=g['final.bias'].grad.numpy(),
                            g_conv1=g['encoder.conv1.weight'].grad.numpy()[:8],
                            g_bn1_w=g['encoder.bn1.weight'].grad.numpy(),
                            g_dec1_deconv=g['dec1.block.1.weight'].grad.numpy()[:4, :4],
                            g_center_conv_b=g['center.block.0.conv.bias'].grad.numpy(),
                            g_l2_conv1=g['encoder.layer2.0.conv1.weight'].grad.numpy()[:4, :8],
                            rm_bn1=net.encoder.bn1.running_mean.numpy(), rv_bn1=net.encoder.bn1.running_var.numpy())
    # ---- the timed network (ResNet101) at 128x128, batch 4 (every BatchNorm population >= 64 samples): loss, logits and a
    # digest of EVERY parameter gradient (L2 norm + its first 64 elements) from the reference's own backward
    net = um.UNetResNet(101, 2, num_filters=32, dropout_2d=0.0, pretrained=True, is_deconv=True)
    net.load_state_dict(unet_ref.seeded_state_dict(net))
    x = unet_ref.synthetic_batch(4, 128, 128, seed=12)
    tgt = losses_ref.synthetic_target(4, 128, 128, seed=12)
    net.train()
    out = net(x)
    loss = ref_loss(rm, True)(out, tgt)
    loss.backward()
    rec = {'logits_train': out.detach().numpy(), 'loss': np.float32(loss.item())}
    for name, p_ in net.named_parameters():
        if p_.grad is not None and not name.startswith('encoder.fc') and not name.split('.')[0] in ('conv1', 'conv2', 'conv3', 'conv4', 'conv5'):
            gflat = p_.grad.reshape(-1)
            rec['n|' + name] = np.float64(gflat.double().norm().item())
            rec['h|' + name] = gflat[:64].numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'unet_r101_128.npz'), **rec)
    # ---- losses on seeded logits
    rng = np.random.default_rng(1234)
    logits = torch.from_numpy(rng.standard_normal((2, 2, 64, 64)).astype(np.float32) * 3).requires_grad_(True)
    tgt = losses_ref.synthetic_target(2, 64, 64, seed=7)
    rec = {}
    for name, weighted, act in (('ce', False, 'softmax'), ('mixed', True, 'softmax'), ('mixed_sigmoid', True, 'sigmoid')):
        logits.grad = None
        t = tgt if weighted else tgt[:, :1]
        loss = ref_loss(rm, weighted, act)(logits, t)
        loss.backward()
        rec['loss_' + name] = np.float32(loss.item())
        rec['dlogits_' + name] = logits.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'loss.npz'), **rec)
    # ---- post-processing chain (reference functions on the skimage-on-scipy shim)
    probs = post_ref.synthetic_probs(3, 64, 64, seed=1234, smooth=2.0)
    rec = {}
    for i, p in enumerate(probs):
        r = pp.resize_image(p, (75, 75))
        lay = pp.categorize_multilayer_image(r)
        lab = pp.label_multilayer_image(lay)
        dil = pp.dilate_image(lab, 2)
        _, scores = pp.build_score(dil, r)
        ero = pp.erode_image(lay[1], 3)
        rec.update({'resized_%d' % i: r.astype(np.float32), 'layers_%d' % i: lay, 'labels_%d' % i: lab,
                    'dilated2_%d' % i: dil, 'dilated3_%d' % i: pp.dilate_image(lab, 3), 'eroded3_%d' % i: ero,
                    'scores0_%d' % i: np.asarray(scores[0], np.float64), 'scores1_%d' % i: np.asarray(scores[1], np.float64)})
    kat = np.array([[0, 0, 1, 1], [1, 0, 0, 0], [1, 1, 1, 0], [0, 0, 1, 0]])
    rec['kat_multiclass'] = pp.label_multiclass_image(kat)
    np.savez_compressed(os.path.join(HERE, 'post.npz'), **rec)
    print('golden files written to', HERE)


if __name__ == '__main__':
    main()
