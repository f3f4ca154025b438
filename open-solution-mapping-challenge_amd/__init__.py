"""MI355X-native U-Net segmentation hot path of neptune-ai/open-solution-mapping-challenge.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed only) calling
hand-written HIP kernels for gfx950 through the C ABI in include/msc.h (libmsc_hip.so, ctypes).

  _lib            ctypes binding of the C ABI; raises if the HIP library is missing (no fallback)
  unet_models     UNetResNet (drop-in for src/unet_models.py:315-403), forward+backward on HIP
  trainer         TrainStep (forward, fused loss, backward, RCCL exchange, Adam as one hipGraph), HipAdam, LossSpec
  models          PyTorchUNet / PyTorchUNetWeighted (+Stream) transformers (src/models.py:50-209)
  postprocessing  the functions of src/postprocessing.py:48-258 on HIP (+ batched variants)
  tta / utils / preparation   test-time augmentation, COCO RLE + bbox encoding, target preparation on the device
  callbacks       the reference's callback protocol for standalone use (src/steps/pytorch/callbacks.py)
  steps           Step / BaseTransformer operator API mirror (src/steps/base.py)
  pipelines       unet / unet_weighted / mask_postprocessing graphs (src/pipelines.py:12-52,248-304)
  distributed     one-process-per-GPU data parallel over RCCL (replaces nn.DataParallel, src/models.py:65)
"""
__version__ = '0.1.0'
