"""TEST INFRASTRUCTURE -- CPU oracle for the U-Net hot path of
neptune-ai/open-solution-mapping-challenge.

Nothing in here is shipped or measured as the product: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and only as the
checker.  The product package (`open-solution-mapping-challenge_amd/`) never imports `oracle`.

Contents
  unet_ref.py     torch-CPU fp32 restatement of UNetResNet (src/unet_models.py:315-403) + the
                  torchvision ResNet stages it borrows
  losses_ref.py   torch-CPU restatement of the losses (src/models.py:310-454,
                  src/steps/pytorch/validation.py:8-28) and of Adam+L2 (src/models.py:57,287-292)
  post_ref.py     numpy/scipy restatement of the mask post-processing chain
                  (src/postprocessing.py:48-258, src/utils.py:231-273,328-339)
  crf_ref.py      exact windowed mean-field dense CRF (src/postprocessing.py:183-225; pydensecrf
                  is not vendored: PARITY UNPINNED)
  post_ref.c      plain-C restatement of threshold -> label -> dilate -> score, used as the
                  single-thread CPU baseline ("port") in bench.py
  ref_import.py   imports the reference's own Python UNMODIFIED from /root/reference on top of
                  shims/ (only in the build container; /root/reference does not exist on the GPU
                  box) -- used to pin the restatements and to generate tests/golden/
  shims/          stand-ins for the third-party modules the reference needs at import time
"""
