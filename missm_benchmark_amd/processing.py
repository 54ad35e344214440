"""GPU-side preprocessing for the image-like modalities.

The reference prepares every sample on the host, one at a time (``transform_dict[m](config)``:
languagebind/image/processing_image.py:18-28, thermal/processing_thermal.py:18-28, depth/processing_depth.py:21-55 -
ToTensor / DepthNorm, ``Resize(224, bicubic)``, ``CenterCrop(224)``, ``Normalize``) and copies the finished float tensors with a
synchronous pageable ``to_device`` (languagebind/__init__.py:87-89; train_ddp.py:224-229).  Here the DECODED image (uint8 HWC
as PIL / numpy hand it over, float32 HW for depth maps) is staged through pinned memory with an asynchronous copy - a quarter of
the bytes of the float tensor for uint8 sources - and one HIP launch per image does the rest on the device
(``missm_preprocess_image``).  Decoding files (PIL / cv2 / decord), the BPE tokenizer and the kaldi filter-bank front end of the
audio model stay outside (host libraries that are not in this image); audio / video tensors prepared elsewhere are fed as is.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import ops

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)     # image/processing_image.py:10-11
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


def to_device_async(x, device):
    """``to_device`` of the reference (languagebind/__init__.py:87-89) through pinned staging: the copies are enqueued on the
    current stream and the host does not wait for them"""
    out = {}
    for k, v in x.items():
        if torch.is_tensor(v) and v.device.type == "cpu" and torch.device(device).type == "cuda":
            v = v if v.is_pinned() else v.pin_memory()
            out[k] = v.to(device, non_blocking=True)
        else:
            out[k] = v.to(device)
    return out


class _GpuImageTransform:
    """callable like the reference's ``transform``: decoded image(s) in, ``pixel_values`` fp32 [N, 3, S, S] on the GPU out"""

    def __init__(self, size: int = 224, mean: Sequence[float] = OPENAI_DATASET_MEAN, std: Sequence[float] = OPENAI_DATASET_STD,
                 device="cuda"):
        self.size, self.mean, self.std, self.device = int(size), tuple(mean), tuple(std), torch.device(device)

    # what the kernel applies to a source value before resampling: v = clip(v * scale, lo, hi) / div
    pre = dict(pre_scale=1.0 / 255.0, pre_min=-3.0e38, pre_max=0.0, pre_div=1.0)
    chw = False

    def _as_tensor(self, img) -> torch.Tensor:
        t = torch.as_tensor(np.ascontiguousarray(img)) if not torch.is_tensor(img) else img.contiguous()
        if t.dim() == 2:
            t = t.unsqueeze(0 if self.chw else -1)
        return t

    def __call__(self, images):
        single = not isinstance(images, (list, tuple))
        imgs = [self._as_tensor(i) for i in ([images] if single else images)]
        out = torch.empty(len(imgs), 3, self.size, self.size, device=self.device, dtype=torch.float32)
        for n, t in enumerate(imgs):
            if t.device.type == "cpu":
                t = (t if t.is_pinned() else t.pin_memory()).to(self.device, non_blocking=True)
            if t.dtype not in (torch.uint8, torch.float32):
                t = t.to(torch.float32)
            ops.preprocess_image(t, out[n], chw=self.chw, mean=self.mean, std=self.std, **self.pre)
        return out[0] if single else out


class ImageTransform(_GpuImageTransform):
    """image / thermal: ToTensor (uint8 HWC -> [0, 1]) -> Resize -> CenterCrop -> Normalize (image/processing_image.py:18-28)"""


class DepthTransform(_GpuImageTransform):
    """depth: DepthNorm (millimetres -> metres, clip to [0.01, max_depth], / max_depth), replicated to three channels, then as
    the image transform (depth/processing_depth.py:21-55).  ``max_depth == 0`` (normalise by the image's own maximum) needs a
    reduction over the image first and is not fused here."""

    def __init__(self, max_depth: float = 10.0, **kw):
        super().__init__(**kw)
        if max_depth <= 0:
            raise NotImplementedError("DepthNorm with max_depth == 0 (divide by the image's own maximum) is not implemented; "
                                      "the reference configuration uses max_depth = 10")
        self.pre = dict(pre_scale=1.0 / 1000.0, pre_min=0.01, pre_max=float(max_depth), pre_div=float(max_depth))
