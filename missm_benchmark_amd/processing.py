"""GPU-side preprocessing for the image-like modalities.

The reference prepares every sample on the host, one at a time (``transform_dict[m](config)``:
languagebind/image/processing_image.py:18-28, thermal/processing_thermal.py:18-28, depth/processing_depth.py:21-55 -
ToTensor / DepthNorm, ``Resize(224, bicubic)``, ``CenterCrop(224)``, ``Normalize``) and copies the finished float tensors with a
synchronous pageable ``to_device`` (languagebind/__init__.py:87-89; train_ddp.py:224-229).  Here the DECODED image (uint8 HWC
as PIL / numpy hand it over, float32 HW for depth maps) is staged through pinned memory with an asynchronous copy - a quarter of
the bytes of the float tensor for uint8 sources - and one HIP launch per image does the rest on the device
(``missm_preprocess_image``).  The audio model's front end (languagebind/audio/processing_audio.py:31-111: resample, Kaldi filter
bank, three-chunk assembly, normalisation) runs on the device too (``AudioTransform``; ``csrc/audio.hip``).  Decoding files (PIL /
cv2 / decord / soundfile) and the BPE tokenizer stay outside (host libraries that are not in this image): decoded arrays go in.
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


DEFAULT_AUDIO_FRAME_SHIFT_MS = 10      # audio/processing_audio.py:29


def sinc_resample_kernels(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """The windowed-sinc table of ``torchaudio.functional.resample`` (``sinc_interp_hann``, its defaults), built on the host in float64
    and rounded to fp32 once: kernels [new, 2 * width + orig] for the frequencies divided by their gcd.  Returns (kernels, width,
    orig, new)."""
    import math
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    k = k * window * (base / orig)
    return torch.from_numpy(k.astype(np.float32)), width, orig, new


class AudioTransform:
    """``AudioTransform`` of the reference (languagebind/audio/processing_audio.py:31-111) on the GPU: called with
    ``(audio_data [channels, n] (or [n]), origin_sr)`` like the reference's transform, returns ``pixel_values`` [3, num_mel_bins,
    target_length] on the device.  Steps: resample to ``audio_sample_rate`` when the rates differ (:44-46), subtract the clip's mean and
    take the Kaldi filter bank of channel 0 (:95-108), cut three ``target_length``-frame chunks - their starts drawn with
    ``np.random.choice`` over the thirds of the admissible range, exactly the reference's draws in the reference's order, so a seeded
    run picks the same chunks - or tile a short clip (:54-88), transpose and normalise ``(x - audio_mean) / (2 audio_std)`` (:89-93).
    ``config`` is the model config (its ``vision_config`` carries the audio fields, configuration_audio.py:203-208) or keyword values."""

    def __init__(self, config=None, *, sample_rate=None, num_mel_bins=None, target_length=None, audio_mean=None, audio_std=None, device="cuda"):
        vc = {}
        if config is not None:
            vc = config.get("vision_config", config) if isinstance(config, dict) else getattr(config, "vision_config", config)
            vc = vc if isinstance(vc, dict) else vars(vc)
        pick = lambda given, key, dflt: given if given is not None else vc.get(key, dflt)      # noqa: E731
        self.sample_rate = int(pick(sample_rate, "audio_sample_rate", 16000))
        self.num_mel_bins = int(pick(num_mel_bins, "num_mel_bins", 112))
        self.target_length = int(pick(target_length, "target_length", 1036))
        self.audio_mean = float(pick(audio_mean, "audio_mean", 0.5))
        self.audio_std = float(pick(audio_std, "audio_std", 0.5))
        if self.num_mel_bins <= 0 or self.target_length <= 0:
            raise ValueError("AudioTransform needs num_mel_bins and target_length (the audio checkpoint's config sets them; the class defaults are 0)")
        self.device = torch.device(device)
        self._kern = {}

    def _to_device(self, x) -> torch.Tensor:
        t = torch.as_tensor(np.ascontiguousarray(x)) if not torch.is_tensor(x) else x
        t = t.to(torch.float32)
        if t.dim() == 1:
            t = t.unsqueeze(0)
        if t.device.type == "cpu":
            t = (t if t.is_pinned() else t.contiguous().pin_memory()).to(self.device, non_blocking=True)
        return t.contiguous()

    def resample(self, wave: torch.Tensor, origin_sr: int) -> torch.Tensor:
        """[channels, n] at origin_sr -> [channels, ceil(new * n / orig)] at self.sample_rate"""
        key = (int(origin_sr), self.sample_rate)
        if key not in self._kern:
            k, width, orig, new = sinc_resample_kernels(*key)
            self._kern[key] = (k.to(self.device), width, orig, new)
        k, width, orig, new = self._kern[key]
        n = wave.shape[-1]
        n_out = -(-new * n // orig)                       # ceil(new * n / orig)
        return torch.stack([ops.sinc_resample(ch.contiguous(), k, orig, new, width, n_out) for ch in wave])

    def get_mel(self, wave: torch.Tensor) -> torch.Tensor:
        """(:95-108) ``audio_data -= audio_data.mean()`` over ALL channels, then kaldi.fbank of channel 0 -> [frames, num_mel_bins]"""
        gm = None
        if wave.shape[0] == 1:
            return ops.kaldi_fbank(wave[0], self.num_mel_bins, self.sample_rate, frame_shift=DEFAULT_AUDIO_FRAME_SHIFT_MS)
        # several channels: the reference subtracts the mean of every channel's samples, the filter bank then reads channel 0
        flat = wave.reshape(-1).contiguous()
        gm = torch.empty(1, device=wave.device, dtype=torch.float32)
        from . import _lib
        _lib.call("missm_buffer_mean", flat.data_ptr(), flat.numel(), gm.data_ptr(), torch.cuda.current_stream().cuda_stream)
        ch0 = wave[0].contiguous()
        frames = _lib.load().missm_fbank_frames(ch0.numel(), float(self.sample_rate), 25.0, float(DEFAULT_AUDIO_FRAME_SHIFT_MS))
        if frames <= 0:
            raise _lib.MissmError("kaldi_fbank: waveform shorter than one frame")
        out = torch.empty(frames, self.num_mel_bins, device=wave.device, dtype=torch.float32)
        _lib.call("missm_kaldi_fbank", ch0.data_ptr(), ch0.numel(), gm.data_ptr(), out.data_ptr(), self.num_mel_bins, float(self.sample_rate), 25.0,
                  float(DEFAULT_AUDIO_FRAME_SHIFT_MS), 20.0, 0.0, 0.97, torch.cuda.current_stream().cuda_stream)
        return out

    def chunk_starts(self, total_frames: int):
        """(:56-74) the three chunk starts of a clip longer than target_length: one ``np.random.choice`` per third of the admissible
        starts, in the reference's order (front, middle, back)"""
        chunk = self.target_length
        if total_frames <= chunk:
            return (0, 0, 0)
        ranges = np.array_split(list(range(0, total_frames - chunk + 1)), 3)
        if len(ranges[1]) == 0:
            ranges[1] = [0]
        if len(ranges[2]) == 0:
            ranges[2] = [0]
        return (int(np.random.choice(ranges[0])), int(np.random.choice(ranges[1])), int(np.random.choice(ranges[2])))

    def waveform2melspec(self, wave: torch.Tensor, starts=None) -> torch.Tensor:
        mel = self.get_mel(wave)
        starts = self.chunk_starts(mel.shape[0]) if starts is None else starts
        return ops.mel_assemble(mel, self.target_length, starts, self.audio_mean, self.audio_std)

    def __call__(self, audio_data_and_origin_sr, starts=None):
        audio_data, origin_sr = audio_data_and_origin_sr
        wave = self._to_device(audio_data)
        if self.sample_rate != int(origin_sr):
            wave = self.resample(wave, int(origin_sr))
        return self.waveform2melspec(wave, starts)
