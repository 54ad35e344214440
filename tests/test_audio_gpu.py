"""GPU audio front end (csrc/audio.hip, processing.AudioTransform) against the CPU oracle's restatement of
languagebind/audio/processing_audio.py:31-111 (torchaudio's kaldi.fbank / functional.resample are absent from this image and unpinned
upstream: the oracle restates their published algorithms - PARITY UNPINNED for those two pieces; the reference's own code around them
is followed line by line).  Tolerances: log mel energies 2e-3 absolute (fp32 in-LDS radix-2 FFT vs pocketfft, logs of near-silent bins),
resampled samples 1e-5, assembled pixel_values 1e-3."""
import math

import numpy as np
import pytest
import torch

import missm_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def proc():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from missm_benchmark_amd import ops, processing
    import missm_benchmark_amd as M
    lb, _ = M.install()
    return ops, processing, lb


def _clip(seconds, sr, seed, channels=1):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(int(seconds * sr)) / sr
    base = 0.3 * torch.sin(2 * math.pi * 440.0 * t) + 0.2 * torch.sin(2 * math.pi * 2500.0 * t + 1.0) + 0.05 * torch.randn(t.shape, generator=g)
    return torch.stack([base * (1.0 - 0.3 * c) + 0.02 * c for c in range(channels)])


@pytest.mark.parametrize("sr,nbins", [(16000, 112), (16000, 128), (48000, 64), (8000, 40)])
def test_kaldi_fbank_vs_oracle(proc, sr, nbins):
    ops, _, _ = proc
    wave = _clip(1.7, sr, 3)[0] + 0.1          # a DC offset: the global mean AND the per-frame DC removal must both act
    mel = ops.kaldi_fbank(wave.cuda(), nbins, sr)
    ref = O.kaldi_fbank((wave - wave.mean())[None], sr, nbins)
    assert mel.shape == ref.shape and mel.shape[0] == 1 + (wave.numel() - int(sr * 0.025)) // int(sr * 0.010)
    assert float((mel.cpu() - ref).abs().max()) < 2e-3
    raw = ops.kaldi_fbank(wave.cuda(), nbins, sr, subtract_global_mean=False)
    assert float((raw.cpu() - O.kaldi_fbank(wave[None], sr, nbins)).abs().max()) < 2e-3
    with pytest.raises(Exception, match="shorter than one frame"):
        ops.kaldi_fbank(wave[:100].cuda(), nbins, sr)


@pytest.mark.parametrize("orig,new", [(44100, 16000), (48000, 16000), (8000, 16000), (22050, 16000)])
def test_sinc_resample_vs_oracle(proc, orig, new):
    _, processing, _ = proc
    wave = _clip(0.9, orig, 5, channels=2)
    tr = processing.AudioTransform(sample_rate=new, num_mel_bins=64, target_length=100)
    got = tr.resample(wave.cuda(), orig)
    ref = O.sinc_resample(wave, orig, new)
    assert got.shape == ref.shape == (2, math.ceil(new * wave.shape[1] / orig))
    assert float((got.cpu() - ref).abs().max()) < 1e-5


def test_audio_transform_vs_oracle_and_reference_draws(proc):
    _, processing, lb = proc
    cfg = {"vision_config": {"audio_sample_rate": 16000, "num_mel_bins": 112, "target_length": 250, "audio_mean": -4.2677393, "audio_std": 4.5689974}}
    tr = lb.transform_dict["audio"](cfg)                    # the drop-in entry (reference: transform_dict['audio'](model.modality_config['audio']))
    assert isinstance(tr, processing.AudioTransform) and (tr.num_mel_bins, tr.target_length) == (112, 250)
    # (a) a clip LONGER than target_length, another sample rate, two channels: resample + three chunks at the reference's own random starts
    long = _clip(4.1, 44100, 7, channels=2)
    np.random.seed(11)
    px = tr((long, 44100))
    assert px.shape == (3, 112, 250) and px.is_cuda
    res = O.sinc_resample(long, 44100, 16000)
    frames = 1 + (res.shape[1] - 400) // 160
    np.random.seed(11)                                      # the reference's draws, in its order (processing_audio.py:59-71)
    ranges = np.array_split(list(range(0, frames - 250 + 1)), 3)
    starts = tuple(int(np.random.choice(r)) for r in ranges)
    ref = O.audio_transform(long, 44100, 16000, 112, 250, -4.2677393, 4.5689974, starts=starts)
    assert float((px.cpu() - ref).abs().max()) < 1e-3
    assert float((tr((long, 44100), starts=starts).cpu() - ref).abs().max()) < 1e-3
    # (b) a clip SHORTER than target_length at the model's rate: tiled (mel.repeat), all three chunks equal
    short = _clip(1.2, 16000, 8)
    ps = tr((short, 16000))
    rs = O.audio_transform(short, 16000, 16000, 112, 250, -4.2677393, 4.5689974)
    assert torch.equal(ps[0], ps[2]) and float((ps.cpu() - rs).abs().max()) < 1e-3
    # (c) exactly target_length frames
    exact = _clip((400 + 249 * 160) / 16000.0, 16000, 9)
    assert 1 + (exact.shape[1] - 400) // 160 == 250
    pe = tr((exact.numpy()[0], 16000))                      # a decoded numpy array, one channel, as soundfile would hand it over
    assert float((pe.cpu() - O.audio_transform(exact, 16000, 16000, 112, 250, -4.2677393, 4.5689974)).abs().max()) < 1e-3
    with pytest.raises(ValueError, match="num_mel_bins"):
        processing.AudioTransform({"vision_config": {"num_mel_bins": 0, "target_length": 0}})


def test_spectrogram_feeds_the_audio_tower(proc):
    """pixel_values of AudioTransform ([3, mel_bins, target_length] per clip) run through an audio tower whose image is the
    (num_mel_bins, target_length) spectrogram (image/modeling_image.py:797-798)"""
    _, processing, lb = proc
    from missm_benchmark_amd.towers import ClipTower, TowerConfig
    tr = processing.AudioTransform(num_mel_bins=32, target_length=64, audio_mean=-4.27, audio_std=4.57)
    px = torch.stack([tr((_clip(0.5 + 0.2 * i, 16000, 20 + i), 16000)) for i in range(2)])
    tower = ClipTower(TowerConfig(kind="vision", hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                                  image_size=(32, 64), patch_size=16), compute_dtype=torch.float32, seed=1).cuda()
    last, pooled = tower(px)
    assert pooled.shape == (2, 64) and last.shape == (2, 1 + 2 * 4, 64) and bool(torch.isfinite(pooled).all())
