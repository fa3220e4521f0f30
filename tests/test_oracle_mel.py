"""The mel oracle restates librosa (absent here): parity unpinned against
librosa itself.  These tests cross-check it against two independent
implementations that ARE installed, and pin the reference's
reshape-not-transpose (feature_extraction.py:49-50)."""
import numpy as np
import pytest
import torch

from oracle import mel
from sound_event_detection_amd import synth

SR, NFFT, NMELS, HOP, WIN = 16000, 2048, 128, 320, 640


def test_shapes_and_reshape_quirk():
    y = synth.synth_clip(0)
    m = mel.melspectrogram(y, SR, NFFT, NMELS, HOP, WIN)
    assert m.shape == (128, 501) and m.dtype == np.float32
    f = mel.clip_feature(y, SR, NFFT, NMELS, HOP, WIN)
    assert f.shape == (501, 128)
    assert np.array_equal(f.ravel(), m.ravel())          # same bytes ...
    assert not np.array_equal(f, m.T)                    # ... and NOT a transpose


def test_filterbank_vs_transformers():
    from transformers.audio_utils import mel_filter_bank
    fb = mel_filter_bank(num_frequency_bins=1025, num_mel_filters=128, min_frequency=0.0,
                         max_frequency=SR / 2, sampling_rate=SR, norm="slaney", mel_scale="slaney")
    ours = mel.mel_filterbank(SR, NFFT, NMELS)
    assert ours.dtype == np.float32 and ours.shape == (128, 1025)
    assert np.abs(ours - fb.T).max() < 1e-8
    assert int((ours != 0).sum()) == 2020


@pytest.mark.parametrize("cfg", [(16000, 2048, 128, 320, 640, 160000), (44100, 2048, 128, 882, 1764, 441000),
                                 (16000, 1024, 64, 160, 400, 16000)])
def test_melspectrogram_vs_torch_stft(cfg):
    sr, n_fft, n_mels, hop, win, n = cfg
    y = synth.synth_clip(3, n, sr)
    ours = mel.melspectrogram(y, sr, n_fft, n_mels, hop, win)
    w = torch.hann_window(win, periodic=True, dtype=torch.float64)
    D = torch.stft(torch.from_numpy(y).double(), n_fft, hop, win, w, center=True, pad_mode="constant",
                   return_complex=True)
    S = (D.abs() ** 2).numpy()
    ref = mel.mel_filterbank(sr, n_fft, n_mels).astype(np.float64) @ S
    assert ours.shape == ref.shape == (n_mels, 1 + n // hop)
    rel = np.abs(ours - ref) / np.maximum(np.abs(ref), 1e-12)
    assert rel.max() < 2e-5


def test_float64_tail_clip_stays_float64():
    # feature_extraction.py:41-42 pads a short clip with float64 zeros
    y = np.append(synth.synth_clip(1, 100000), np.zeros(60000))
    assert y.dtype == np.float64
    m = mel.melspectrogram(y, SR, NFFT, NMELS, HOP, WIN)
    assert m.dtype == np.float64 and m.shape == (128, 501)


def test_optional_stages():
    S = np.array([[1.0, 10.0, 0.0]], np.float32)
    assert np.allclose(mel.power_to_db(S), [[0.0, 10.0, -100.0]])
