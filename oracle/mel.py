"""Oracle: mel frontend.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates what feature_extraction.py:41-50 obtains from
``librosa.feature.melspectrogram(y=, sr=, n_fft=, n_mels=, hop_length=,
win_length=)`` with every other argument at its librosa >= 0.10 default
(window='hann', center=True, pad_mode='constant', power=2.0, htk=False,
norm='slaney', fmin=0, fmax=sr/2), followed by the reference's
reshape-not-transpose (feature_extraction.py:49-50).

librosa is a third-party dependency of the reference that is not installed
here and not pinned by the reference; this file restates its published
algorithm (librosa 0.10 ``core/spectrum.py::stft``, ``filters.py::mel``,
``core/convert.py::mel_frequencies``).  Parity unpinned; cross-checked in
tests/test_oracle_mel.py against torch.stft and transformers.audio_utils.
"""
from __future__ import annotations

import numpy as np


def hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz,
                    min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr: int, n_fft: int, n_mels: int) -> np.ndarray:
    """float32 [n_mels, 1 + n_fft//2]; Slaney scale, Slaney area normalisation."""
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, float(sr) / 2, n_bins, endpoint=True)
    mel_pts = np.linspace(hz_to_mel_slaney(0.0), hz_to_mel_slaney(float(sr) / 2), n_mels + 2)
    mel_f = mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_bins), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def hann_periodic(win_length: int) -> np.ndarray:
    """scipy.signal.get_window('hann', win_length, fftbins=True), float64."""
    n = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)


def padded_window(n_fft: int, win_length: int) -> np.ndarray:
    """Hann(win_length) centred in n_fft zeros (librosa.util.pad_center)."""
    w = np.zeros(n_fft, dtype=np.float64)
    lpad = (n_fft - win_length) // 2
    w[lpad:lpad + win_length] = hann_periodic(win_length)
    return w


def num_frames(n_samples: int, hop_length: int) -> int:
    """center=True: 1 + n_samples // hop_length frames."""
    return 1 + n_samples // hop_length


def stft_power(y: np.ndarray, n_fft: int, hop_length: int, win_length: int) -> np.ndarray:
    """|STFT|^2 as librosa computes it: [1 + n_fft//2, T], dtype follows y.

    The window is float64, so window * frame and the rFFT run in float64 and
    the result is stored as complex64 for float32 audio; the magnitude is
    taken first and then squared (``np.abs(D) ** 2.0``).
    """
    y = np.asarray(y)
    cdtype = np.complex64 if y.dtype == np.float32 else np.complex128
    pad = n_fft // 2
    yp = np.concatenate([np.zeros(pad, y.dtype), y, np.zeros(pad, y.dtype)])
    T = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(T)[None, :]
    frames = yp[idx]                                        # [n_fft, T]
    win = padded_window(n_fft, win_length)[:, None]
    D = np.fft.rfft(win * frames, axis=0).astype(cdtype)    # float64 FFT, stored narrower
    return np.abs(D) ** 2.0


def melspectrogram(y: np.ndarray, sr: int, n_fft: int, n_mels: int,
                   hop_length: int, win_length: int) -> np.ndarray:
    """[n_mels, T] power mel spectrogram (no log), dtype follows y."""
    S = stft_power(y, n_fft, hop_length, win_length)
    basis = mel_filterbank(sr, n_fft, n_mels)
    return np.einsum("ft,mf->mt", S, basis, optimize=True)


def clip_feature(y: np.ndarray, sr: int, n_fft: int, n_mels: int,
                 hop_length: int, win_length: int) -> np.ndarray:
    """What extract_mel_spectrograms stores per clip: the [n_mels, T] array's
    row-major bytes re-read as [T, n_mels] (feature_extraction.py:49-50)."""
    mel = melspectrogram(y, sr, n_fft, n_mels, hop_length, win_length)
    return np.reshape(mel, (mel.shape[1], mel.shape[0]))


# ---- optional stages the reference does not have (defaults: off) -----------

def power_to_db(S: np.ndarray, amin: float = 1e-10) -> np.ndarray:
    """10 * log10(max(S, amin)); ref=1.0, no top_db clamp."""
    return (10.0 * np.log10(np.maximum(S, amin))).astype(S.dtype)


def normalise(feat: np.ndarray, mean: np.ndarray, std: np.ndarray) -> np.ndarray:
    """Per-mel-bin (x - mean) / std on a [n_mels, T] array."""
    return ((feat - mean[:, None]) / std[:, None]).astype(feat.dtype)
