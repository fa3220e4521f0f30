"""Mel frontend on MI355X behind the reference's call signatures.

Mirrors /root/reference/feature_extraction.py: ``sampling_rates`` (:9-13),
``extract_mel_spectrograms`` (:16-70) and ``get_features`` (:73-101).  The
arithmetic that the reference delegates to ``librosa.feature.melspectrogram``
(:45-46) runs in the fused HIP kernel of csrc/mel.hip through the C ABI
(``sedmi_melspec_f32``); audio decoding stays on the CPU.

Tensor-level entry points (added underneath the reference's file-level one):

``melspectrogram(y, sr, n_fft, n_mels, hop_length, win_length)``
    y: float32 CUDA tensor [B, N]  ->  [B, n_mels, T], what librosa returns.
``mel_features(...)``
    the same bytes viewed as [B, T, n_mels] — the reference's
    ``np.reshape(mel_spec, (shape[1], shape[0]))`` (:49-50), which is a
    reinterpretation and NOT a transpose.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import weakref

import numpy as np
import torch

from . import _lib

sampling_rates = {
    'TUT': 44100,
    'desed_2022': 16000,
    'BirdSED': 16000
}


class MelPlan:
    """Device tables (window, twiddles, sparse Slaney filterbank) for one
    (sr, n_fft, win_length, hop_length, n_mels) on one GPU."""

    def __init__(self, sr, n_fft, win_length, hop_length, n_mels, device):
        self.key = (int(sr), int(n_fft), int(win_length), int(hop_length), int(n_mels))
        self.device = torch.device(device)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().sedmi_mel_plan_create(C.byref(handle), *self.key),
                       "sedmi_mel_plan_create")
        self.handle = handle
        self.n_mels = int(n_mels)
        self.n_bins = 1 + int(n_fft) // 2
        self.hop = int(hop_length)
        self._fin = weakref.finalize(self, _lib.lib().sedmi_mel_plan_destroy, handle)

    def num_frames(self, n_samples: int) -> int:
        return 1 + int(n_samples) // self.hop

    def filterbank(self) -> np.ndarray:
        out = np.empty((self.n_mels, self.n_bins), dtype=np.float32)
        _lib.check(_lib.lib().sedmi_mel_filterbank(self.handle, out.ctypes.data_as(C.c_void_p)))
        return out


_plans: dict = {}


def get_plan(sr, n_fft, win_length, hop_length, n_mels, device) -> MelPlan:
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (device.index, int(sr), int(n_fft), int(win_length), int(hop_length), int(n_mels))
    plan = _plans.get(key)
    if plan is None:
        plan = _plans[key] = MelPlan(sr, n_fft, win_length, hop_length, n_mels, device)
    return plan


def melspectrogram(y: torch.Tensor, sr: int, n_fft: int, n_mels: int, hop_length: int,
                   win_length: int, *, log: bool = False, mean: torch.Tensor | None = None,
                   std: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """Power mel spectrogram of a batch of clips, [B, n_mels, T] float32.

    ``log`` / ``mean``+``std`` switch on the optional stages the reference
    does not have (10*log10(max(S,1e-10)); per-mel-bin standardisation)."""
    if not y.is_cuda:
        raise _lib.SedmiError("melspectrogram needs a CUDA (ROCm) tensor; there is no CPU path")
    if y.dtype not in (torch.float32, torch.int16):
        raise TypeError("melspectrogram expects float32 audio or int16 PCM (taken as sample / 32768)")
    squeeze = y.dim() == 1
    if squeeze:
        y = y.unsqueeze(0)
    y = y.contiguous()
    B, N = y.shape
    plan = get_plan(sr, n_fft, win_length, hop_length, n_mels, y.device)
    T = plan.num_frames(N)
    if out is None:
        out = torch.empty((B, n_mels, T), dtype=torch.float32, device=y.device)
    elif out.numel() != B * n_mels * T or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("out must be a contiguous float32 tensor of B*n_mels*T elements")
    flags = 0
    istd = None
    if log:
        flags |= _lib.SEDMI_MEL_LOG
    if mean is not None or std is not None:
        if mean is None or std is None:
            raise ValueError("mean and std must be given together")
        flags |= _lib.SEDMI_MEL_NORMALISE
        mean = mean.to(y.device, torch.float32).contiguous()
        istd = (1.0 / std.to(y.device, torch.float32)).contiguous()
    entry = "sedmi_melspec_i16" if y.dtype == torch.int16 else "sedmi_melspec_f32"
    with torch.cuda.device(y.device):
        _lib.check(getattr(_lib.lib(), entry)(plan.handle, _lib.ptr(y), B, N, flags, _lib.ptr(mean),
                                              _lib.ptr(istd), _lib.ptr(out), _lib.stream_ptr(y.device)), entry)
    res = out.view(B, n_mels, T)
    return res[0] if squeeze else res


def mel_features(y: torch.Tensor, sr: int, n_fft: int, n_mels: int, hop_length: int, win_length: int,
                 **kw) -> torch.Tensor:
    """[B, T, n_mels]: the mel-major array's bytes re-read frame-major, as
    feature_extraction.py:49-50 does with np.reshape."""
    m = melspectrogram(y, sr, n_fft, n_mels, hop_length, win_length, **kw)
    if m.dim() == 2:
        return m.reshape(m.shape[1], m.shape[0])
    return m.reshape(m.shape[0], m.shape[2], m.shape[1])


def mel_statistics(mel: torch.Tensor):
    """float64 per-mel-bin (sum, sum of squares, count) of a [B, n_mels, T]
    tensor, for the optional dataset-level normalisation; the three values
    add across ranks."""
    B, M, T = mel.shape
    s = torch.zeros(M, dtype=torch.float64, device=mel.device)
    s2 = torch.zeros(M, dtype=torch.float64, device=mel.device)
    with torch.cuda.device(mel.device):
        _lib.check(_lib.lib().sedmi_mel_stats_f64(_lib.ptr(mel.contiguous()), B, M, T, _lib.ptr(s),
                                                  _lib.ptr(s2), _lib.stream_ptr(mel.device)))
    return s, s2, float(B * T)


# ---------------------------------------------------------------------------
# file-level interface of the reference
# ---------------------------------------------------------------------------

def _load_wav(path: str, sr: int, offset: float, duration: float) -> np.ndarray:
    """Mono float32 samples [offset, offset+duration) seconds of a WAV file,
    at rate ``sr``: the counterpart of librosa.load(path, sr=, offset=,
    duration=) (feature_extraction.py:37), on the CPU like the reference's.
    A file at another rate is resampled after slicing, as librosa does, with a
    polyphase filter (scipy.signal.resample_poly); librosa's default resampler
    (soxr_hq) is not available here, so resampled audio is close to but not
    bit-identical with the reference's -- files at the dataset's own rate
    (16 kHz DESED / BirdSED, 44.1 kHz TUT) are read exactly."""
    from scipy.io import wavfile
    native, data = wavfile.read(path, mmap=True)
    a, b = int(round(offset * native)), int(round(offset * native)) + int(round(duration * native))
    data = np.asarray(data[a:b])
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(1 << (8 * data.dtype.itemsize - 1))
    elif data.dtype == np.uint8:
        data = (data.astype(np.float32) - 128.0) / 128.0
    data = data.astype(np.float32)
    if data.ndim == 2:
        data = data.mean(axis=1)
    if native != sr and len(data):
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(native))
        n_out = int(np.ceil(len(data) * sr / native))          # librosa.resample's output length
        data = resample_poly(data.astype(np.float64), sr // g, native // g)[:n_out].astype(np.float32)
    return data


def _wav_duration(path: str) -> float:
    from scipy.io import wavfile
    native, data = wavfile.read(path, mmap=True)
    return data.shape[0] / float(native)


def extract_mel_spectrograms(audio_files, dataset_location, dataset, clip_length, n_fft, n_mels, hop_length,
                             win_length, save_location=None, get_ast_features=False, *, device=None,
                             audio_loader=None, audio_duration=None, batch_clips=256):
    """Same contract as feature_extraction.py:16-70: one feature per
    ``clip_length`` ms piece of every file; returns {(i, file): ndarray
    [T, n_mels]} or, with ``save_location``, {"<file>_<i>.npy": path}.

    All clips of a call are batched through the GPU ``batch_clips`` at a time.
    ``audio_loader(path, sr, offset_s, duration_s) -> float32[n]`` and
    ``audio_duration(path) -> seconds`` replace librosa.load /
    librosa.get_duration; the defaults read WAV files.  Features are float32
    (the reference yields float64 for a zero-padded last piece, :41-42, and
    casts to float32 before the model, training.py:346)."""
    if get_ast_features:
        raise NotImplementedError("AST embeddings need a pretrained checkpoint fetched by name "
                                  "(feature_extraction.py:21-22); not part of this path")
    sampling_rate = sampling_rates[dataset]
    clip_s = 0.001 * clip_length
    device = torch.device(device if device is not None else "cuda")
    loader = audio_loader or _load_wav
    get_dur = audio_duration or _wav_duration
    n_clip = int(sampling_rate * clip_s)

    keys, pending, features = [], [], {}

    def flush():
        if not pending:
            return
        y = torch.from_numpy(np.stack(pending)).to(device)
        feats = mel_features(y, sampling_rate, n_fft, n_mels, hop_length, win_length).cpu().numpy()
        for (i, audio_file), feat in zip(keys, feats):
            if save_location is not None:
                name = audio_file.replace('/', '_slash_') + '_' + str(i) + '.npy'
                np.save(save_location + '/' + name, feat)
                features[name] = save_location + '/' + name
            else:
                features[(i, audio_file)] = feat
        keys.clear()
        pending.clear()

    for audio_file in audio_files:
        file_path = dataset_location + '/' + dataset + '/' + audio_file
        audio_file = audio_file.replace('\\', '/')
        audio_length = get_dur(file_path)
        for i in range(int(np.ceil(audio_length / clip_s))):
            clip = np.asarray(loader(file_path, sampling_rate, i * clip_s, clip_s), dtype=np.float32)
            if len(clip) < n_clip:                       # zero-pad the last piece (:41-42)
                clip = np.concatenate([clip, np.zeros(n_clip - len(clip), np.float32)])
            keys.append((i, audio_file))
            pending.append(clip[:n_clip])          # a loader may round a fractional sample count up
            if len(pending) >= batch_clips:
                flush()
    flush()
    return features


def get_features(name, dataset_location, dataset, fold, clip_length, n_fft, n_mels, hop_length, win_length,
                 audio_file_folder=None, audio_files=None, get_ast_features=False, **kw):
    """Feature cache of feature_extraction.py:73-101: directory name and file
    naming are the reference's, so caches are interchangeable."""
    dir_name = name + f'_{fold}_{clip_length}_{n_fft}_{n_mels}_{hop_length}_{win_length}'
    dir_path = dataset_location + dataset + '/' + dir_name
    if os.path.isdir(dir_path):
        features = {}
        for filename in glob.glob(dir_path + '/*.npy'):
            features[filename.replace('\\', '/').split('/')[-1]] = filename
        return features
    os.makedirs(dir_path, exist_ok=True)
    if audio_file_folder is not None:
        prefix = dataset_location + dataset + '/'
        audio_files = [f.replace(prefix, '') for f in glob.glob(prefix + audio_file_folder)]
    elif audio_files is None:
        audio_files = []
    return extract_mel_spectrograms(audio_files, dataset_location, dataset, clip_length, n_fft, n_mels,
                                    hop_length, win_length, save_location=dir_path,
                                    get_ast_features=get_ast_features, **kw)
