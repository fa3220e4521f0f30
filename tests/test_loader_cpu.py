"""The WAV loader behind extract_mel_spectrograms / predict_files (the stand-in for librosa.load at
feature_extraction.py:37): exact at the file's own rate, polyphase-resampled otherwise."""
import os

import numpy as np
from scipy.io import wavfile

from sound_event_detection_amd import feature_extraction as fe


def _tone(path, sr, seconds, f0=440.0):
    t = np.arange(int(sr * seconds)) / sr
    pcm = np.round(0.3 * np.sin(2 * np.pi * f0 * t) * 32767).astype(np.int16)
    wavfile.write(path, sr, pcm)
    return pcm.astype(np.float32) / 32768.0


def test_native_rate_is_read_exactly_with_offset_and_duration(tmp_path):
    p = os.path.join(str(tmp_path), "a.wav")
    y = _tone(p, 16000, 2.5)
    got = fe._load_wav(p, 16000, 0.5, 1.25)
    assert got.dtype == np.float32 and np.array_equal(got, y[8000:28000])
    assert fe._wav_duration(p) == 2.5
    tail = fe._load_wav(p, 16000, 2.0, 10.0)              # a last piece shorter than asked for: the caller pads it
    assert np.array_equal(tail, y[32000:])


def test_other_rates_are_resampled_to_the_dataset_rate(tmp_path):
    p = os.path.join(str(tmp_path), "b.wav")
    _tone(p, 44100, 3.0)
    got = fe._load_wav(p, 16000, 0.5, 2.0)
    assert got.dtype == np.float32 and len(got) == 32000   # ceil(88200 * 16000 / 44100), librosa.resample's length
    t = np.arange(len(got)) / 16000 + 0.5
    want = 0.3 * np.sin(2 * np.pi * 440.0 * t)
    assert np.abs(got - want)[200:-200].max() < 1e-3       # away from the segment's edges


def test_resampled_piece_lengths_and_the_zero_padding_of_a_short_last_clip(tmp_path):
    """A 44.1 kHz file read at 16 kHz: every piece has ceil(n_native * 16000 / 44100) samples (librosa.resample's
    rule, applied per piece because the reference calls librosa.load per clip, feature_extraction.py:37), and the
    last, shorter piece is what the caller zero-pads to the clip length (feature_extraction.py:41-42)."""
    p = os.path.join(str(tmp_path), "c.wav")
    _tone(p, 44100, 2.3)                                    # 101430 native samples
    first = fe._load_wav(p, 16000, 0.0, 1.0)
    last = fe._load_wav(p, 16000, 2.0, 1.0)                 # 0.3 s left: 13230 native samples
    assert len(first) == 16000 and len(last) == int(np.ceil(13230 * 16000 / 44100)) == 4800
    odd = fe._load_wav(p, 16000, 0.0, 1000 / 44100)         # 1000 native samples -> ceil(362.8)
    assert len(odd) == 363
    n_clip = 16000
    padded = np.concatenate([last, np.zeros(n_clip - len(last), np.float32)])
    assert padded.dtype == np.float32 and len(padded) == n_clip and not padded[len(last):].any()
    assert np.abs(padded[:len(last)]).max() > 0.2           # the audio is still there
