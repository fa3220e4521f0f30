"""HIP mel frontend vs the CPU oracle (needs an MI355X)."""
import numpy as np
import pytest
import torch

from oracle import mel as omel
from sound_event_detection_amd import synth

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4      # north-star tolerance for the (log-)mel stage


def _rel_err(got, ref):
    return np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)


@pytest.mark.parametrize("cfg", [
    (16000, 2048, 128, 320, 640, 160000, 3),     # DESED / BirdSED configuration
    (44100, 2048, 128, 882, 1764, 441000, 2),    # TUT configuration
    (16000, 1024, 64, 160, 400, 16000, 2),       # smaller transform
    (16000, 4096, 128, 640, 1280, 80000, 2),     # larger transform
    (16000, 2048, 40, 320, 640, 5000, 2),        # clip shorter than a few hops, ragged tile
    (16000, 2048, 128, 321, 641, 15999, 2),      # odd hop, odd length: sample pairs straddle both clip edges
    (22050, 1024, 64, 441, 1023, 22051, 3),
])
def test_mel_matches_oracle(cfg):
    from sound_event_detection_amd import feature_extraction as fe
    sr, n_fft, n_mels, hop, win, n, B = cfg
    y = synth.synth_clips(10, B, n, sr)
    ref = np.stack([omel.melspectrogram(c, sr, n_fft, n_mels, hop, win) for c in y])
    got = fe.melspectrogram(torch.from_numpy(y).cuda(), sr, n_fft, n_mels, hop, win).cpu().numpy()
    assert got.shape == ref.shape and got.dtype == np.float32
    err = _rel_err(got, ref)
    assert err.max() < REL_TOL, f"max rel err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"


@pytest.mark.parametrize("cfg", [
    (16000, 2048, 128, 320, 640, 160000, 3),
    (16000, 1024, 64, 160, 400, 16000, 2),
    (16000, 4096, 128, 640, 1280, 80000, 2),
    (16000, 2048, 128, 321, 641, 15999, 2),      # odd hop, odd length: 2-byte samples straddle both clip edges
])
def test_int16_pcm_input_is_bit_identical_to_the_converted_float_input(cfg):
    """sample / 32768 is exact in float32 (what a PCM16 WAV becomes in
    librosa.load, feature_extraction.py:37): the int16 entry point must return
    the float32 entry point's output for the converted samples, bit for bit,
    full-scale samples included."""
    from sound_event_detection_amd import feature_extraction as fe
    sr, n_fft, n_mels, hop, win, n, B = cfg
    rng = np.random.Generator(np.random.PCG64(n))
    pcm = rng.integers(-32768, 32768, (B, n), dtype=np.int16)
    pcm[0, :4] = [-32768, 32767, 0, -1]
    as_float = torch.from_numpy(pcm.astype(np.float32) / 32768.0).cuda()
    want = fe.melspectrogram(as_float, sr, n_fft, n_mels, hop, win)
    got = fe.melspectrogram(torch.from_numpy(pcm).cuda(), sr, n_fft, n_mels, hop, win)
    assert torch.equal(got, want)
    ref = omel.melspectrogram(pcm[1].astype(np.float32) / 32768.0, sr, n_fft, n_mels, hop, win)
    assert _rel_err(got[1].cpu().numpy(), ref).max() < REL_TOL


def test_filterbank_matches_oracle():
    from sound_event_detection_amd import feature_extraction as fe
    plan = fe.get_plan(16000, 2048, 640, 320, 128, "cuda")
    assert np.abs(plan.filterbank() - omel.mel_filterbank(16000, 2048, 128)).max() < 4e-9
    plan = fe.get_plan(44100, 2048, 1764, 882, 128, "cuda")
    assert np.abs(plan.filterbank() - omel.mel_filterbank(44100, 2048, 128)).max() < 4e-9


def test_feature_is_reshape_not_transpose():
    from sound_event_detection_amd import feature_extraction as fe
    y = synth.synth_clips(0, 2)
    yd = torch.from_numpy(y).cuda()
    m = fe.melspectrogram(yd, 16000, 2048, 128, 320, 640)
    f = fe.mel_features(yd, 16000, 2048, 128, 320, 640)
    assert f.shape == (2, 501, 128)
    assert torch.equal(f.reshape(2, -1), m.reshape(2, -1))
    ref = omel.clip_feature(y[1], 16000, 2048, 128, 320, 640)
    assert _rel_err(f[1].cpu().numpy(), ref).max() < REL_TOL


def test_silence_and_impulse():
    from sound_event_detection_amd import feature_extraction as fe
    y = np.zeros((2, 16000), np.float32)
    y[1, 8000] = 1.0
    got = fe.melspectrogram(torch.from_numpy(y).cuda(), 16000, 2048, 128, 320, 640).cpu().numpy()
    assert np.all(got[0] == 0)
    ref = omel.melspectrogram(y[1], 16000, 2048, 128, 320, 640)
    assert np.abs(got[1] - ref).max() <= 1e-4 * np.abs(ref).max()


def test_optional_log_and_normalise():
    from sound_event_detection_amd import feature_extraction as fe
    y = synth.synth_clips(5, 2)
    yd = torch.from_numpy(y).cuda()
    ref = np.stack([omel.melspectrogram(c, 16000, 2048, 128, 320, 640) for c in y])
    db = fe.melspectrogram(yd, 16000, 2048, 128, 320, 640, log=True).cpu().numpy()
    assert np.abs(db - omel.power_to_db(ref)).max() < 1e-3       # 1e-4 relative on S = 4.3e-4 dB
    mean = torch.linspace(-1, 1, 128)
    std = torch.linspace(0.5, 2, 128)
    nz = fe.melspectrogram(yd, 16000, 2048, 128, 320, 640, log=True, mean=mean, std=std).cpu().numpy()
    want = (omel.power_to_db(ref) - mean.numpy()[None, :, None]) / std.numpy()[None, :, None]
    assert np.abs(nz - want).max() < 2e-3
    s, s2, n = fe.mel_statistics(torch.from_numpy(ref).cuda())
    assert np.allclose(s.cpu().numpy(), ref.astype(np.float64).sum((0, 2)), rtol=1e-12)
    assert np.allclose(s2.cpu().numpy(), (ref.astype(np.float64) ** 2).sum((0, 2)), rtol=1e-12)
    assert n == 2 * 501


def test_large_batch_roundtrip_property():
    """Linearity at a batch that spans many workgroups: mel(a*y) = a^2 mel(y)."""
    from sound_event_detection_amd import feature_extraction as fe
    g = torch.Generator(device="cuda").manual_seed(1)
    y = torch.randn(64, 160000, device="cuda", generator=g) * 0.1
    m1 = fe.melspectrogram(y, 16000, 2048, 128, 320, 640)
    m2 = fe.melspectrogram(y * 2.0, 16000, 2048, 128, 320, 640)
    assert torch.allclose(m2, 4.0 * m1, rtol=1e-6, atol=0)
    # identical clips at different batch positions give identical rows
    y[63] = y[0]
    m3 = fe.melspectrogram(y, 16000, 2048, 128, 320, 640)
    assert torch.equal(m3[63], m3[0])
