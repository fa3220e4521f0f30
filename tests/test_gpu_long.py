"""Minutes-long recordings at their full size (needs an MI355X).

The reference lengthens the sequence its models see with ``--clip-length``
(training.py:57-61; chopping at feature_extraction.py:35-38): the GRU of
ShakeRCNN (models.py:114) then runs frames/4 steps, the encoder of
ShakeTransformer (models.py:195-196) attends over frames/4 tokens, the CRNN's
GRU runs frames/2 steps.  These cases run 1500-step / 6000-token sequences
through the HIP path and compare with the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from sound_event_detection_amd import synth

pytestmark = pytest.mark.gpu
ATOL_LOGITS = 1e-3


def _shake(kind):
    from sound_event_detection_amd import models
    if kind == "rcnn":
        m, spec = models.ShakeRCNN(10, 0), synth.shake_rcnn_state_dict_spec(10)
    else:
        m, spec = models.ShakeTransformer(10, 0.0), synth.shake_transformer_state_dict_spec(10)
    sd = synth.synth_shake_state_dict(spec, 0)
    m.load_state_dict(sd)
    m.eval()
    return m.cuda(), sd


@pytest.mark.timeout(600)
def test_shake_rcnn_60s_recording_1500_gru_steps_vs_oracle():
    from oracle import shake
    m, sd = _shake("rcnn")
    x = torch.from_numpy(synth.synth_shake_feats(31, 1, 6000))          # 60 s at 100 feature frames / s
    al = [0.35, 0.55, 0.75]
    want = shake.shake_rcnn_forward(sd, x, al).numpy()
    got = m(x.cuda(), al).cpu().numpy()
    assert got.shape == want.shape == (1, 1500, 10)
    np.testing.assert_allclose(got, want, rtol=0, atol=ATOL_LOGITS)


@pytest.mark.timeout(600)
def test_shake_rcnn_two_80s_recordings_2000_gru_steps_vs_oracle():
    from oracle import shake
    m, sd = _shake("rcnn")
    x = torch.from_numpy(synth.synth_shake_feats(33, 2, 8000))
    al = [0.15, 0.85, 0.5]
    want = shake.shake_rcnn_forward(sd, x, al).numpy()
    got = m(x.cuda(), al).cpu().numpy()
    assert got.shape == want.shape == (2, 2000, 10)
    np.testing.assert_allclose(got, want, rtol=0, atol=ATOL_LOGITS)


@pytest.mark.timeout(900)
def test_shake_transformer_240s_recording_6000_tokens_vs_oracle():
    # 6000 tokens = 47 query tiles x 188 key blocks per head (the last key block partial: 6000 = 187 * 32 + 16)
    from oracle import shake
    m, sd = _shake("transformer")
    x = torch.from_numpy(synth.synth_shake_feats(32, 1, 24000))
    al = [0.35, 0.55, 0.75]
    want = shake.shake_transformer_forward(sd, x, al).numpy()
    got = m(x.cuda(), al).cpu().numpy()
    assert got.shape == want.shape == (1, 6000, 10)
    np.testing.assert_allclose(got, want, rtol=0, atol=ATOL_LOGITS)


@pytest.mark.parametrize("n_clips, T", [(1, 4400), (3, 4100)])
def test_attention_from_packed_key_blocks_agrees_bit_for_bit_with_the_per_workgroup_split(n_clips, T, monkeypatch):
    # From 1024 tokens per clip on, the keys and values of a layer are split into f16 halves and laid out as the
    # attention kernel's LDS planes once, by a pass of their own; below that (and with SEDMI_SHAKE_ATT_PACKED=0) every
    # workgroup does it for itself.  Same operand bits, same products in the same order: identical logits.
    # 4400 -> 1100 tokens = 34 key blocks + 12 keys; 4100 -> 1025 tokens = 32 blocks + 1 key
    m, _ = _shake("transformer")
    x = torch.from_numpy(synth.synth_shake_feats(50 + n_clips, n_clips, T)).cuda()
    al = [0.4, 0.6, 0.2]
    monkeypatch.setenv("SEDMI_SHAKE_ATT_PACKED", "1")
    packed = m(x, al).cpu().numpy()
    monkeypatch.setenv("SEDMI_SHAKE_ATT_PACKED", "0")
    plain = m(x, al).cpu().numpy()
    assert packed.shape == (n_clips, T // 4, 10) and np.isfinite(packed).all()
    assert np.array_equal(packed, plain)


@pytest.mark.parametrize("kind", ["rcnn", "transformer"])
@pytest.mark.parametrize("T", [10, 37, 1001])
def test_shake_frame_counts_are_floored_like_the_reference(kind, T):
    # AvgPool2d floors odd sizes (models.py:109-111, :167,178): 1001 -> 500 -> 250 -> 250; 37 -> 18 -> 9; 10 -> 5 -> 2
    from oracle import shake
    m, sd = _shake(kind)
    x = torch.from_numpy(synth.synth_shake_feats(40 + T, 2, T))
    al = [0.25, 0.5, 0.75]
    fwd = shake.shake_rcnn_forward if kind == "rcnn" else shake.shake_transformer_forward
    want = fwd(sd, x, al).numpy()
    got = m(x.cuda(), al).cpu().numpy()
    assert got.shape == want.shape == (2, T // 4, 10)
    np.testing.assert_allclose(got, want, rtol=0, atol=ATOL_LOGITS)


@pytest.mark.timeout(600)
def test_crnn_predict_files_60s_clips_n_fft_4096(tmp_path):
    """BirdSED-shaped long recordings through the file-level harness: clip_length 60 000 ms (3001 frames, 1500
    GRU steps per clip), n_fft 4096; chopping, tail padding, cutoff and decode as training.py:333-358."""
    from scipy.io import wavfile
    from oracle import crnn as ocrnn, decode as odec, mel as omel
    from sound_event_detection_amd.nnet import CRNN
    from sound_event_detection_amd.predict import predict_files
    sr, n_fft, n_mels, hop, win, clip_ms = 16000, 4096, 128, 320, 640, 60000
    d = os.path.join(str(tmp_path), "BirdSED", "audio")
    os.makedirs(d)
    files = {}
    for name, seconds, seed in (("long.wav", 150.5, 0), ("short.wav", 61.0, 40)):
        n = int(round(seconds * sr))
        y = np.concatenate([synth.synth_clip(seed + k) for k in range(n // 160000 + 1)])[:n]
        pcm = np.clip(np.round(y * 32768.0), -32768, 32767).astype(np.int16)
        wavfile.write(os.path.join(d, name), sr, pcm)
        files["audio/" + name] = pcm.astype(np.float32) / 32768.0
    m = CRNN(dropout=0.5, rnn_layers=2, n_in_channel=1, nclass=10, attention=True, n_RNN_cell=128, activation='glu',
             rnn_type='BGRU', kernel_size=[3] * 7, padding=[1] * 7, stride=[1] * 7,
             nb_filters=[16, 32, 64, 128, 128, 128, 128], pooling=[[2, 2]] + [[1, 2]] * 6, dropout_recurrent=0)
    sd = synth.synth_crnn_state_dict(0, nclass=10)
    m.load_state_dict(sd)
    m.eval()
    m = m.cuda()
    id2cls = {i: f"cls{i}" for i in range(10)}
    names = list(files)
    preds, probs = predict_files(m, names, str(tmp_path) + "/", "BirdSED", clip_ms, n_fft, n_mels, hop, win, id2cls, 40,
                                 decision_threshold=0.5, batch_clips=3, return_probs=True)
    n_clip = sr * clip_ms // 1000
    want_events = []
    for f in names:
        y = files[f]
        dur = len(y) / sr
        k = int(np.ceil(dur / (clip_ms / 1000)))
        assert probs[f].shape == (k * 1500, 10)
        feats = []
        for i in range(k):
            c = y[i * n_clip:(i + 1) * n_clip]
            c = np.concatenate([c, np.zeros(n_clip - len(c), np.float32)])
            feats.append(omel.clip_feature(c, sr, n_fft, n_mels, hop, win))
        assert feats[0].shape == (3001, 128)
        want_p = ocrnn.crnn_forward(sd, torch.from_numpy(np.stack(feats))).numpy().reshape(-1, 10)
        assert np.abs(want_p - probs[f]).max() < 1e-3
        want_events += odec.get_prediction_from_raw_output(probs[f], id2cls, dur, 40, f, 0.5, False)
    assert len(preds) == len(want_events) and len(preds) > 50
    for g, w in zip(preds, want_events):
        assert g[0] == w[0] and g[3] == w[3] and g[1] == w[1] and g[2] == w[2]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_clips,frames", [(1, 40), (16, 24), (17, 16), (33, 16), (64, 12), (128, 12), (190, 8), (200, 8)])
def test_shake_rcnn_recurrence_in_one_launch_agrees_with_per_step_launches(n_clips, frames, monkeypatch):
    """The H = 1024 recurrence runs as one persistent launch for up to 192 clips (groups of 16 or 32 clips, one or two
    clip sets, workgroups exchanging h through device memory every step) and as one launch per step beyond; both
    must give the oracle's logits.  SEDMI_SHAKE_GRU_PERSISTENT=0 selects the per-step kernel,
    SEDMI_SHAKE_GRU_EXCHANGE=flag|tag the persistent kernel's hand-off protocol."""
    from oracle import shake
    m, sd = _shake("rcnn")
    x = torch.from_numpy(synth.synth_shake_feats(60 + n_clips, n_clips, frames))
    al = [0.45, 0.25, 0.65]
    want = shake.shake_rcnn_forward(sd, x, al).numpy()
    monkeypatch.setenv("SEDMI_SHAKE_GRU_PERSISTENT", "1")
    got = {}
    for exchange in ("flag", "tag"):                    # the two hand-off protocols of the persistent kernel
        monkeypatch.setenv("SEDMI_SHAKE_GRU_EXCHANGE", exchange)
        got[exchange] = m(x.cuda(), al).cpu().numpy()
        again = m(x.cuda(), al).cpu().numpy()
        assert np.isfinite(got[exchange]).all()
        assert np.array_equal(got[exchange], again)     # deterministic: no atomics, fixed reduction order
        np.testing.assert_allclose(got[exchange], want, rtol=0, atol=ATOL_LOGITS)
    monkeypatch.delenv("SEDMI_SHAKE_GRU_EXCHANGE")
    monkeypatch.setenv("SEDMI_SHAKE_GRU_PERSISTENT", "0")
    per_step = m(x.cuda(), al).cpu().numpy()
    np.testing.assert_allclose(per_step, want, rtol=0, atol=ATOL_LOGITS)
    np.testing.assert_allclose(got["flag"], per_step, rtol=0, atol=2e-4)
    np.testing.assert_allclose(got["tag"], got["flag"], rtol=0, atol=2e-4)


@pytest.mark.parametrize("exchange", ["tag", "flag"])
def test_recurrence_that_gives_up_is_an_exception_never_nan_or_an_empty_event_list(monkeypatch, exchange):
    """The single-launch recurrence needs all its workgroups resident at once; when a wave's bounded wait for a
    neighbour runs out the kernel drains and the logits are NaN.  With the wait cut to zero polls
    (SEDMI_SHAKE_GRU_SPIN_LIMIT=0: the first poll that finds a producer not ready gives up) the failure is driven
    on purpose: ``forward`` must notice, fall back to one launch per step and return correct logits; the
    pipeline must raise instead of decoding NaNs into "no events"."""
    from sound_event_detection_amd import _lib
    from sound_event_detection_amd.pipeline import SedPipeline
    m, sd = _shake("rcnn")
    x = torch.from_numpy(synth.synth_shake_feats(77, 3, 400)).cuda()
    al = [0.3, 0.5, 0.7]
    want = m(x, al).clone()                                   # the single-launch recurrence, undisturbed
    assert m.recurrence == "auto" and torch.isfinite(want).all()
    monkeypatch.setenv("SEDMI_SHAKE_GRU_EXCHANGE", exchange)
    monkeypatch.setenv("SEDMI_SHAKE_GRU_SPIN_LIMIT", "0")
    pipe = SedPipeline(m, sr=16000, n_fft=2048, hop_length=160, win_length=640)
    out = pipe.run_features(x, alphas=al)
    with pytest.raises(_lib.SedmiRecurrenceError, match="gave up"):
        pipe.events(out, {i: f"c{i}" for i in range(10)}, ["a", "b", "c"])
    assert m.recurrence == "per_step"                         # the next run cannot fail that way
    out = pipe.run_features(x, alphas=al)
    ev = pipe.events(out, {i: f"c{i}" for i in range(10)}, ["a", "b", "c"])
    assert torch.isfinite(out["probs"]).all() and (out["probs"] - want).abs().max() < 1e-4
    # the nn.Module path: warns, switches, reruns, returns finite logits
    m.recurrence = "auto"
    with pytest.warns(RuntimeWarning, match="one launch per step"):
        got = m(x, al)
    assert m.recurrence == "per_step" and (got - want).abs().max() < 1e-4
    monkeypatch.delenv("SEDMI_SHAKE_GRU_SPIN_LIMIT")
    m.recurrence = "auto"
    pipe2 = SedPipeline(m, sr=16000, n_fft=2048, hop_length=160, win_length=640)
    out2 = pipe2.run_features(x, alphas=al)
    assert pipe2.events(out2, {i: f"c{i}" for i in range(10)}, ["a", "b", "c"]) == ev and m.recurrence == "auto"
