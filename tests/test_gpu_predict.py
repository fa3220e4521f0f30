"""File-level harness (training.py:328-365 counterpart): chopping, tail padding, per-file decode, TSV."""
import os

import numpy as np
import pytest
import torch

from sound_event_detection_amd import synth

pytestmark = pytest.mark.gpu
SR, N_FFT, N_MELS, HOP, WIN = 16000, 2048, 128, 320, 640


@pytest.fixture(scope="module")
def model():
    from sound_event_detection_amd.nnet import CRNN
    m = CRNN(dropout=0.5, rnn_layers=2, n_in_channel=1, nclass=10, attention=True, n_RNN_cell=128, activation='glu',
             rnn_type='BGRU', kernel_size=[3] * 7, padding=[1] * 7, stride=[1] * 7,
             nb_filters=[16, 32, 64, 128, 128, 128, 128], pooling=[[2, 2]] + [[1, 2]] * 6, dropout_recurrent=0)
    m.load_state_dict(synth.synth_crnn_state_dict(0, nclass=10))
    m.eval()
    return m.cuda()


def _write_files(root):
    from scipy.io import wavfile
    d = os.path.join(root, "desed_2022", "audio")
    os.makedirs(d)
    files = {}
    for name, seconds, seed in (("a.wav", 25.3, 0), ("b.wav", 10.0, 3), ("c.wav", 3.7, 5), ("d.wav", 20.02, 6)):
        n = int(round(seconds * SR))
        y = np.concatenate([synth.synth_clip(seed + k) for k in range(n // 160000 + 1)])[:n]
        pcm = np.clip(np.round(y * 32768.0), -32768, 32767).astype(np.int16)
        wavfile.write(os.path.join(d, name), SR, pcm)
        files["audio/" + name] = pcm.astype(np.float32) / 32768.0
    return files


def test_predict_files_matches_reference_procedure(model, tmp_path):
    from oracle import crnn as ocrnn, decode as odec, mel as omel
    from sound_event_detection_amd.predict import predict_files, write_predictions_tsv
    files = _write_files(str(tmp_path))
    id2cls = {i: f"cls{i}" for i in range(10)}
    loc = str(tmp_path) + "/"
    names = list(files)
    preds, probs = predict_files(model, names, loc, "desed_2022", 10000, N_FFT, N_MELS, HOP, WIN, id2cls, 40,
                                 decision_threshold=0.5, apply_sigmoid=False, batch_clips=4, return_probs=True)
    sd = synth.synth_crnn_state_dict(0, nclass=10)
    want_all = []
    for f in names:
        y = files[f]
        dur = len(y) / SR
        n_clips = int(np.ceil(dur / 10.0))
        assert probs[f].shape == (n_clips * 250, 10)
        feats = []
        for i in range(n_clips):
            c = y[i * 160000:(i + 1) * 160000]
            c = np.concatenate([c, np.zeros(160000 - len(c), np.float32)])
            feats.append(omel.clip_feature(c, SR, N_FFT, N_MELS, HOP, WIN))
        want_p = ocrnn.crnn_forward(sd, torch.from_numpy(np.stack(feats))).numpy().reshape(-1, 10)
        assert np.abs(want_p - probs[f]).max() < 1e-3
        # events: the reference decoder on the very probabilities the GPU produced, so that frames
        # within float32 noise of the threshold cannot flip the comparison
        want_all += odec.get_prediction_from_raw_output(probs[f], id2cls, dur, 40, f, 0.5, False)
    assert len(preds) == len(want_all) and len(preds) > 20
    for g, w in zip(preds, want_all):
        assert g[0] == w[0] and g[3] == w[3]
        assert isinstance(g[1], np.float64) and g[1] == w[1] and g[2] == w[2]
    out = tmp_path / "test_predictions.tsv"
    write_predictions_tsv(preds, str(out))
    lines = out.read_text().splitlines()
    assert lines[0] == "filename\tonset\toffset\tevent_label"
    assert len(lines) == len(preds) + 1
    f0, on, off, lab = lines[1].split("\t")
    assert f0 == preds[0][0] and float(on) == preds[0][1] and float(off) == preds[0][2] and lab == preds[0][3]


def test_batching_does_not_change_events(model, tmp_path):
    from sound_event_detection_amd.predict import predict_files
    files = _write_files(str(tmp_path))
    id2cls = {i: f"cls{i}" for i in range(10)}
    args = (model, list(files), str(tmp_path) + "/", "desed_2022", 10000, N_FFT, N_MELS, HOP, WIN, id2cls, 40)
    a = predict_files(*args, batch_clips=1)
    b = predict_files(*args, batch_clips=64)
    assert a == b


def test_extract_mel_spectrograms_contract_and_cache(tmp_path):
    # feature_extraction.py:16-70 (keys, tail padding, reshape-not-transpose) and :73-101 (cache layout)
    from oracle import mel as omel
    from sound_event_detection_amd.feature_extraction import extract_mel_spectrograms, get_features
    files = _write_files(str(tmp_path))
    loc = str(tmp_path) + "/"
    names = ["audio/a.wav", "audio/c.wav"]
    feats = extract_mel_spectrograms(names, str(tmp_path), "desed_2022", 10000, N_FFT, N_MELS, HOP, WIN, batch_clips=3)
    assert list(feats) == [(0, "audio/a.wav"), (1, "audio/a.wav"), (2, "audio/a.wav"), (0, "audio/c.wav")]
    for (i, f), got in feats.items():
        c = files[f][i * 160000:(i + 1) * 160000]
        c = np.concatenate([c, np.zeros(160000 - len(c), np.float32)])
        want = omel.clip_feature(c, SR, N_FFT, N_MELS, HOP, WIN)
        assert got.shape == want.shape == (501, 128) and got.dtype == np.float32
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
    cached = get_features("mel", loc, "desed_2022", 1, 10000, N_FFT, N_MELS, HOP, WIN, audio_files=names)
    d = os.path.join(str(tmp_path), "desed_2022", "mel_1_10000_2048_128_320_640")
    assert sorted(cached) == ["audio_slash_a.wav_0.npy", "audio_slash_a.wav_1.npy", "audio_slash_a.wav_2.npy",
                              "audio_slash_c.wav_0.npy"]
    assert all(os.path.dirname(v) == d for v in cached.values())
    assert np.array_equal(np.load(cached["audio_slash_a.wav_1.npy"]), feats[(1, "audio/a.wav")])
    again = get_features("mel", loc, "desed_2022", 1, 10000, N_FFT, N_MELS, HOP, WIN, audio_files=[])
    assert sorted(again) == sorted(cached)
