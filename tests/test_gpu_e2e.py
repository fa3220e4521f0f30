"""End-to-end acceptance (needs an MI355X): the same synthetic clips through the CPU path (oracle mel -> model ->
the reference's decoder) and through the HIP path (SedPipeline: mel -> model -> decode in one pass), compared as
the event lists a user gets.

Rule (oracle/acceptance.py): every frame whose thresholded bit differs must lie within 1e-3 of the threshold on
the CPU side, and with those frames masked the two event lists are identical, float64 times bit for bit.
Follows evaluation.py:49-80 and the prediction loop of training.py:345-358."""
import numpy as np
import pytest
import torch

from oracle import acceptance, crnn as ocrnn, decode as odec, mel as omel, shake as oshake
from sound_event_detection_amd import synth

pytestmark = pytest.mark.gpu
SR, N_FFT, HOP, WIN = 16000, 2048, 320, 640
THR, TOL = 0.5, 1e-3


def _same_events(a, b):
    assert len(a) == len(b)
    for g, w in zip(a, b):
        assert g[0] == w[0] and g[3] == w[3]
        assert np.float64(g[1]).tobytes() == np.float64(w[1]).tobytes()
        assert np.float64(g[2]).tobytes() == np.float64(w[2]).tobytes()


def _check(p_cpu, p_gpu, gpu_events, counts, frames_per_clip, names, durations, id2cls, apply_sigmoid):
    from sound_event_detection_amd import evaluation as ev
    n_cls = p_cpu.shape[-1]
    p_cpu, p_gpu = p_cpu.reshape(-1, n_cls), p_gpu.reshape(-1, n_cls)
    n_diff, n_bad, near = acceptance.frame_differences(p_cpu, p_gpu, THR, TOL, apply_sigmoid)
    assert n_bad == 0, f"{n_bad} of {n_diff} differing frames lie further than {TOL} from the threshold"
    # the events the pipeline returned are the decode of its own frame predictions ...
    own = acceptance.cpu_events(p_gpu, counts, frames_per_clip, names, id2cls, durations, 40, THR, apply_sigmoid)
    if not apply_sigmoid:           # (with a sigmoid in the decoder, ulp-level differences of expf may move a tie)
        _same_events(gpu_events, own)
    # ... and with the frames within tolerance of the threshold masked, both sides give the same list
    m_cpu = acceptance.masked(p_cpu, near, THR, apply_sigmoid)
    m_gpu = acceptance.masked(p_gpu, near, THR, apply_sigmoid)
    want = acceptance.cpu_events(m_cpu, counts, frames_per_clip, names, id2cls, durations, 40, THR, apply_sigmoid)
    got, o = [], 0
    for name, n, dur in zip(names, counts, durations):
        rows = n * frames_per_clip
        got += ev.get_prediction_from_raw_output(torch.from_numpy(m_gpu[o:o + rows]).cuda(), id2cls, dur, 40, name,
                                                 THR, apply_sigmoid)
        o += rows
    _same_events(got, want)
    assert len(want) > 0
    if n_diff == 0:
        _same_events(gpu_events, acceptance.cpu_events(p_cpu, counts, frames_per_clip, names, id2cls, durations, 40,
                                                       THR, apply_sigmoid))
    return n_diff, int(near.sum()), len(want)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("precision", ["split_f16", "f32"])
def test_crnn_events_of_64_clips_match_the_cpu_path(precision):
    from sound_event_detection_amd.nnet import CRNN
    from sound_event_detection_amd.pipeline import SedPipeline
    n = 64
    y = synth.synth_clips(100, n)
    sd = synth.synth_crnn_state_dict(0)
    feats = np.stack([omel.clip_feature(c, SR, N_FFT, 128, HOP, WIN) for c in y])
    p_cpu = torch.cat([ocrnn.crnn_forward(sd, torch.from_numpy(feats[i:i + 16])) for i in range(0, n, 16)]).numpy()
    m = CRNN(dropout=0.5, rnn_layers=2, n_in_channel=1, nclass=10, attention=True, n_RNN_cell=128, activation='glu',
             rnn_type='BGRU', kernel_size=[3] * 7, padding=[1] * 7, stride=[1] * 7,
             nb_filters=[16, 32, 64, 128, 128, 128, 128], pooling=[[2, 2]] + [[1, 2]] * 6, dropout_recurrent=0)
    m.load_state_dict(sd)
    m.eval()
    m.conv_precision = precision
    m = m.cuda()
    # recordings of 1, 2, 3, ... clips; the last clip of every recording is cut short by the duration
    counts, left = [], n
    while left:
        counts.append(min(left, 1 + len(counts) % 5))
        left -= counts[-1]
    names = [f"rec{i}.wav" for i in range(len(counts))]
    durations = [10.0 * c - (0.0 if i % 2 else 3.21) for i, c in enumerate(counts)]
    cutoffs = [odec.cutoff_frames(d, 40) for d in durations]
    id2cls = {i: f"cls{i}" for i in range(10)}
    pipe = SedPipeline(m, sr=SR, n_fft=N_FFT, hop_length=HOP, win_length=WIN)
    out = pipe.run(torch.from_numpy(y).cuda(), clips_per_recording=counts, cutoffs=cutoffs)
    gpu_events = pipe.events(out, id2cls, names)
    p_gpu = out["probs"].cpu().numpy()
    assert np.abs(p_gpu - p_cpu).max() < 1e-3
    n_diff, n_near, n_events = _check(p_cpu, p_gpu, gpu_events, counts, 250, names, durations, id2cls, False)
    print(f"{precision}: {n_diff} differing frames, {n_near} frames within {TOL} of the threshold, {n_events} events")


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["rcnn", "transformer", "basic"])
def test_other_models_events_from_audio_match_the_cpu_path(kind):
    """The other --model choices (training.py:200-205) from audio: mel at the model's feature width (768 mel bins
    at hop 160 -> 1001 frames -> 250 outputs for the Shake models; 64 at hop 320 for BasicRCNN), logits out,
    sigmoid inside the decoder (training.py:358)."""
    from sound_event_detection_amd import models
    from sound_event_detection_amd.pipeline import SedPipeline
    n = 4
    y = synth.synth_clips(300, n)
    al = [0.3, 0.5, 0.7]
    if kind == "basic":
        n_mels, hop, pool = 64, HOP, 2
        sd = synth.synth_shake_state_dict(synth.basic_rcnn_state_dict_spec(10, 64), 0)
        m = models.BasicRCNN(10, 64)
    elif kind == "rcnn":
        n_mels, hop, pool = 768, 160, 4
        sd = synth.synth_shake_state_dict(synth.shake_rcnn_state_dict_spec(10), 0)
        m = models.ShakeRCNN(10, 0)
    else:
        n_mels, hop, pool = 768, 160, 4
        sd = synth.synth_shake_state_dict(synth.shake_transformer_state_dict_spec(10), 0)
        m = models.ShakeTransformer(10, 0.0)
    # the power mel of these clips is O(10) with bursts of several hundred; the synthetic BatchNorm statistics of
    # these models expect O(1) features, so feed a quieter recording (amplitude / 4)
    y = (y * np.float32(0.25)).astype(np.float32)
    feats = np.stack([omel.clip_feature(c, SR, N_FFT, n_mels, hop, WIN) for c in y])
    x = torch.from_numpy(feats)
    if kind == "basic":
        p_cpu = oshake.basic_rcnn_forward(sd, x).numpy()
    elif kind == "rcnn":
        p_cpu = oshake.shake_rcnn_forward(sd, x, al).numpy()
    else:
        p_cpu = oshake.shake_transformer_forward(sd, x, al).numpy()
    m.load_state_dict(sd)
    m.eval()
    m = m.cuda()
    pipe = SedPipeline(m, sr=SR, n_fft=N_FFT, n_mels=n_mels, hop_length=hop, win_length=WIN)
    assert pipe.apply_sigmoid
    counts, names, durations = [1, 3], ["a.wav", "b.wav"], [10.0, 27.3]
    cutoffs = [odec.cutoff_frames(d, 40) for d in durations]
    id2cls = {i: f"cls{i}" for i in range(10)}
    out = pipe.run(torch.from_numpy(y).cuda(), clips_per_recording=counts, cutoffs=cutoffs, alphas=al)
    mel_rel = np.abs(out["mel"].cpu().numpy().reshape(feats.shape) - feats).max() / np.abs(feats).max()
    assert mel_rel < 1e-4
    gpu_events = pipe.events(out, id2cls, names)
    p_gpu = out["probs"].cpu().numpy()
    frames = feats.shape[1] // pool
    assert p_gpu.shape == p_cpu.shape == (n, frames, 10)
    assert np.abs(p_gpu - p_cpu).max() < 4e-3          # logits; probabilities move by at most a quarter of this
    n_diff, n_near, n_events = _check(p_cpu, p_gpu, gpu_events, counts, frames, names, durations, id2cls, True)
    print(f"{kind}: {n_diff} differing frames, {n_near} frames within {TOL} of the threshold, {n_events} events")


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dataset,sr,win,hop,n_cls", [("BirdSED", 16000, 640, 320, 92), ("BirdSED-1", 16000, 640, 320, 1),
                                                       ("TUT", 44100, 1764, 882, 6)])
def test_crnn_events_for_each_dataset_shape_match_the_cpu_path(dataset, sr, win, hop, n_cls):
    """The three dataset configurations of the reference from audio to events: BirdSED with its 92 classes
    (utils.py:13; the head runs as a kernel of its own beyond 16 classes) and as a single class (utils.py:54-55;
    the reference's squeeze() then hands the decoder a 1-D array, evaluation.py:62-63), and TUT at 44.1 kHz with 6
    classes (feature_extraction.py:10, utils.py:12: win 1764, hop 882 from training.py:160-162 -> 501 frames)."""
    from sound_event_detection_amd.nnet import CRNN
    from sound_event_detection_amd.pipeline import SedPipeline
    n = 6
    n_samples = 10 * sr
    # the synthetic checkpoint's BatchNorm statistics belong to power mels of a 640-sample window: a longer window
    # collects proportionally more of a tone's energy, so the recording is made that much quieter
    y = np.stack([synth.synth_clip(500 + i, n_samples, sr) for i in range(n)]) * np.float32(640.0 / win)
    sd = synth.synth_crnn_state_dict(7, nclass=n_cls)
    feats = np.stack([omel.clip_feature(c, sr, N_FFT, 128, hop, win) for c in y])
    assert feats.shape == (n, 501, 128)
    p_cpu = ocrnn.crnn_forward(sd, torch.from_numpy(feats)).numpy()
    m = CRNN(dropout=0.5, rnn_layers=2, n_in_channel=1, nclass=n_cls, attention=True, n_RNN_cell=128, activation='glu',
             rnn_type='BGRU', kernel_size=[3] * 7, padding=[1] * 7, stride=[1] * 7,
             nb_filters=[16, 32, 64, 128, 128, 128, 128], pooling=[[2, 2]] + [[1, 2]] * 6, dropout_recurrent=0)
    m.load_state_dict(sd)
    m.eval()
    m = m.cuda()
    counts, names, durations = [2, 1, 3], ["a.wav", "b.wav", "c.wav"], [17.5, 10.0, 30.0]
    cutoffs = [odec.cutoff_frames(d, 40) for d in durations]
    id2cls = {i: f"cls{i}" for i in range(n_cls)}
    pipe = SedPipeline(m, sr=sr, n_fft=N_FFT, hop_length=hop, win_length=win)
    out = pipe.run(torch.from_numpy(y).cuda(), clips_per_recording=counts, cutoffs=cutoffs)
    gpu_events = pipe.events(out, id2cls, names)
    p_gpu = out["probs"].cpu().numpy()
    assert p_gpu.shape == p_cpu.shape == (n, 250, n_cls)
    assert np.abs(p_gpu - p_cpu).max() < 1e-3
    n_diff, n_near, n_events = _check(p_cpu, p_gpu, gpu_events, counts, 250, names, durations, id2cls, False)
    print(f"{dataset}: {n_diff} differing frames, {n_near} frames within {TOL} of the threshold, {n_events} events")
    if n_cls == 1:
        # what the reference's prediction loop hands over after squeeze() (training.py:348): a 1-D array per file
        from sound_event_detection_amd import evaluation as ev
        flat = torch.from_numpy(acceptance.masked(p_cpu.reshape(-1, 1), np.abs(p_cpu.reshape(-1, 1) - THR) <= TOL, THR, False))
        one_d = ev.get_prediction_from_raw_output(flat[:500, 0].cuda(), id2cls, 17.5, 40, "a.wav", THR, False)
        two_d = odec.get_prediction_from_raw_output(flat[:500].numpy(), id2cls, 17.5, 40, "a.wav", THR, False)
        _same_events(one_d, two_d)
