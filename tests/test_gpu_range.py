"""Input level of the default (split-f16) arithmetic of the CRNN (needs an MI355X).

Block 0 reads the raw power mel spectrogram, whose magnitude follows the audio
level squared; the f16 halves the default mode splits every operand into top
out at 65504 and lose their low bits far below 1.  The library scales each clip
by an exact power of two in front of the split, so any level must give the
reference's float32 result as long as the checkpoint's BatchNorm statistics
belong to the features being fed."""
import numpy as np
import pytest
import torch

from oracle import crnn as ocrnn, mel as omel
from sound_event_detection_amd import synth

pytestmark = pytest.mark.gpu
PROB_TOL = 1e-3
PRECISIONS = ["split_f16", "f32"]


def _model(sd, precision):
    from sound_event_detection_amd.nnet import CRNN
    m = CRNN(dropout=0.5, rnn_layers=2, n_in_channel=1, nclass=10, attention=True, n_RNN_cell=128, activation='glu',
             rnn_type='BGRU', kernel_size=[3] * 7, padding=[1] * 7, stride=[1] * 7,
             nb_filters=[16, 32, 64, 128, 128, 128, 128], pooling=[[2, 2]] + [[1, 2]] * 6, dropout_recurrent=0)
    m.load_state_dict(sd)
    m.eval()
    m.conv_precision = precision
    return m.cuda()


def _rescaled(sd, power_scale):
    """A checkpoint that training on features ``power_scale`` times larger could have left behind.  Louder
    features: block 0's BatchNorm running statistics follow its input (the conv is linear, its bias is not
    scaled).  Quieter features: the first conv's weights are larger instead (running_var would drop below
    BatchNorm's eps of 1e-3 and the block would stop normalising)."""
    out = {k: v.clone() for k, v in sd.items()}
    if power_scale >= 1.0:
        b = sd["cnn.cnn.conv0.bias"].double()
        mean = sd["cnn.cnn.batchnorm0.running_mean"].double()
        out["cnn.cnn.batchnorm0.running_mean"] = ((mean - b) * power_scale + b).float()
        out["cnn.cnn.batchnorm0.running_var"] = (sd["cnn.cnn.batchnorm0.running_var"].double() * power_scale ** 2).float()
    else:
        out["cnn.cnn.conv0.weight"] = (sd["cnn.cnn.conv0.weight"].double() / power_scale).float()
    return out


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("amp", [1e-3, 1e3, 32768.0])
def test_audio_level_with_matching_checkpoint(amp, precision):
    """Audio scaled by 1e-3, 1e3 and to int16 scale: the power mel moves by 1e-6 .. 1e9."""
    from sound_event_detection_amd import feature_extraction as fe
    y = synth.synth_clips(0, 3) * np.float32(amp)
    sd = _rescaled(synth.synth_crnn_state_dict(0), float(amp) ** 2)
    ref_feats = np.stack([omel.clip_feature(c, 16000, 2048, 128, 320, 640) for c in y])
    feats = fe.mel_features(torch.from_numpy(y).cuda(), 16000, 2048, 128, 320, 640)
    rel = np.abs(feats.cpu().numpy() - ref_feats).max() / np.abs(ref_feats).max()
    assert rel < 1e-4
    want = ocrnn.crnn_forward(sd, torch.from_numpy(ref_feats)).numpy()
    got = _model(sd, precision)(feats).cpu().numpy()
    assert want.std() > 0.05                               # the network is alive at this level
    assert np.abs(got - want).max() < PROB_TOL


@pytest.mark.parametrize("precision", PRECISIONS)
def test_quiet_audio_with_unmatched_checkpoint(precision):
    # features 1e-6 of what the checkpoint saw: block 0 is all bias, any implementation agrees
    feats = np.stack([omel.clip_feature(c, 16000, 2048, 128, 320, 640) for c in synth.synth_clips(0, 2) * np.float32(1e-3)])
    sd = synth.synth_crnn_state_dict(0)
    want = ocrnn.crnn_forward(sd, torch.from_numpy(feats)).numpy()
    got = _model(sd, precision)(torch.from_numpy(feats).cuda()).cpu().numpy()
    assert np.abs(got - want).max() < PROB_TOL


@pytest.mark.parametrize("precision", PRECISIONS)
def test_loud_audio_with_unmatched_checkpoint_stays_finite(precision):
    """Features 1e6 x what the checkpoint saw drive every activation to ~1e6 and every sigmoid gate to a step
    function: float32 rounding decides which side a gate falls on, and the reference's own CPU arithmetic differs
    from its float64 evaluation by 0.7 on the probabilities there (tests/test_oracle_crnn.py), so no parity can
    be asked for.  What must hold: finite probabilities in [0, 1] in both modes (the default mode saturates
    activations at 65504 between the layers, documented in include/sedmi.h)."""
    feats = np.stack([omel.clip_feature(c, 16000, 2048, 128, 320, 640) for c in synth.synth_clips(0, 2) * np.float32(1e3)])
    got = _model(synth.synth_crnn_state_dict(0), precision)(torch.from_numpy(feats).cuda()).cpu().numpy()
    assert np.isfinite(got).all() and got.min() >= 0.0 and got.max() <= 1.0


@pytest.mark.parametrize("precision", PRECISIONS)
def test_mixed_levels_in_one_batch_and_degenerate_clips(precision):
    """Per-clip scaling: a clip's result must not depend on how loud its neighbours are; all-zero clips and a
    clip holding one huge value go through."""
    rng = np.random.Generator(np.random.PCG64(5))
    x = rng.gamma(0.5, 2.0, (6, 64, 128)).astype(np.float32)
    x[1] *= 1e-7
    x[2] *= 3e4
    x[3] = 0.0
    x[4, 10, 5] = 6e4
    sd = synth.synth_crnn_state_dict(2)
    m = _model(sd, precision)
    xd = torch.from_numpy(x).cuda()
    got = m(xd)
    for i in (0, 1, 3, 4, 5):
        assert torch.equal(m(xd[i:i + 1].contiguous()), got[i:i + 1])
    # Reference arithmetic in float64 as the yardstick, and the error the reference's own float32 evaluation (CPU,
    # torch) makes against it, per clip.  Clips 0, 1, 3, 5 are well conditioned (float32 vs float64: ~3e-7) and
    # must meet the 1e-3 tolerance.  Clip 2 (3e4 x louder than the checkpoint expects: every gate a step function)
    # and clip 4 (one 6e4 spike in an O(1) clip: values of ~1e4 next to O(1) ones inside the later blocks' dot
    # products) are ill-conditioned for ANY float32 arithmetic -- the float32 CPU evaluation is 0.43 and 2.1e-3
    # away from float64 on them (pinned in tests/test_oracle_crnn.py) -- so there the HIP path is held to the
    # same yardstick with the float32 CPU path's own error as the allowance.
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    want64 = ocrnn.crnn_forward(sd64, torch.from_numpy(x).double(), explicit_gru=True).numpy()
    cpu32 = ocrnn.crnn_forward(sd, torch.from_numpy(x)).numpy()
    g = got.cpu().numpy()
    for i in range(6):
        cpu_err = float(np.abs(cpu32[i] - want64[i]).max())
        gpu_err = float(np.abs(g[i] - want64[i]).max())
        assert gpu_err <= max(PROB_TOL, 3.0 * cpu_err), (i, gpu_err, cpu_err)
    assert torch.isfinite(got).all()
