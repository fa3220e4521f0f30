"""HIP CRNN vs outputs of the reference's CNN / BidirectionalGRU classes
(tests/golden/crnn_*.npz) and vs the CPU oracle (needs an MI355X)."""
import os

import numpy as np
import pytest
import torch

from oracle import crnn as ocrnn
from sound_event_detection_amd import synth

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-3          # north-star tolerance on frame probabilities (absolute)


PRECISIONS = ["split_f16", "f32"]


def _model(seed, ncls, precision="split_f16"):
    from sound_event_detection_amd.nnet import CRNN
    # the constructor call of training.py:207-212
    m = CRNN(dropout=0.5, rnn_layers=2, n_in_channel=1, nclass=ncls, attention=True, n_RNN_cell=128,
             activation='glu', rnn_type='BGRU', kernel_size=[3] * 7, padding=[1] * 7, stride=[1] * 7,
             nb_filters=[16, 32, 64, 128, 128, 128, 128], pooling=[[2, 2]] + [[1, 2]] * 6, dropout_recurrent=0)
    sd = synth.synth_crnn_state_dict(seed, nclass=ncls)
    missing = m.load_state_dict(sd)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(m.state_dict().keys()) == list(sd.keys())          # the reference's 83 keys, in order
    m.eval()
    m.conv_precision = precision
    return m.cuda(), sd


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("fname,seed,ncls", [("crnn_seed0.npz", 0, 10), ("crnn_seed3_small.npz", 3, 6)])
def test_stages_match_reference_golden(golden_dir, fname, seed, ncls, precision):
    g = np.load(os.path.join(golden_dir, fname))
    m, _ = _model(seed, ncls, precision)
    x = torch.from_numpy(g["feats"]).cuda()
    seq, rec, probs = m.forward_stages(x)
    ref_seq = np.transpose(g["cnn_out"][..., 0], (0, 2, 1))        # [B,C,T',1] -> [B,T',C]
    scale = np.abs(ref_seq).max()
    assert np.abs(seq.cpu().numpy() - ref_seq).max() < (1e-5 if precision == "f32" else 3e-5) * scale
    assert np.abs(rec.cpu().numpy() - g["gru_out"]).max() < 1e-4
    assert np.abs(probs.cpu().numpy() - g["probs"]).max() < PROB_TOL
    full = m(x)
    # the staged head is the float32 kernel; inside the full forward (default precision, <= 16 classes) the head's
    # products ride in the last GRU layer in split-f16 arithmetic
    assert torch.equal(full, probs) if precision == "f32" else (full - probs).abs().max() < 1e-6
    assert np.abs(full.cpu().numpy() - g["probs"]).max() < 1e-4     # well inside the 1e-3 budget


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("tag,seed,ncls", [("seed0", 0, 10), ("seed3_small", 3, 6), ("92cls", 5, 92), ("1cls", 6, 1)])
def test_forward_matches_reference_crnn_forward(golden_dir, tag, seed, ncls, precision):
    """Against outputs of the reference's CRNN.forward itself (nnet/CRNN.py:191-268), not only its
    sub-modules: tests/golden/crnn_forward.npz."""
    g = np.load(os.path.join(golden_dir, "crnn_forward.npz"))
    feats = g[f"feats_{tag}"] if f"feats_{tag}" in g else np.load(os.path.join(golden_dir, f"crnn_{tag}.npz"))["feats"]
    m, _ = _model(seed, ncls, precision)
    got = m(torch.from_numpy(feats).cuda()).cpu().numpy()
    want = g[f"probs_{tag}"]
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-4          # well inside the 1e-3 budget


def test_gru_stage_alone_matches_reference_golden(golden_dir):
    """Feeds the reference's own conv output to the HIP GRU, so that conv
    rounding cannot mask or cause a recurrence error."""
    from sound_event_detection_amd import _lib
    g = np.load(os.path.join(golden_dir, "crnn_seed0.npz"))
    m, _ = _model(0, 10)
    seq = torch.from_numpy(np.ascontiguousarray(np.transpose(g["cnn_out"][..., 0], (0, 2, 1)))).cuda()
    B, Tp, _ = seq.shape
    handle = m._get_handle(seq.device)
    lib = _lib.lib()
    ws_bytes = lib.sedmi_crnn_workspace_bytes(handle, B, 2 * Tp)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    rec = torch.empty((B, Tp, 256), device="cuda")
    _lib.check(lib.sedmi_crnn_gru_f32(handle, _lib.ptr(seq), B, Tp, _lib.ptr(rec), _lib.ptr(ws), ws_bytes,
                                      _lib.stream_ptr(seq.device)))
    assert np.abs(rec.cpu().numpy() - g["gru_out"]).max() < 2e-5


def test_gru_layer0_fused_and_separate_kernels_agree(golden_dir, monkeypatch):
    """Layer 0 runs with its input projection inside the recurrence by default
    and as GEMM + recurrence with SEDMI_GRU0_FUSED=0: both against the
    reference's GRU output, and against each other, on a ragged batch."""
    from sound_event_detection_amd import _lib
    g = np.load(os.path.join(golden_dir, "crnn_seed0.npz"))
    m, _ = _model(0, 10)
    ref_seq = np.ascontiguousarray(np.transpose(g["cnn_out"][..., 0], (0, 2, 1)))
    rng = np.random.Generator(np.random.PCG64(11))
    big = (rng.standard_normal((37, 61, 128)) * 2).astype(np.float32)       # 37 clips: a partly filled workgroup
    lib = _lib.lib()
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SEDMI_GRU0_FUSED", mode)
        for name, arr in (("ref", ref_seq), ("big", big)):
            seq = torch.from_numpy(arr).cuda()
            B, Tp, _ = seq.shape
            handle = m._get_handle(seq.device)
            ws_bytes = lib.sedmi_crnn_workspace_bytes(handle, B, 2 * Tp)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
            rec = torch.empty((B, Tp, 256), device="cuda")
            _lib.check(lib.sedmi_crnn_gru_f32(handle, _lib.ptr(seq), B, Tp, _lib.ptr(rec), _lib.ptr(ws), ws_bytes,
                                              _lib.stream_ptr(seq.device)))
            outs[mode, name] = rec.cpu().numpy()
        assert np.abs(outs[mode, "ref"] - g["gru_out"]).max() < 2e-5
    assert np.abs(outs["1", "big"] - outs["0", "big"]).max() < 2e-5
    assert np.isfinite(outs["1", "big"]).all()


@pytest.mark.parametrize("ncls", [1, 10, 16])
def test_head_inside_the_last_gru_layer_agrees_with_the_head_kernel(ncls, monkeypatch):
    """With at most 16 classes the full forward forms the head's products inside
    layer 1's recurrence; SEDMI_HEAD_FUSED=0 runs the separate head kernel."""
    m, sd = _model(4, ncls)
    rng = np.random.Generator(np.random.PCG64(ncls))
    x = torch.from_numpy(rng.gamma(0.5, 2.0, (37, 61, 128)).astype(np.float32)).cuda()     # a partly filled workgroup
    fused = m(x)
    monkeypatch.setenv("SEDMI_HEAD_FUSED", "0")
    separate = m(x)
    assert fused.shape == separate.shape == (37, 30, ncls)
    assert (fused - separate).abs().max() < 1e-6
    want = ocrnn.crnn_forward(sd, x.cpu()).numpy()
    assert np.abs(fused.cpu().numpy() - want).max() < PROB_TOL


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("B,T", [(1, 501), (5, 501), (33, 64), (70, 9), (3, 2), (2, 7501), (600, 250)])
def test_forward_matches_oracle_on_ragged_shapes(B, T, precision):
    m, sd = _model(1, 10, precision)
    rng = np.random.Generator(np.random.PCG64(B * 1000 + T))
    x = (rng.gamma(0.5, 2.0, (B, T, 128))).astype(np.float32)       # power-like, heavy tailed
    got = m(torch.from_numpy(x).cuda()).cpu().numpy()
    want = ocrnn.crnn_forward(sd, torch.from_numpy(x)).numpy()
    assert got.shape == want.shape == (B, T // 2, 10)
    assert np.abs(got - want).max() < PROB_TOL


@pytest.mark.parametrize("precision", PRECISIONS)
def test_batch_independence_and_determinism(precision):
    """Clips are independent (they shard across GPUs): a clip's output must not
    depend on its batch position or on its neighbours, and must repeat exactly."""
    m, _ = _model(2, 10, precision)
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(67, 501, 128, device="cuda", generator=g) * 3
    a = m(x)
    b = m(x)
    assert torch.equal(a, b)
    perm = torch.randperm(67, device="cuda", generator=g)
    c = m(x[perm])
    assert torch.equal(c, a[perm])
    assert torch.equal(m(x[10:11]), a[10:11])


@pytest.mark.parametrize("precision", PRECISIONS)
def test_batches_larger_than_one_pass_through_the_conv_stack(precision):
    """The conv stack takes 1024 clips per pass (blocks 2-4 in inner passes of 512) and the GRU / head kernels tile
    the whole batch: 2085 clips (two full passes and a ragged third of 37) of six distinct feature maps must give, at
    every batch position, the bits the same feature map gives in a batch of its own -- first, last and straddling
    positions of every outer and inner pass included."""
    m, _ = _model(2, 10, precision)
    g = torch.Generator(device="cuda").manual_seed(11)
    base = torch.rand(6, 501, 128, device="cuda", generator=g) * 3
    want = m(base)
    idx = torch.arange(2085, device="cuda") % 6
    got = m(base[idx].contiguous())
    assert got.shape == (2085, 250, 10)
    assert torch.equal(got, want[idx])


def test_reload_weights_rebuilds_handle():
    m, sd = _model(4, 10)
    x = torch.rand(2, 101, 128, device="cuda")
    a = m(x)
    sd2 = synth.synth_crnn_state_dict(5, nclass=10)
    m.load_state_dict(sd2)
    b = m(x)
    want = ocrnn.crnn_forward(sd2, x.cpu()).numpy()
    assert not torch.equal(a, b)
    assert np.abs(b.cpu().numpy() - want).max() < PROB_TOL


def test_errors_are_loud():
    from sound_event_detection_amd import _lib
    m, _ = _model(0, 10)
    with pytest.raises(_lib.SedmiError):
        m(torch.zeros(1, 501, 128))                 # CPU tensor: no fallback
    with pytest.raises(ValueError):
        m(torch.zeros(1, 501, 64, device="cuda"))
    m.train()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 501, 128, device="cuda"))


def test_split_f16_tracks_f32_on_a_large_batch():
    """The two arithmetic modes of the conv blocks on 256 benchmark-like clips:
    probabilities must agree far inside the 1e-3 budget."""
    m, _ = _model(0, 10, "f32")
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand(256, 501, 128, device="cuda", generator=g) ** 8 * 300       # heavy-tailed like a power mel
    a = m(x)
    m.conv_precision = "split_f16"
    b = m(x)
    assert float((a - b).abs().max()) < 2e-4


@pytest.mark.parametrize("ncls", [1, 4, 32, 33, 92])
def test_class_counts_cover_both_head_kernels(ncls):
    # <= 32 classes: weights in LDS; more: scalar-load variant
    m, sd = _model(2, ncls)
    rng = np.random.Generator(np.random.PCG64(ncls))
    x = rng.gamma(0.5, 2.0, (3, 70, 128)).astype(np.float32)
    got = m(torch.from_numpy(x).cuda()).cpu().numpy()
    want = ocrnn.crnn_forward(sd, torch.from_numpy(x)).numpy()
    assert got.shape == want.shape == (3, 35, ncls)
    assert np.abs(got - want).max() < PROB_TOL
