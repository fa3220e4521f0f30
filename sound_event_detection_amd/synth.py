"""Deterministic synthetic inputs for the SED hot path: clips and CRNN weights.

Nothing here touches a GPU or the oracle.  Both the parity tests and the
benchmark draw their inputs from these generators so that the CPU checker and
the HIP path always see identical bytes, and so that no weight file has to be
committed (the reference's own ``weights_{fold}.pth`` files are not
distributed with it: /root/reference/.gitignore:3).

Clip i is white noise of amplitude 0.1 from ``PCG64(1234 + i)`` plus one to
four gated sinusoid bursts, which gives the classifier something to fire on.
Weights follow the state_dict schema of the reference's baseline CRNN as it is
constructed at training.py:207-212 (83 keys, see ``crnn_state_dict_spec``).
"""
from __future__ import annotations

import numpy as np
import torch

# training.py:207-212 — the only CRNN configuration the reference ever builds.
CRNN_FILTERS = (16, 32, 64, 128, 128, 128, 128)
CRNN_POOLING = ((2, 2), (1, 2), (1, 2), (1, 2), (1, 2), (1, 2), (1, 2))
CRNN_HIDDEN = 128
CRNN_GRU_LAYERS = 2


def synth_clip(index: int, n_samples: int = 160000, sr: int = 16000) -> np.ndarray:
    """Clip ``index``: float32 [n_samples], noise 0.1 rms + 1..4 tone bursts."""
    rng = np.random.Generator(np.random.PCG64(1234 + int(index)))
    y = rng.standard_normal(n_samples).astype(np.float32) * np.float32(0.1)
    n_bursts = int(rng.integers(1, 5))
    t = np.arange(n_samples, dtype=np.float64) / float(sr)
    for _ in range(n_bursts):
        f0 = float(rng.uniform(200.0, 0.45 * sr))
        amp = float(rng.uniform(0.1, 0.5))
        start = int(rng.integers(0, max(1, n_samples - sr // 4)))
        length = int(rng.integers(sr // 4, 3 * sr))
        stop = min(n_samples, start + length)
        y[start:stop] += (amp * np.sin(2.0 * np.pi * f0 * t[start:stop])).astype(np.float32)
    return y


def synth_clips(first: int, count: int, n_samples: int = 160000, sr: int = 16000) -> np.ndarray:
    """float32 [count, n_samples], clip indices first .. first+count-1."""
    out = np.empty((count, n_samples), dtype=np.float32)
    for k in range(count):
        out[k] = synth_clip(first + k, n_samples, sr)
    return out


def synth_labelled_clip(index: int, n_cls: int = 10, n_samples: int = 160000, sr: int = 16000, frames: int = 250):
    """A clip with ground truth: noise 0.05 rms plus 1..4 events, each of a class drawn at random.  Class c
    sounds as a tone (with a weaker octave) in its own frequency band, 300 Hz * 1.35**c, amplitude-modulated at
    3 + c Hz; an event of class c switches label[frame, c] on for the 40 ms frames it overlaps (the
    rasterisation of utils.py:255-281).  Returns (float32 [n_samples], float32 [frames, n_cls])."""
    rng = np.random.Generator(np.random.PCG64(80_000 + int(index)))
    y = rng.standard_normal(n_samples).astype(np.float32) * np.float32(0.05)
    labels = np.zeros((frames, n_cls), dtype=np.float32)
    t = np.arange(n_samples, dtype=np.float64) / float(sr)
    per_frame = n_samples / frames
    for _ in range(int(rng.integers(1, 5))):
        c = int(rng.integers(0, n_cls))
        f0 = 300.0 * 1.35 ** c * float(rng.uniform(0.97, 1.03))
        amp = float(rng.uniform(0.15, 0.5))
        start = int(rng.integers(0, n_samples - sr // 2))
        stop = min(n_samples, start + int(rng.integers(sr // 2, 4 * sr)))
        seg = t[start:stop]
        env = 0.6 + 0.4 * np.sin(2.0 * np.pi * (3.0 + c) * seg)
        y[start:stop] += (amp * env * (np.sin(2.0 * np.pi * f0 * seg) + 0.3 * np.sin(4.0 * np.pi * f0 * seg))).astype(np.float32)
        labels[int(start // per_frame):int(np.ceil(stop / per_frame)), c] = 1.0
    return y, labels


def synth_labelled_clips(first: int, count: int, n_cls: int = 10):
    """(float32 [count, 160000], float32 [count, 250, n_cls]) for clip indices first .. first + count - 1."""
    pairs = [synth_labelled_clip(first + k, n_cls) for k in range(count)]
    return np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])


def crnn_state_dict_spec(nclass: int = 10, n_in_channel: int = 1,
                         filters=CRNN_FILTERS, hidden: int = CRNN_HIDDEN,
                         gru_layers: int = CRNN_GRU_LAYERS):
    """Ordered (key, shape) list of the baseline CRNN's state_dict.

    Key names are those PyTorch derives from the reference's module tree:
    nnet/CNN.py:69-91 (conv{i}, batchnorm{i}, glu{i}.linear inside the
    ``cnn`` Sequential of ``CNN``, itself the ``cnn`` attribute of ``CRNN``),
    nnet/RNN.py:19-26 (``rnn.rnn`` = nn.GRU), nnet/CRNN.py:109-113 (dense,
    dense_softmax).
    """
    spec = []
    cin = n_in_channel
    for i, cout in enumerate(filters):
        p = f"cnn.cnn.conv{i}"
        spec += [(f"{p}.weight", (cout, cin, 3, 3)), (f"{p}.bias", (cout,))]
        p = f"cnn.cnn.batchnorm{i}"
        spec += [(f"{p}.weight", (cout,)), (f"{p}.bias", (cout,)),
                 (f"{p}.running_mean", (cout,)), (f"{p}.running_var", (cout,)),
                 (f"{p}.num_batches_tracked", ())]
        p = f"cnn.cnn.glu{i}.linear"
        spec += [(f"{p}.weight", (cout, cout)), (f"{p}.bias", (cout,))]
        cin = cout
    for layer in range(gru_layers):
        n_in = filters[-1] if layer == 0 else 2 * hidden
        for sfx in ("", "_reverse"):
            spec += [(f"rnn.rnn.weight_ih_l{layer}{sfx}", (3 * hidden, n_in)),
                     (f"rnn.rnn.weight_hh_l{layer}{sfx}", (3 * hidden, hidden)),
                     (f"rnn.rnn.bias_ih_l{layer}{sfx}", (3 * hidden,)),
                     (f"rnn.rnn.bias_hh_l{layer}{sfx}", (3 * hidden,))]
    spec += [("dense.weight", (nclass, 2 * hidden)), ("dense.bias", (nclass,)),
             ("dense_softmax.weight", (nclass, 2 * hidden)), ("dense_softmax.bias", (nclass,))]
    return spec


def synth_crnn_state_dict(seed: int = 0, nclass: int = 10, head_gain: float = 4.0):
    """Deterministic CRNN weights as an ordered dict of CPU float32 tensors.

    Fan-in scaled uniform weights.  BatchNorm running statistics are what
    training would have left behind on these clips: near (0, 1) for blocks
    1..6, and (0.8, 190) for block 0, whose input is the heavy-tailed power mel
    spectrogram (no log in the reference: std ~13, bursts up to several
    hundred).  That keeps every block's output O(1) and the network as well
    conditioned as a trained one: float32 and float64 evaluations of it agree
    to ~1e-5 on the probabilities.  The strong head is scaled by ``head_gain``
    so that frame probabilities spread on both sides of 0.5 instead of sitting
    at 0.5 +- 0.02.
    """
    rng = np.random.Generator(np.random.PCG64(977 + int(seed)))
    sd = {}
    for key, shape in crnn_state_dict_spec(nclass):
        leaf = key.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[key] = torch.tensor(100, dtype=torch.int64)
            continue
        if "batchnorm" in key:
            first = key.startswith("cnn.cnn.batchnorm0.")
            if leaf == "weight":
                v = rng.uniform(0.8, 1.2, shape)
            elif leaf == "bias":
                v = rng.uniform(-0.1, 0.1, shape)
            elif leaf == "running_mean":
                v = rng.uniform(-0.1, 0.1, shape) + (0.8 if first else 0.0)
            else:  # running_var
                v = rng.uniform(0.8, 1.2, shape) * (190.0 if first else 1.0)
        elif leaf.startswith("bias"):
            v = rng.uniform(-0.1, 0.1, shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            bound = np.sqrt(3.0 / fan_in)
            if key.startswith("cnn.cnn.conv"):
                bound *= 1.25         # keeps the signal alive through 7 GLU gates
            if ".glu" in key:
                bound *= 1.45
            if key == "dense.weight":
                bound *= head_gain
            v = rng.uniform(-bound, bound, shape)
        sd[key] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape).copy())
    return sd


# ---------------------------------------------------------------------------
# ShakeRCNN / ShakeTransformer (models.py:103-221)
# ---------------------------------------------------------------------------
SHAKE_BLOCKS = (("cnn1", 1, 16, 5), ("cnn2", 16, 32, 3), ("cnn3", 32, 64, 3))


def _shake_cnn_spec():
    spec = []
    for name, cin, cout, k in SHAKE_BLOCKS:
        spec += [(f"{name}.conv.weight", (2 * cout, cin, k, k)), (f"{name}.conv.bias", (2 * cout,))]
        for bn, c in (("bn1", 2 * cout),):
            spec += [(f"{name}.{bn}.weight", (c,)), (f"{name}.{bn}.bias", (c,)), (f"{name}.{bn}.running_mean", (c,)),
                     (f"{name}.{bn}.running_var", (c,)), (f"{name}.{bn}.num_batches_tracked", ())]
        for br in ("branch1", "branch2"):
            spec += [(f"{name}.shake.{br}.weight", (cout, cout, 3, 3)), (f"{name}.shake.{br}.bias", (cout,))]
        spec += [(f"{name}.bn2.weight", (cout,)), (f"{name}.bn2.bias", (cout,)), (f"{name}.bn2.running_mean", (cout,)),
                 (f"{name}.bn2.running_var", (cout,)), (f"{name}.bn2.num_batches_tracked", ())]
    return spec


def shake_rcnn_state_dict_spec(num_classes: int = 10):
    """Ordered (key, shape) list of ShakeRCNN's state_dict (models.py:105-121)."""
    spec = _shake_cnn_spec()
    for sfx in ("", "_reverse"):
        spec += [(f"rnn.weight_ih_l0{sfx}", (3072, 6144)), (f"rnn.weight_hh_l0{sfx}", (3072, 1024)),
                 (f"rnn.bias_ih_l0{sfx}", (3072,)), (f"rnn.bias_hh_l0{sfx}", (3072,))]
    spec += [("layernorm.weight", (2048,)), ("layernorm.bias", (2048,)),
             ("fc1.weight", (num_classes, 2048)), ("fc1.bias", (num_classes,))]
    return spec


def shake_transformer_state_dict_spec(num_classes: int = 10, layers: int = 6):
    """Ordered (key, shape) list of ShakeTransformer's state_dict (models.py:185-202)."""
    spec = _shake_cnn_spec()
    spec += [("fc1.weight", (1024, 6144)), ("fc1.bias", (1024,))]
    for i in range(layers):
        p = f"encoder.layers.{i}."
        spec += [(p + "self_attn.in_proj_weight", (3072, 1024)), (p + "self_attn.in_proj_bias", (3072,)),
                 (p + "self_attn.out_proj.weight", (1024, 1024)), (p + "self_attn.out_proj.bias", (1024,)),
                 (p + "linear1.weight", (2048, 1024)), (p + "linear1.bias", (2048,)),
                 (p + "linear2.weight", (1024, 2048)), (p + "linear2.bias", (1024,)),
                 (p + "norm1.weight", (1024,)), (p + "norm1.bias", (1024,)),
                 (p + "norm2.weight", (1024,)), (p + "norm2.bias", (1024,))]
    spec += [("fc2.weight", (num_classes, 1024)), ("fc2.bias", (num_classes,))]
    return spec


def synth_shake_state_dict(spec, seed: int = 0, head_gain: float = 3.0):
    """Deterministic weights for a Shake model: fan-in scaled uniform matrices,
    BatchNorm / LayerNorm statistics near (0, 1), head scaled by ``head_gain``."""
    rng = np.random.Generator(np.random.PCG64(4711 + int(seed)))
    sd = {}
    head_key = "fc2.weight" if any(k == "fc2.weight" for k, _ in spec) else "fc1.weight"
    for key, shape in spec:
        leaf = key.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[key] = torch.tensor(100, dtype=torch.int64)
            continue
        n = int(np.prod(shape)) if shape else 1
        norm_like = ".bn" in key or "norm" in key
        if norm_like and leaf == "weight":
            v = rng.uniform(0.8, 1.2, n)
        elif norm_like and leaf == "running_var":
            v = rng.uniform(0.8, 1.2, n)
        elif norm_like or leaf.startswith("bias") or leaf.endswith("bias"):
            v = rng.uniform(-0.1, 0.1, n)
        else:
            fan_in = int(np.prod(shape[1:]))
            bound = np.sqrt(3.0 / fan_in)
            if ".conv." in key:
                bound *= 1.5
            if key == head_key:
                bound *= head_gain
            v = rng.uniform(-bound, bound, n)
        sd[key] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape).copy())
    return sd


def synth_shake_feats(seed: int, n_clips: int, n_frames: int = 1000, n_bins: int = 768) -> np.ndarray:
    """[n_clips, n_frames, n_bins] float32 stand-in for the AST features the Shake models are fed
    (feature_extraction.py:23,57): unit normal noise plus a slow per-bin drift, seeded."""
    rng = np.random.Generator(np.random.PCG64(9000 + int(seed)))
    x = rng.standard_normal((n_clips, n_frames, n_bins), dtype=np.float32)
    drift = np.sin(np.arange(n_frames, dtype=np.float32)[:, None] * 0.02 + np.arange(n_bins, dtype=np.float32)[None] * 0.05)
    return (x + 0.5 * drift).astype(np.float32)


def basic_rcnn_state_dict_spec(num_classes: int = 10, frequency_dim: int = 64):
    """Ordered (key, shape) list of BasicRCNN's state_dict (models.py:29-43, default hyper-parameters)."""
    k = ((frequency_dim - 8) // 2) * 128
    return [("conv1.weight", (128, 1, 3, 3)), ("conv1.bias", (128,)),
            ("conv2.weight", (128, 128, 3, 3)), ("conv2.bias", (128,)),
            ("conv3.weight", (128, 128, 3, 3)), ("conv3.bias", (128,)),
            ("gru.weight_ih_l0", (96, k)), ("gru.weight_hh_l0", (96, 32)),
            ("gru.bias_ih_l0", (96,)), ("gru.bias_hh_l0", (96,)),
            ("fc1.weight", (num_classes, 32)), ("fc1.bias", (num_classes,))]
