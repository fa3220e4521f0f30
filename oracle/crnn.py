"""Oracle: baseline CRNN eval-mode forward.  TEST INFRASTRUCTURE ONLY.

Functional restatement over a plain state_dict, using the same torch CPU
primitives the reference's modules resolve to:

* conv stack  — nnet/CNN.py:66-98 (Conv2d 3x3/1/1, BatchNorm2d eps 1e-3 in
  eval mode, the reference's own GLU = Linear(x) * sigmoid(x) over the
  channel axis at nnet/CNN.py:5-16, Dropout = identity, AvgPool2d);
* recurrent   — nnet/RNN.py:19-30 (nn.GRU, bidirectional, batch_first);
* glue + head — nnet/CRNN.py:192 (unsqueeze), :212-213 (squeeze freq,
  [B,C,T] -> [B,T,C]), :265-266, :142-143 (sigmoid(Linear)); the attention
  branch at :146-156 only feeds the weak output that forward() drops at :268.

Pinned against the imported reference classes for the conv stack and the GRU
(tests/golden/crnn_seed*.npz) and against outputs of the reference's
CRNN.forward itself for 1, 6, 10 and 92 classes (tests/golden/crnn_forward.npz),
all written by oracle/make_golden.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

FILTERS = (16, 32, 64, 128, 128, 128, 128)
POOLING = ((2, 2), (1, 2), (1, 2), (1, 2), (1, 2), (1, 2), (1, 2))
BN_EPS = 1e-3


def conv_block(sd, i: int, x: torch.Tensor, pool, return_pre_pool: bool = False):
    """One block of nnet/CNN.py:66-98 on NCHW input."""
    p = "cnn.cnn."
    x = F.conv2d(x, sd[f"{p}conv{i}.weight"], sd[f"{p}conv{i}.bias"], stride=1, padding=1)
    x = F.batch_norm(x, sd[f"{p}batchnorm{i}.running_mean"], sd[f"{p}batchnorm{i}.running_var"],
                     sd[f"{p}batchnorm{i}.weight"], sd[f"{p}batchnorm{i}.bias"],
                     training=False, eps=BN_EPS)
    lin = F.linear(x.permute(0, 2, 3, 1), sd[f"{p}glu{i}.linear.weight"], sd[f"{p}glu{i}.linear.bias"])
    x = lin.permute(0, 3, 1, 2) * torch.sigmoid(x)
    if return_pre_pool:
        return x
    return F.avg_pool2d(x, tuple(pool))


def cnn_forward(sd, feats: torch.Tensor, filters=FILTERS, pooling=POOLING, upto: int | None = None):
    """feats [B, T, F] -> NCHW activation after ``upto`` blocks (all if None)."""
    x = feats.unsqueeze(1)
    n = len(filters) if upto is None else upto
    for i in range(n):
        x = conv_block(sd, i, x, pooling[i])
    return x


def _gru_module(sd, n_in: int, hidden: int, layers: int, prefix: str = "rnn.rnn.") -> torch.nn.GRU:
    gru = torch.nn.GRU(n_in, hidden, bidirectional=True, batch_first=True, num_layers=layers)
    gru.load_state_dict({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)})
    gru.eval()
    return gru


def gru_forward(sd, x: torch.Tensor, hidden: int = 128, layers: int = 2) -> torch.Tensor:
    """[B, T, n_in] -> [B, T, 2*hidden] through torch.nn.GRU (the primitive
    nnet/RNN.py:19 instantiates)."""
    with torch.no_grad():
        out, _ = _gru_module(sd, x.shape[-1], hidden, layers)(x)
    return out


def gru_forward_explicit(sd, x: torch.Tensor, hidden: int = 128, layers: int = 2) -> torch.Tensor:
    """Same recurrence written out gate by gate (PyTorch's GRU definition:
    gates ordered r, z, n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn));
    h' = (1 - z) * n + z * h; h0 = 0).  Used to check the module-based path
    and as the specification the HIP recurrence kernel is written to."""
    B, T, _ = x.shape
    H = hidden
    inp = x
    for layer in range(layers):
        outs = []
        for sfx, order in (("", range(T)), ("_reverse", range(T - 1, -1, -1))):
            w_ih = sd[f"rnn.rnn.weight_ih_l{layer}{sfx}"]
            w_hh = sd[f"rnn.rnn.weight_hh_l{layer}{sfx}"]
            b_ih = sd[f"rnn.rnn.bias_ih_l{layer}{sfx}"]
            b_hh = sd[f"rnn.rnn.bias_hh_l{layer}{sfx}"]
            gi_all = F.linear(inp, w_ih, b_ih)                  # [B, T, 3H]
            h = torch.zeros(B, H, dtype=x.dtype)
            out = torch.empty(B, T, H, dtype=x.dtype)
            for t in order:
                gi = gi_all[:, t]
                gh = F.linear(h, w_hh, b_hh)
                r = torch.sigmoid(gi[:, :H] + gh[:, :H])
                z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
                n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
                h = (1.0 - z) * n + z * h
                out[:, t] = h
            outs.append(out)
        inp = torch.cat(outs, dim=-1)
    return inp


def head_forward(sd, x: torch.Tensor) -> torch.Tensor:
    """Strong head of nnet/CRNN.py:142-143: sigmoid(Linear(x))."""
    return torch.sigmoid(F.linear(x, sd["dense.weight"], sd["dense.bias"]))


def crnn_forward(sd, feats: torch.Tensor, explicit_gru: bool = False) -> torch.Tensor:
    """feats [B, T, F] float32 -> frame probabilities [B, T', n_cls]."""
    with torch.no_grad():
        x = cnn_forward(sd, feats)                 # [B, C, T', 1]
        x = x.squeeze(-1).permute(0, 2, 1)         # [B, T', C]
        x = (gru_forward_explicit if explicit_gru else gru_forward)(sd, x.contiguous())
        return head_forward(sd, x)
