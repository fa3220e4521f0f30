"""Oracle: ShakeRCNN / ShakeTransformer eval-mode forward.  TEST INFRASTRUCTURE ONLY.

Functional restatement over a plain state_dict of models.py:70-221 with the
torch CPU primitives the reference's modules resolve to:

* ShakeCNNBlock (models.py:155-180): Conv2d(in, 2*out, k, padding='same') ->
  BatchNorm2d (eps 1e-5) -> torch.nn.GLU(dim=1) (a * sigmoid(b), channels
  halved) -> ShakeShakeBlock -> BatchNorm2d -> ReLU -> AvgPool2d;
* ShakeShakeBlock (models.py:70-100): x * sigmoid(a*conv1(x) + (1-a)*conv2(x));
  in EVAL mode the reference draws a = torch.rand(1) per block per forward
  (models.py:88-91, train/eval inverted), so ``a`` is an explicit input here:
  three values per forward, in block order;
* ShakeRCNN (models.py:124-142): 3 blocks -> transpose(1,2) + flatten ->
  GRU(6144, 1024, bidirectional) -> LayerNorm(2048) -> LeakyReLU(0.01) ->
  Linear(2048, n_cls): logits;
* ShakeTransformer (models.py:204-221): 3 blocks -> Linear(6144,1024) -> 6 x
  TransformerEncoderLayer(d 1024, 8 heads, ff 2048, ReLU, post-LN) ->
  Linear(1024, n_cls): logits.

Pinned against the reference classes themselves (oracle/make_golden.py imports
models.py and stores logits for fixed ``a`` triples in tests/golden/shake_*.npz).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BLOCKS = (  # name, in, out, kernel, pool
    ("cnn1", 1, 16, 5, (2, 2)),
    ("cnn2", 16, 32, 3, (2, 2)),
    ("cnn3", 32, 64, 3, (1, 2)),
)


def shake_block(sd, name: str, x: torch.Tensor, k: int, pool, a: float) -> torch.Tensor:
    p = name + "."
    x = F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=k // 2)
    x = F.batch_norm(x, sd[p + "bn1.running_mean"], sd[p + "bn1.running_var"], sd[p + "bn1.weight"],
                     sd[p + "bn1.bias"], training=False, eps=1e-5)
    x = F.glu(x, dim=1)
    x1 = F.conv2d(x, sd[p + "shake.branch1.weight"], sd[p + "shake.branch1.bias"], padding=1)
    x2 = F.conv2d(x, sd[p + "shake.branch2.weight"], sd[p + "shake.branch2.bias"], padding=1)
    at = torch.tensor(a, dtype=x.dtype)
    x = x * torch.sigmoid(at * x1 + (1.0 - at) * x2)
    x = F.batch_norm(x, sd[p + "bn2.running_mean"], sd[p + "bn2.running_var"], sd[p + "bn2.weight"],
                     sd[p + "bn2.bias"], training=False, eps=1e-5)
    x = F.relu(x)
    return F.avg_pool2d(x, pool, pool)


def shake_cnn(sd, feats: torch.Tensor, alphas) -> torch.Tensor:
    """feats [B, T, F] -> [B, T/4, 64 * F/8] (transpose(1,2) + flatten of models.py:133-134)."""
    x = feats.unsqueeze(1)
    for (name, _, _, k, pool), a in zip(BLOCKS, alphas):
        x = shake_block(sd, name, x, k, pool, float(a))
    return torch.flatten(torch.transpose(x, 1, 2), start_dim=2)


def shake_rcnn_forward(sd, feats: torch.Tensor, alphas) -> torch.Tensor:
    with torch.no_grad():
        x = shake_cnn(sd, feats, alphas)
        gru = torch.nn.GRU(x.shape[-1], 1024, batch_first=True, bidirectional=True).to(x.dtype)
        gru.load_state_dict({k[len("rnn."):]: v for k, v in sd.items() if k.startswith("rnn.")})
        x, _ = gru(x)
        x = F.layer_norm(x, (2048,), sd["layernorm.weight"], sd["layernorm.bias"], 1e-5)
        x = F.leaky_relu(x, 0.01)
        return F.linear(x, sd["fc1.weight"], sd["fc1.bias"])


def _encoder_layer(sd, p: str, x: torch.Tensor, nhead: int = 8) -> torch.Tensor:
    B, T, D = x.shape
    qkv = F.linear(x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
    q, k, v = qkv.split(D, dim=-1)
    hd = D // nhead
    q = q.view(B, T, nhead, hd).transpose(1, 2)
    k = k.view(B, T, nhead, hd).transpose(1, 2)
    v = v.view(B, T, nhead, hd).transpose(1, 2)
    att = torch.softmax(q @ k.transpose(-1, -2) / (hd ** 0.5), dim=-1) @ v
    att = att.transpose(1, 2).reshape(B, T, D)
    att = F.linear(att, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
    x = F.layer_norm(x + att, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    ff = F.linear(F.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                  sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return F.layer_norm(x + ff, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)


def shake_transformer_forward(sd, feats: torch.Tensor, alphas, layers: int = 6) -> torch.Tensor:
    with torch.no_grad():
        x = shake_cnn(sd, feats, alphas)
        x = F.linear(x, sd["fc1.weight"], sd["fc1.bias"])
        for i in range(layers):
            x = _encoder_layer(sd, f"encoder.layers.{i}.", x)
        return F.linear(x, sd["fc2.weight"], sd["fc2.bias"])


def basic_rcnn_forward(sd, feats: torch.Tensor) -> torch.Tensor:
    """BasicRCNN.forward (models.py:45-66), default hyper-parameters: feats [B, T, F] -> logits [B, T//2, n_cls]."""
    with torch.no_grad():
        x = feats.reshape(feats.shape[0], 1, feats.shape[1], feats.shape[2]).to(sd["conv1.weight"].dtype)
        x = F.relu(F.max_pool2d(F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], padding=1), (1, 5), 1))
        x = F.relu(F.max_pool2d(F.conv2d(x, sd["conv2.weight"], sd["conv2.bias"], padding=1), (1, 5), 1))
        x = F.relu(F.max_pool2d(F.conv2d(x, sd["conv3.weight"], sd["conv3.bias"], padding=1), (2, 2), 2))
        b, c, t, w = x.shape
        x = torch.reshape(x, (b, t, c, w)).flatten(2)            # models.py:59-61: a reinterpretation, not a permute
        gru = torch.nn.GRU(x.shape[-1], 32, batch_first=True).to(x.dtype)
        gru.load_state_dict({k[len("gru."):]: v for k, v in sd.items() if k.startswith("gru.")})
        x, _ = gru(x)
        return F.linear(F.relu(x), sd["fc1.weight"], sd["fc1.bias"])
