"""Baseline CRNN on MI355X behind the reference's interface.

Mirrors /root/reference/nnet/CRNN.py: same constructor keywords (:12-33), same
``forward(x, pad_mask=None, embeddings=None, classes_mask=None)`` returning the
strong frame probabilities ``[B, T', n_class]`` (:191-268), same state_dict
keys (83 for the configuration of training.py:207-212), so
``model.load_state_dict(torch.load('weights_1.pth'))`` works unchanged.

The module tree only owns the tensors.  ``forward`` packs them once into a
``sedmi_crnn`` handle (BatchNorm folded to an affine, matrices laid out as MFMA
fragments) and runs conv blocks, GRU and head in the HIP kernels of
csrc/crnn.hip through the C ABI.  Inference only: there is no backward and no
CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch
import torch.nn as nn

from .. import _lib
from .CNN import CNN
from .RNN import BidirectionalGRU

_SUPPORTED_FILTERS = [16, 32, 64, 128, 128, 128, 128]


class CRNN(nn.Module):
    def __init__(self, n_in_channel=1, nclass=10, attention=True, activation="glu", dropout=0.5,
                 train_cnn=True, rnn_type="BGRU", n_RNN_cell=128, n_layers_RNN=2, dropout_recurrent=0,
                 cnn_integration=False, freeze_bn=False, use_embeddings=False, embedding_size=527,
                 embedding_type="global", frame_emb_enc_dim=512, aggregation_type="global",
                 dropstep_recurrent=0.0, **kwargs):
        super().__init__()
        if cnn_integration or use_embeddings or dropstep_recurrent:
            raise NotImplementedError("cnn_integration / use_embeddings / dropstep_recurrent are never enabled "
                                      "by the reference's training.py and are not implemented")
        if rnn_type != "BGRU":
            raise NotImplementedError("Only BGRU supported for CRNN for now")
        if isinstance(nclass, (tuple, list)):
            if len(nclass) > 1:
                raise NotImplementedError("multi-head CRNN is not implemented")
            nclass = nclass[0]
        self.n_in_channel = n_in_channel
        self.attention = attention
        self.cnn_integration = cnn_integration
        self.freeze_bn = freeze_bn
        self.use_embeddings = use_embeddings
        self.embedding_type = embedding_type
        self.aggregation_type = aggregation_type
        self.nclass = nclass
        self.dropstep_recurrent = dropstep_recurrent
        self.train_cnn = train_cnn
        self.n_RNN_cell = n_RNN_cell
        self.n_layers_RNN = n_layers_RNN

        # unknown keywords (e.g. training.py's rnn_layers=2) fall through to CNN and are ignored there,
        # as in the reference (CRNN.py:69-71, CNN.py:45)
        self.cnn = CNN(n_in_channel=n_in_channel, activation=activation, conv_dropout=dropout, **kwargs)
        self.rnn = BidirectionalGRU(n_in=self.cnn.nb_filters[-1], n_hidden=n_RNN_cell,
                                    dropout=dropout_recurrent, num_layers=n_layers_RNN)
        self.dropout = nn.Dropout(dropout)
        self.dense = nn.Linear(n_RNN_cell * 2, nclass)
        self.sigmoid = nn.Sigmoid()
        if attention:
            # kept for state_dict compatibility; its output (the weak prediction) is dropped by
            # forward() in the reference (CRNN.py:268), so it is never evaluated here
            self.dense_softmax = nn.Linear(n_RNN_cell * 2, nclass)
            self.softmax = nn.Softmax(dim=-1)
        self._handle = None
        self._handle_sig = None
        self._handle_fin = None
        # arithmetic of the conv blocks, the GRU input projections and the recurrence: "split_f16" (default;
        # float32 operands as two f16 halves on the f16 matrix cores, float32-grade accuracy; input features of
        # any magnitude, activations between layers must stay below 65504) or "f32" (exact float32, ~3x slower)
        self.conv_precision = os.environ.get("SEDMI_CONV_PRECISION", "split_f16")

    # ------------------------------------------------------------------ handle
    def _signature(self):
        ts = list(self.parameters()) + list(self.buffers())
        return tuple((t.data_ptr(), t._version) for t in ts)

    def _check_supported(self):
        cnn = self.cnn
        ok = (list(cnn.nb_filters) == _SUPPORTED_FILTERS and self.n_in_channel == 1
              and cnn.pooling == [(2, 2)] + [(1, 2)] * 6 and list(cnn.kernel_size) == [3] * 7
              and list(cnn.padding) == [1] * 7 and list(cnn.stride) == [1] * 7
              and self.n_RNN_cell == 128 and self.n_layers_RNN == 2)
        if not ok:
            raise NotImplementedError(
                "the HIP path implements the CRNN configuration of training.py:207-212 only: 1 input channel, "
                "filters 16,32,64,128,128,128,128, 3x3/1/1 convs, pooling (2,2),(1,2)x6, 2-layer BGRU(128)")

    def _apply_precision(self):
        modes = {"f32": _lib.PRECISION_F32, "split_f16": _lib.PRECISION_SPLIT_F16}
        if self.conv_precision not in modes:
            raise ValueError(f"conv_precision must be one of {sorted(modes)}")
        _lib.check(_lib.lib().sedmi_crnn_set_precision(self._handle, modes[self.conv_precision]))

    def _get_handle(self, device):
        sig = (device.index, self._signature())
        if self._handle is not None and self._handle_sig == sig:
            self._apply_precision()
            return self._handle
        self._check_supported()
        self._release()
        sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in self.state_dict().items()
              if v.is_floating_point()}
        w = _lib.CrnnWeights()
        w.n_class, w.n_freq, w.hidden, w.bn_eps = int(self.nclass), 128, 128, 1e-3
        fp = C.POINTER(C.c_float)

        def p(key):
            return C.cast(C.c_void_p(sd[key].data_ptr()), fp)

        for i in range(7):
            w.filters[i] = _SUPPORTED_FILTERS[i]
            w.pool_t[i], w.pool_f[i] = self.cnn.pooling[i]
            w.conv_w[i], w.conv_b[i] = p(f"cnn.cnn.conv{i}.weight"), p(f"cnn.cnn.conv{i}.bias")
            w.bn_w[i], w.bn_b[i] = p(f"cnn.cnn.batchnorm{i}.weight"), p(f"cnn.cnn.batchnorm{i}.bias")
            w.bn_mean[i] = p(f"cnn.cnn.batchnorm{i}.running_mean")
            w.bn_var[i] = p(f"cnn.cnn.batchnorm{i}.running_var")
            w.glu_w[i], w.glu_b[i] = p(f"cnn.cnn.glu{i}.linear.weight"), p(f"cnn.cnn.glu{i}.linear.bias")
        for layer in range(2):
            for d, sfx in enumerate(("", "_reverse")):
                w.gru_w_ih[layer][d] = p(f"rnn.rnn.weight_ih_l{layer}{sfx}")
                w.gru_w_hh[layer][d] = p(f"rnn.rnn.weight_hh_l{layer}{sfx}")
                w.gru_b_ih[layer][d] = p(f"rnn.rnn.bias_ih_l{layer}{sfx}")
                w.gru_b_hh[layer][d] = p(f"rnn.rnn.bias_hh_l{layer}{sfx}")
        w.dense_w, w.dense_b = p("dense.weight"), p("dense.bias")
        handle = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().sedmi_crnn_create(C.byref(handle), C.byref(w)), "sedmi_crnn_create")
        self._handle, self._handle_sig = handle, sig
        self._handle_fin = weakref.finalize(self, _lib.lib().sedmi_crnn_destroy, handle)
        self._apply_precision()
        return handle

    def _release(self):
        if self._handle_fin is not None:
            self._handle_fin()          # runs sedmi_crnn_destroy once
        self._handle = self._handle_sig = self._handle_fin = None

    # ---- what pipeline.SedPipeline needs from a model: output kind, shapes, and a forward into given buffers
    returns_logits = False              # CRNN.forward ends in a sigmoid (CRNN.py:143; training.py:358)
    time_pooling = 2
    feature_width = 128

    @property
    def n_outputs(self):
        return int(self.nclass)

    def _workspace_bytes(self, handle, B, T):
        return _lib.lib().sedmi_crnn_workspace_bytes(handle, B, T)

    def _forward_into(self, handle, x_ptr, B, T, out, ws, ws_bytes, stream, alphas=None):
        _lib.check(_lib.lib().sedmi_crnn_forward_f32(handle, x_ptr, B, T, _lib.ptr(out), _lib.ptr(ws), ws_bytes,
                                                     stream), "sedmi_crnn_forward_f32")

    # ----------------------------------------------------------------- forward
    def _prepare(self, x):
        if self.training:
            raise RuntimeError("this CRNN is inference-only: call model.eval() first (training.py:275,313)")
        if not x.is_cuda:
            raise _lib.SedmiError("CRNN.forward needs a CUDA (ROCm) tensor: the model runs in HIP kernels only")
        if x.dim() != 3 or x.shape[2] != 128:
            raise ValueError(f"expected features [batch, frames, 128], got {tuple(x.shape)}")
        return x.to(torch.float32).contiguous(), self._get_handle(x.device)

    def forward(self, x, pad_mask=None, embeddings=None, classes_mask=None):
        if pad_mask is not None or embeddings is not None or classes_mask is not None:
            raise NotImplementedError("pad_mask / embeddings / classes_mask are not used by the reference's "
                                      "training.py and are not implemented")
        x, handle = self._prepare(x)
        B, T, _ = x.shape
        lib = _lib.lib()
        Tp = T // 2
        probs = torch.empty((B, Tp, self.nclass), dtype=torch.float32, device=x.device)
        if B == 0 or Tp == 0:
            return probs
        ws_bytes = lib.sedmi_crnn_workspace_bytes(handle, B, T)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.sedmi_crnn_forward_f32(handle, _lib.ptr(x), B, T, _lib.ptr(probs), _lib.ptr(ws),
                                                  ws_bytes, _lib.stream_ptr(x.device)), "sedmi_crnn_forward_f32")
        return probs

    # stage-level access for layer-by-layer parity tests --------------------------
    def forward_stages(self, x):
        """(seq [B,T',128], rec [B,T',256], probs [B,T',n_class]) from the three
        stage entry points of the C ABI."""
        x, handle = self._prepare(x)
        B, T, _ = x.shape
        lib, dev, Tp = _lib.lib(), x.device, T // 2
        ws_bytes = lib.sedmi_crnn_workspace_bytes(handle, B, T)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        seq = torch.empty((B, Tp, 128), dtype=torch.float32, device=dev)
        rec = torch.empty((B, Tp, 256), dtype=torch.float32, device=dev)
        probs = torch.empty((B, Tp, self.nclass), dtype=torch.float32, device=dev)
        st = _lib.stream_ptr(dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sedmi_crnn_cnn_f32(handle, _lib.ptr(x), B, T, _lib.ptr(seq), _lib.ptr(ws), ws_bytes, st))
            _lib.check(lib.sedmi_crnn_gru_f32(handle, _lib.ptr(seq), B, Tp, _lib.ptr(rec), _lib.ptr(ws), ws_bytes, st))
            _lib.check(lib.sedmi_crnn_head_f32(handle, _lib.ptr(rec), B * Tp, _lib.ptr(probs), st))
        return seq, rec, probs
