"""BasicRCNN, ShakeRCNN and ShakeTransformer on MI355X behind the reference's interface.

Mirrors /root/reference/models.py: ``ShakeRCNN(num_classes, dropout=0)``
(:105-121), ``forward(x) -> logits [B, T/4, num_classes]`` (:124-142),
``get_weak_logits`` (:144-152) and the state_dict keys
(``cnn{1,2,3}.{conv,bn1,shake.branch1,shake.branch2,bn2}.*``, ``rnn.*_l0[_reverse]``,
``layernorm.*``, ``fc1.*``); ``ShakeTransformer(num_classes, dropout)`` (:185-202),
``forward(x) -> logits`` (:204-221) with keys ``cnn*``, ``fc1.*``,
``encoder.layers.{0..5}.*``, ``fc2.*`` — so ``load_state_dict`` of a reference
checkpoint works for both.

The module tree only owns tensors; ``forward`` packs them once into a
``sedmi_shake_rcnn`` / ``sedmi_shake_transformer`` handle and runs the HIP
kernels of csrc/shake.hip through the C ABI.  Inference only, CUDA (ROCm) tensors only, no CPU path.

ShakeRCNN's recurrence: up to 192 clips per call it runs as ONE kernel launch whose 128 / 256 workgroups must
all be resident at once (one per compute unit), i.e. on a GPU that runs nothing else meanwhile.  If that
does not hold, the kernel gives up after a bounded wait and the forward's logits are NaN; ``forward`` notices
(it waits for the stream and reads the kernel's status word), switches the model to one launch per step
(``model.recurrence = "per_step"``, 1.7-2x slower for few long recordings) and runs again;
``SedPipeline`` raises ``SedmiRecurrenceError`` from ``events()`` / ``check()`` instead of decoding NaNs.

ShakeShake mixing weight: in eval mode the reference draws ``a = torch.rand(1)``
per block on every forward (:88-91).  ``forward`` does the same (CPU generator,
three draws in block order, so the same ``torch.manual_seed`` gives the same
``a`` as the reference) unless ``alphas=(a1, a2, a3)`` is passed.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch
import torch.nn as nn

from . import _lib


class BasicRCNN(nn.Module):
    """models.py:7-66 with the constructor defaults (the only form training.py:201 builds):
    ``BasicRCNN(num_classes, frequency_dim)``; ``forward(x [B, T, frequency_dim]) -> logits
    [B, T // 2, num_classes]``; state_dict keys conv{1,2,3}.*, gru.*_l0, fc1.*."""

    def __init__(self, num_classes, frequency_dim, conv_kernels=None, conv_channels=None, maxpool_kernels=None,
                 gru_hidden_channels=32, bidirectional_gru=False, dropout=0.):
        super().__init__()
        if (conv_kernels not in (None, [3, 3, 3]) or conv_channels not in (None, [128, 128, 128])
                or maxpool_kernels not in (None, [5, 5, 2]) or gru_hidden_channels != 32 or bidirectional_gru):
            raise NotImplementedError("the HIP path implements BasicRCNN's default hyper-parameters only "
                                      "(3x3 convs, 128 channels, max-pools 5/5/2, GRU(32), one direction)")
        self.num_classes = int(num_classes)
        self.frequency_dim = int(frequency_dim)
        self.dropout = nn.Dropout(dropout)
        self.conv1 = nn.Conv2d(1, 128, 3, padding=1)
        self.conv2 = nn.Conv2d(128, 128, 3, padding=1)
        self.conv3 = nn.Conv2d(128, 128, 3, padding=1)
        self.gru_input_dim = int(((self.frequency_dim - 8) // 2) * 128)
        self.gru = nn.GRU(self.gru_input_dim, 32, batch_first=True, bidirectional=False)
        self.fc1 = nn.Linear(32, num_classes)
        self._handle = self._handle_sig = self._handle_fin = None

    def _get_handle(self, device):
        ts = list(self.parameters()) + list(self.buffers())
        sig = (device.index, tuple((t.data_ptr(), t._version) for t in ts))
        if self._handle is not None and self._handle_sig == sig:
            return self._handle
        if self._handle_fin is not None:
            self._handle_fin()
        sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in self.state_dict().items()}
        fp = C.POINTER(C.c_float)

        def p(key):
            return C.cast(C.c_void_p(sd[key].data_ptr()), fp)

        w = _lib.BasicRcnnWeights()
        w.n_class, w.frequency_dim = self.num_classes, self.frequency_dim
        for i in range(3):
            w.conv_w[i], w.conv_b[i] = p(f"conv{i + 1}.weight"), p(f"conv{i + 1}.bias")
        w.gru_w_ih, w.gru_w_hh = p("gru.weight_ih_l0"), p("gru.weight_hh_l0")
        w.gru_b_ih, w.gru_b_hh = p("gru.bias_ih_l0"), p("gru.bias_hh_l0")
        w.fc_w, w.fc_b = p("fc1.weight"), p("fc1.bias")
        handle = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().sedmi_basic_rcnn_create(C.byref(handle), C.byref(w)), "sedmi_basic_rcnn_create")
        self._handle, self._handle_sig = handle, sig
        self._handle_fin = weakref.finalize(self, _lib.lib().sedmi_basic_rcnn_destroy, handle)
        return handle

    # ---- what pipeline.SedPipeline needs from a model: output kind, shapes, and a forward into given buffers
    returns_logits = True               # the caller applies the sigmoid (training.py:358)
    time_pooling = 2

    @property
    def n_outputs(self):
        return self.num_classes

    @property
    def feature_width(self):
        return self.frequency_dim

    def _workspace_bytes(self, handle, B, T):
        return _lib.lib().sedmi_basic_rcnn_workspace_bytes(handle, B, T)

    def _forward_into(self, handle, x_ptr, B, T, out, ws, ws_bytes, stream, alphas=None):
        _lib.check(_lib.lib().sedmi_basic_rcnn_forward_f32(handle, x_ptr, B, T, _lib.ptr(out), _lib.ptr(ws), ws_bytes,
                                                           stream), "sedmi_basic_rcnn_forward_f32")

    def forward(self, x):
        if self.training:
            raise RuntimeError("inference-only model: call model.eval() first")
        if not x.is_cuda:
            raise _lib.SedmiError("forward needs a CUDA (ROCm) tensor: the model runs in HIP kernels only")
        if x.dim() != 3 or x.shape[2] != self.frequency_dim:
            raise ValueError(f"expected features [batch, frames, {self.frequency_dim}], got {tuple(x.shape)}")
        x = x.to(torch.float32).contiguous()                     # x.float() at models.py:48
        handle = self._get_handle(x.device)
        B, T, _ = x.shape
        lib = _lib.lib()
        logits = torch.empty((B, T // 2, self.num_classes), dtype=torch.float32, device=x.device)
        if B == 0 or T < 2:
            return logits
        ws_bytes = lib.sedmi_basic_rcnn_workspace_bytes(handle, B, T)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.sedmi_basic_rcnn_forward_f32(handle, _lib.ptr(x), B, T, _lib.ptr(logits), _lib.ptr(ws),
                                                        ws_bytes, _lib.stream_ptr(x.device)),
                       "sedmi_basic_rcnn_forward_f32")
        return logits


class ShakeShakeBlock(nn.Module):
    """Parameter container of models.py:70-100 (in == out, stride 1: identity shortcut)."""

    def __init__(self, in_channels, out_channels, stride=1):
        super().__init__()
        if stride != 1 or in_channels != out_channels:
            raise NotImplementedError("only the identity-shortcut ShakeShakeBlock the reference's models build")
        self.branch1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1)
        self.branch2 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1)


class ShakeCNNBlock(nn.Module):
    """Parameter container of models.py:155-180."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, avgpool_kernel=(1, 2),
                 avgpool_stride=1, dropout=0):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, 2 * out_channels, kernel_size=kernel_size, padding="same", stride=stride)
        self.bn1 = nn.BatchNorm2d(2 * out_channels)
        self.shake = ShakeShakeBlock(out_channels, out_channels)
        self.bn2 = nn.BatchNorm2d(out_channels)


class _ShakeBase(nn.Module):
    N_BINS = 768

    def _build_cnn(self, dropout):
        self.cnn1 = ShakeCNNBlock(1, 16, dropout=dropout, kernel_size=(5, 5), avgpool_kernel=(2, 2), avgpool_stride=2)
        self.cnn2 = ShakeCNNBlock(16, 32, dropout=dropout, avgpool_kernel=(2, 2), avgpool_stride=2)
        self.cnn3 = ShakeCNNBlock(32, 64, dropout=dropout, avgpool_kernel=(1, 2), avgpool_stride=(1, 2))
        self._handle = self._handle_sig = self._handle_fin = None

    def _signature(self):
        ts = list(self.parameters()) + list(self.buffers())
        return tuple((t.data_ptr(), t._version) for t in ts)

    def _release(self):
        if self._handle_fin is not None:
            self._handle_fin()
        self._handle = self._handle_sig = self._handle_fin = None

    def _host_state(self):
        return {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in self.state_dict().items()
                if v.is_floating_point()}

    @staticmethod
    def _fill_cnn(w, p):
        for i in range(3):
            b = f"cnn{i + 1}."
            w.conv_w[i], w.conv_b[i] = p(b + "conv.weight"), p(b + "conv.bias")
            w.bn1_w[i], w.bn1_b[i] = p(b + "bn1.weight"), p(b + "bn1.bias")
            w.bn1_mean[i], w.bn1_var[i] = p(b + "bn1.running_mean"), p(b + "bn1.running_var")
            w.br1_w[i], w.br1_b[i] = p(b + "shake.branch1.weight"), p(b + "shake.branch1.bias")
            w.br2_w[i], w.br2_b[i] = p(b + "shake.branch2.weight"), p(b + "shake.branch2.bias")
            w.bn2_w[i], w.bn2_b[i] = p(b + "bn2.weight"), p(b + "bn2.bias")
            w.bn2_mean[i], w.bn2_var[i] = p(b + "bn2.running_mean"), p(b + "bn2.running_var")

    returns_logits = True               # the caller applies the sigmoid (training.py:358)
    time_pooling = 4
    feature_width = N_BINS

    @property
    def n_outputs(self):
        return self.num_classes

    @staticmethod
    def draw_alphas(alphas=None):
        """Three ShakeShake mixing weights as a C float array: the given ones, or three
        ``torch.rand(1)`` draws from the CPU generator in block order (models.py:91)."""
        if alphas is None:
            alphas = [float(torch.rand(1)) for _ in range(3)]
        alphas = [float(a) for a in alphas]
        if len(alphas) != 3:
            raise ValueError("alphas: one ShakeShake mixing weight per block (3)")
        return (C.c_float * 3)(*alphas)

    def _prepare(self, x, alphas):
        if self.training:
            raise RuntimeError("inference-only model: call model.eval() first")
        if not x.is_cuda:
            raise _lib.SedmiError("forward needs a CUDA (ROCm) tensor: the model runs in HIP kernels only")
        if x.dim() != 3 or x.shape[2] != self.N_BINS:
            raise ValueError(f"expected features [batch, frames, {self.N_BINS}], got {tuple(x.shape)}")
        return x.to(torch.float32).contiguous(), self.draw_alphas(alphas)


class ShakeRCNN(_ShakeBase):
    def __init__(self, num_classes, dropout=0):
        super().__init__()
        self.num_classes = int(num_classes)
        self.dropout = nn.Dropout(dropout)
        self._build_cnn(dropout)
        self.flatten = nn.Flatten(start_dim=2)
        self.rnn = nn.GRU(6144, 1024, batch_first=True, bidirectional=True)
        self.layernorm = nn.LayerNorm(2048)
        self.fc1 = nn.Linear(2048, num_classes)
        self.sigmoid = nn.Sigmoid()
        self.softmax = nn.Softmax(dim=-1)
        self.relu = nn.LeakyReLU()
        self._recurrence = "auto"

    @property
    def recurrence(self):
        """"auto": the single-launch recurrence whenever the batch fits it; "per_step": always one launch per
        step (no requirement that the GPU be otherwise idle)."""
        return self._recurrence

    @recurrence.setter
    def recurrence(self, mode):
        if mode not in ("auto", "per_step"):
            raise ValueError('recurrence must be "auto" or "per_step"')
        self._recurrence = mode
        if self._handle is not None:
            self._apply_recurrence(self._handle)

    def _apply_recurrence(self, handle):
        mode = _lib.RECURRENCE_AUTO if self._recurrence == "auto" else _lib.RECURRENCE_PER_STEP
        _lib.check(_lib.lib().sedmi_shake_rcnn_set_recurrence(handle, mode), "sedmi_shake_rcnn_set_recurrence")

    def _check_run(self, handle, ws, ws_bytes, B, T, stream):
        """Waits for ``stream`` and raises SedmiRecurrenceError if the recurrence of the forward that used ``ws``
        gave up; the model is then switched to one launch per step so that the next forward cannot."""
        try:
            _lib.check(_lib.lib().sedmi_shake_rcnn_check(handle, _lib.ptr(ws), ws_bytes, B, T, stream),
                       "sedmi_shake_rcnn_check")
        except _lib.SedmiRecurrenceError:
            self.recurrence = "per_step"
            raise

    def _get_handle(self, device):
        sig = (device.index, self._signature())
        if self._handle is not None and self._handle_sig == sig:
            return self._handle
        self._release()
        sd = self._host_state()
        fp = C.POINTER(C.c_float)

        def p(key):
            return C.cast(C.c_void_p(sd[key].data_ptr()), fp)

        w = _lib.ShakeRcnnWeights()
        w.n_class = self.num_classes
        self._fill_cnn(w.cnn, p)
        for d, sfx in enumerate(("", "_reverse")):
            w.gru_w_ih[d], w.gru_w_hh[d] = p(f"rnn.weight_ih_l0{sfx}"), p(f"rnn.weight_hh_l0{sfx}")
            w.gru_b_ih[d], w.gru_b_hh[d] = p(f"rnn.bias_ih_l0{sfx}"), p(f"rnn.bias_hh_l0{sfx}")
        w.ln_w, w.ln_b = p("layernorm.weight"), p("layernorm.bias")
        w.fc_w, w.fc_b = p("fc1.weight"), p("fc1.bias")
        handle = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().sedmi_shake_rcnn_create(C.byref(handle), C.byref(w)), "sedmi_shake_rcnn_create")
        self._handle, self._handle_sig = handle, sig
        self._handle_fin = weakref.finalize(self, _lib.lib().sedmi_shake_rcnn_destroy, handle)
        self._apply_recurrence(handle)
        return handle

    def _workspace_bytes(self, handle, B, T):
        return _lib.lib().sedmi_shake_rcnn_workspace_bytes(handle, B, T)

    def _forward_into(self, handle, x_ptr, B, T, out, ws, ws_bytes, stream, alphas=None):
        _lib.check(_lib.lib().sedmi_shake_rcnn_forward_f32(handle, x_ptr, B, T, self.draw_alphas(alphas), _lib.ptr(out),
                                                           _lib.ptr(ws), ws_bytes, stream),
                   "sedmi_shake_rcnn_forward_f32")

    def forward(self, x, alphas=None):
        x, al = self._prepare(x, alphas)
        handle = self._get_handle(x.device)
        B, T, _ = x.shape
        lib = _lib.lib()
        logits = torch.empty((B, T // 4, self.num_classes), dtype=torch.float32, device=x.device)
        if B == 0 or T < 4:
            return logits
        ws_bytes = lib.sedmi_shake_rcnn_workspace_bytes(handle, B, T)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            st = _lib.stream_ptr(x.device)
            for attempt in (0, 1):
                _lib.check(lib.sedmi_shake_rcnn_forward_f32(handle, _lib.ptr(x), B, T, al, _lib.ptr(logits),
                                                            _lib.ptr(ws), ws_bytes, st), "sedmi_shake_rcnn_forward_f32")
                try:
                    self._check_run(handle, ws, ws_bytes, B, T, st)       # waits for the stream
                    break
                except _lib.SedmiRecurrenceError as e:
                    if attempt:
                        raise
                    import warnings
                    warnings.warn(f"{e}; running this forward again with one launch per step "
                                  f"(model.recurrence is now \"per_step\")", RuntimeWarning)
        return logits

    def forward_cnn(self, x, alphas=None):
        """The three ShakeCNNBlocks + transpose/flatten (models.py:126-134): [B, T/4, 6144]."""
        x, al = self._prepare(x, alphas)
        handle = self._get_handle(x.device)
        B, T, _ = x.shape
        lib = _lib.lib()
        seq = torch.empty((B, T // 4, 6144), dtype=torch.float32, device=x.device)
        if B == 0 or T < 4:
            return seq
        ws_bytes = lib.sedmi_shake_rcnn_workspace_bytes(handle, B, T)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.sedmi_shake_cnn_f32(handle, _lib.ptr(x), B, T, al, _lib.ptr(seq), _lib.ptr(ws), ws_bytes,
                                               _lib.stream_ptr(x.device)), "sedmi_shake_cnn_f32")
        return seq

    def get_weak_logits(self, x):
        # models.py:144-152, plain tensor ops on the (tiny) logits
        return torch.mean(self.softmax(x) * self.sigmoid(x), dim=1)


class ShakeTransformer(_ShakeBase):
    def __init__(self, num_classes, dropout):
        super().__init__()
        self.num_classes = int(num_classes)
        self.dropout = nn.Dropout(dropout)
        self._build_cnn(dropout)
        self.flatten = nn.Flatten(start_dim=2)
        self.fc1 = nn.Linear(6144, 1024)
        encoder_layer = nn.TransformerEncoderLayer(d_model=1024, nhead=8, dropout=dropout, batch_first=True)
        self.encoder = nn.TransformerEncoder(encoder_layer, num_layers=6)
        self.fc2 = nn.Linear(1024, num_classes)
        self.sigmoid = nn.Sigmoid()
        self.softmax = nn.Softmax(dim=-1)
        self.relu = nn.LeakyReLU()

    def _get_handle(self, device):
        sig = (device.index, self._signature())
        if self._handle is not None and self._handle_sig == sig:
            return self._handle
        self._release()
        sd = self._host_state()
        fp = C.POINTER(C.c_float)

        def p(key):
            return C.cast(C.c_void_p(sd[key].data_ptr()), fp)

        w = _lib.ShakeTransformerWeights()
        w.n_class, w.n_layers = self.num_classes, len(self.encoder.layers)
        self._fill_cnn(w.cnn, p)
        w.fc1_w, w.fc1_b, w.fc2_w, w.fc2_b = p("fc1.weight"), p("fc1.bias"), p("fc2.weight"), p("fc2.bias")
        for i in range(w.n_layers):
            b, l = f"encoder.layers.{i}.", w.layers[i]
            l.in_proj_w, l.in_proj_b = p(b + "self_attn.in_proj_weight"), p(b + "self_attn.in_proj_bias")
            l.out_proj_w, l.out_proj_b = p(b + "self_attn.out_proj.weight"), p(b + "self_attn.out_proj.bias")
            l.lin1_w, l.lin1_b = p(b + "linear1.weight"), p(b + "linear1.bias")
            l.lin2_w, l.lin2_b = p(b + "linear2.weight"), p(b + "linear2.bias")
            l.norm1_w, l.norm1_b = p(b + "norm1.weight"), p(b + "norm1.bias")
            l.norm2_w, l.norm2_b = p(b + "norm2.weight"), p(b + "norm2.bias")
        handle = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().sedmi_shake_transformer_create(C.byref(handle), C.byref(w)),
                       "sedmi_shake_transformer_create")
        self._handle, self._handle_sig = handle, sig
        self._handle_fin = weakref.finalize(self, _lib.lib().sedmi_shake_transformer_destroy, handle)
        return handle

    def _workspace_bytes(self, handle, B, T):
        return _lib.lib().sedmi_shake_transformer_workspace_bytes(handle, B, T)

    def _forward_into(self, handle, x_ptr, B, T, out, ws, ws_bytes, stream, alphas=None):
        _lib.check(_lib.lib().sedmi_shake_transformer_forward_f32(handle, x_ptr, B, T, self.draw_alphas(alphas),
                                                                  _lib.ptr(out), _lib.ptr(ws), ws_bytes, stream),
                   "sedmi_shake_transformer_forward_f32")

    def forward(self, x, alphas=None):
        x, al = self._prepare(x, alphas)
        handle = self._get_handle(x.device)
        B, T, _ = x.shape
        lib = _lib.lib()
        logits = torch.empty((B, T // 4, self.num_classes), dtype=torch.float32, device=x.device)
        if B == 0 or T < 4:
            return logits
        ws_bytes = lib.sedmi_shake_transformer_workspace_bytes(handle, B, T)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.sedmi_shake_transformer_forward_f32(handle, _lib.ptr(x), B, T, al, _lib.ptr(logits),
                                                               _lib.ptr(ws), ws_bytes, _lib.stream_ptr(x.device)),
                       "sedmi_shake_transformer_forward_f32")
        return logits
