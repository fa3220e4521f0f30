"""Thresholding, scoring and event decoding on MI355X behind the reference's
signatures.

Mirrors /root/reference/evaluation.py: ``evaluate`` (:5-47) and
``get_prediction_from_raw_output`` (:49-80).  The arithmetic runs in the HIP
kernels of csrc/decode.hip through the C ABI (``sedmi_decode_events``,
``sedmi_threshold_i64``, ``sedmi_confusion_counts``).  ``evaluate_single_label``
(:83-143) is not provided: its only caller in the reference cannot be imported.

Additions underneath the reference's interface:

``decode_recordings``  — decodes many recordings in one launch (the reference
    loops over files in Python, training.py:335-360);
``threshold``          — the int64 0/1 map of evaluation.py:21 on device;
``median_window``      — optional odd smoothing window, default 1 = off (the
    reference has no median filter).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _as_device_f32(t, device=None) -> torch.Tensor:
    if not torch.is_tensor(t):
        t = torch.as_tensor(np.asarray(t))
    if not t.is_cuda:
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        t = t.to(device)
    return t.to(torch.float32).contiguous()


def threshold(pred: torch.Tensor, decision_threshold: float = 0.5, apply_sigmoid: bool = False) -> torch.Tensor:
    """``torch.where(sigmoid?(pred) > thr, 1, 0)`` as int64, on the device."""
    p = _as_device_f32(pred)
    out = torch.empty(p.shape, dtype=torch.int64, device=p.device)
    with torch.cuda.device(p.device):
        _lib.check(_lib.lib().sedmi_threshold_i64(_lib.ptr(p), p.numel(), float(decision_threshold),
                                                  int(bool(apply_sigmoid)), _lib.ptr(out),
                                                  _lib.stream_ptr(p.device)), "sedmi_threshold_i64")
    return out


def decode_recordings(pred: torch.Tensor, seg_lengths, cutoffs, decision_threshold: float = 0.5,
                      apply_sigmoid: bool = False, median_window: int = 1, capacity: int | None = None):
    """Run-length decode of several recordings stored back to back.

    pred [total_frames, n_cls]; recording s owns ``seg_lengths[s]`` frames and
    keeps the first ``cutoffs[s]``.  Returns int32 numpy arrays
    (seg, cls, onset_frame, offset_frame) ordered by recording, class, onset —
    offsets exclusive and relative to the recording."""
    p = _as_device_f32(pred)
    if p.dim() == 1:
        p = p.unsqueeze(1)
    total, n_cls = p.shape
    seg_lengths = np.asarray(seg_lengths, dtype=np.int64)
    n_seg = len(seg_lengths)
    starts = np.zeros(n_seg + 1, dtype=np.int64)
    np.cumsum(seg_lengths, out=starts[1:])
    if starts[-1] != total:
        raise ValueError(f"segment lengths sum to {starts[-1]}, prediction has {total} frames")
    dev = p.device
    d_start = torch.from_numpy(starts).to(dev)
    d_cut = torch.from_numpy(np.asarray(cutoffs, dtype=np.int64)).to(dev)
    if capacity is None:
        capacity = total * n_cls // 2 + n_seg * n_cls + 1     # a run needs >= 1 on-frame and a gap
    ev = torch.empty((4, max(capacity, 1)), dtype=torch.int32, device=dev)
    n_ev = torch.zeros(1, dtype=torch.int64, device=dev)
    lib = _lib.lib()
    ws_bytes = lib.sedmi_decode_workspace_bytes(n_seg, n_cls)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.sedmi_decode_events(_lib.ptr(p), total, n_cls, _lib.ptr(d_start), _lib.ptr(d_cut), n_seg,
                                           float(decision_threshold), int(bool(apply_sigmoid)), int(median_window),
                                           _lib.ptr(ev[0]), _lib.ptr(ev[1]), _lib.ptr(ev[2]), _lib.ptr(ev[3]),
                                           capacity, _lib.ptr(n_ev), _lib.ptr(ws), ws_bytes,
                                           _lib.stream_ptr(dev)), "sedmi_decode_events")
    n = int(n_ev.item())
    if n > capacity:
        raise _lib.SedmiError(f"{n} events exceed capacity {capacity}")
    host = ev[:, :n].cpu().numpy()
    return host[0], host[1], host[2], host[3]


def get_prediction_from_raw_output(raw_prediction, id2cls, audio_duration, block_length, file_name,
                                   decision_threshold=0.5, apply_sigmoid=False, median_window=1):
    """Same contract as evaluation.py:49-80: list of
    ``(file_name, onset_s, offset_s, class_name)`` with np.float64 times equal
    to ``frame_index * (float(block_length) / 1000)``, class-major order."""
    p = _as_device_f32(raw_prediction)
    if p.dim() == 1:                       # n_cls == 1 after the caller's squeeze() (evaluation.py:62-63)
        p = p.unsqueeze(1)
    n_cls = p.shape[1]
    block_s = float(block_length) / 1000
    cutoff = int(audio_duration / block_s) + 1
    _, cls, on, off = decode_recordings(p, [p.shape[0]], [cutoff], decision_threshold, apply_sigmoid,
                                        median_window)
    keys = list(id2cls.keys())
    if keys != list(range(n_cls)):
        # the reference walks id2cls in dict order and reads column `idx` for each key
        order = {k: i for i, k in enumerate(keys)}
        sel = sorted((i for i, c in enumerate(cls) if int(c) in order),
                     key=lambda i: (order[int(cls[i])], int(on[i])))
        cls, on, off = cls[sel], on[sel], off[sel]
    found_events = []
    for c, s, e in zip(cls, on, off):
        found_events.append((file_name, np.int64(s) * block_s, np.int64(e) * block_s, id2cls[int(c)]))
    return found_events


def evaluate(model, device, data_loader, id2cls, decision_threshold=0.5, apply_sigmoid=True):
    """Same contract as evaluation.py:5-47.  Predictions are thresholded and
    reduced to per-class TP/FP/FN/TN on the device; precision / recall / F1 /
    accuracy follow from the counts exactly as sklearn defines them with
    ``zero_division=0.0``."""
    model.to(device)
    model.eval()
    num_classes = len(id2cls.keys())
    dev = torch.device(device)
    counts = torch.zeros((num_classes, 4), dtype=torch.int64, device=dev)
    lib = _lib.lib()
    with torch.no_grad():
        for features, labels in data_loader:
            if type(features) is list:
                features = [f.to(dev) for f in features]
            else:
                features = features.to(dev)
            pred = model(features).to(torch.float32).contiguous().view(-1, num_classes)
            lab = labels.to(dev, torch.float32).contiguous().view(-1, num_classes)
            with torch.cuda.device(dev):
                _lib.check(lib.sedmi_confusion_counts(_lib.ptr(pred), _lib.ptr(lab), pred.shape[0], num_classes,
                                                      float(decision_threshold), int(bool(apply_sigmoid)),
                                                      _lib.ptr(counts), _lib.stream_ptr(dev)),
                           "sedmi_confusion_counts")
    tp, fp, fn, tn = (counts[:, k].cpu().numpy().astype(np.float64) for k in range(4))

    def prf(tp, fp, fn):
        p = tp / (tp + fp) if tp + fp > 0 else 0.0
        r = tp / (tp + fn) if tp + fn > 0 else 0.0
        f = 2 * tp / (2 * tp + fp + fn) if 2 * tp + fp + fn > 0 else 0.0
        return float(p), float(r), float(f)

    class_results = {}
    for idx, cls in id2cls.items():
        col = 0 if num_classes == 1 else idx
        p, r, f = prf(tp[col], fp[col], fn[col])
        class_results[cls] = {'f1': f, 'precision': p, 'recall': r}
    p, r, f = prf(tp.sum(), fp.sum(), fn.sum())
    total = tp.sum() + fp.sum() + fn.sum() + tn.sum()
    results = {'recall': r, 'precision': p, 'f1': f,
               'accuracy': float((tp.sum() + tn.sum()) / total) if total > 0 else 0.0}
    return results, class_results
