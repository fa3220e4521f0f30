"""The whole hot path as one object: audio clips on the GPU in, frame
predictions and decoded events out.

This is the batched counterpart of the reference's file-level prediction loop
(training.py:333-365), which runs mel extraction, the model and the decoder one
clip at a time with two PCIe copies per clip.  Here a batch of clips stays in
HBM from the first kernel to the last: mel frontend -> model -> threshold /
run-length decode, all enqueued on the current stream through the C ABI with
buffers that are allocated once and reused (a steady-state step allocates
nothing and never synchronises with the host).

Any of the four ``--model`` choices of training.py:200-217 can sit in the
middle: the baseline ``nnet.CRNN`` (returns probabilities), ``models.BasicRCNN``,
``models.ShakeRCNN`` and ``models.ShakeTransformer`` (return logits; the decoder
then applies the sigmoid, ``apply_sigmoid = (model != 'Baseline')``,
training.py:358).  The mel width follows the model: 128 for the CRNN,
``frequency_dim`` for BasicRCNN, 768 for the Shake models (the width their GRU /
``fc1`` input of 6144 = 64 * 768 / 8 dictates, models.py:114,194).

Two feature stages the reference does not have (its ``--normalize`` flag is
parsed and never read, training.py:91-94) can be switched on; both default to
off, which is the reference's behaviour:

``log=True``        10 * log10(max(S, 1e-10)) on the mel spectrogram;
``normalise=True``  per-mel-bin (x - mean) / std.  With ``mean`` / ``std`` given
                    (dataset-level statistics computed beforehand) the stage
                    rides in the mel kernel.  Without them the statistics are
                    those of the clips of the call itself, over all ranks of
                    ``group`` when one is given: mel pass -> float64 sums per bin
                    (``sedmi_mel_stats_f64``) -> one all-reduce of 2 * n_mels + 1
                    doubles (``parallel.all_reduce_mel_stats``) -> in-place
                    normalisation (``sedmi_mel_apply_f32``) -> model.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from . import feature_extraction as fe


class SedPipeline:
    def __init__(self, model, *, sr=16000, n_fft=2048, n_mels=None, hop_length=320, win_length=640,
                 block_length=40, decision_threshold=0.5, apply_sigmoid=None, median_window=1,
                 log=False, normalise=False, mean=None, std=None, group=None):
        """``model``: one of the package's model classes in eval mode on a CUDA
        device.  ``n_mels`` defaults to the model's feature width and must equal
        it.  ``apply_sigmoid`` defaults to what training.py:358 passes: False
        for the baseline CRNN (it already returns probabilities), True for the
        models that return logits.  ``log`` / ``normalise`` / ``mean`` / ``std`` /
        ``group``: the optional feature stages described in the module docstring
        (``group``: a ``torch.distributed`` process group, or True for the
        default group; None = this process's clips only)."""
        for attr in ("returns_logits", "time_pooling", "feature_width", "n_outputs", "_forward_into", "_get_handle"):
            if not hasattr(model, attr):
                raise TypeError(f"{type(model).__name__} is not one of this package's models (no {attr})")
        self.model = model
        width = int(model.feature_width)
        if n_mels is not None and int(n_mels) != width:
            raise ValueError(f"{type(model).__name__} takes features of width {width}, not n_mels={n_mels}")
        self.sr, self.n_fft, self.n_mels = int(sr), int(n_fft), width
        self.hop, self.win = int(hop_length), int(win_length)
        self.block_length = block_length
        self.thr = float(decision_threshold)
        self.apply_sigmoid = bool(model.returns_logits if apply_sigmoid is None else apply_sigmoid)
        self.median_window = int(median_window)
        self.log, self.normalise = bool(log), bool(normalise)
        if (mean is None) != (std is None):
            raise ValueError("mean and std must be given together")
        if mean is not None and not self.normalise:
            raise ValueError("mean / std are only used with normalise=True")
        self.mean, self.std, self.group = mean, std, group
        self._fixed_stats = None        # (device, mean f32, 1/std f32) of user-supplied statistics
        self._buf = None

    # ---------------------------------------------------------------- buffers
    def _buffers(self, B, T, device, with_mel):
        """Shape-dependent tensors only, sized for the largest batch seen so far
        with this frame count: smaller batches use a prefix of each buffer.  No
        model handle is cached here: it is asked for on every run, so that new
        weights (``load_state_dict``), a device move or a changed
        ``conv_precision`` take effect at once."""
        b = self._buf
        n_cls = int(self.model.n_outputs)
        if (b is not None and b["T"] == T and b["device"] == device and b["cap_B"] >= B and b["n_cls"] == n_cls
                and (b["mel"] is not None or not with_mel)):
            return b
        cap_B = B if b is None or b["T"] != T or b["device"] != device else max(B, b["cap_B"])
        Tp = T // int(self.model.time_pooling)
        total = cap_B * Tp
        cap_ev = total * n_cls // 2 + cap_B * n_cls + 1          # a run needs >= 1 on-frame and a gap
        dws_bytes = max(_lib.lib().sedmi_decode_workspace_bytes(cap_B, n_cls), 1)
        self._buf = b = dict(
            T=T, Tp=Tp, n_cls=n_cls, device=device, cap_B=cap_B, cap_ev=cap_ev,
            mel=torch.empty((cap_B, self.n_mels, T), dtype=torch.float32, device=device) if with_mel else None,
            out=torch.empty((cap_B, Tp, n_cls), dtype=torch.float32, device=device),
            ws=None, ws_bytes=0,
            dws=torch.empty(dws_bytes, dtype=torch.uint8, device=device), dws_bytes=dws_bytes,
            seg_start=torch.zeros(cap_B + 1, dtype=torch.int64, device=device),
            seg_cut=torch.zeros(cap_B, dtype=torch.int64, device=device),
            seg_key=None,
            ev=torch.empty((4, cap_ev), dtype=torch.int32, device=device),
            n_events=torch.zeros(1, dtype=torch.int64, device=device),
        )
        return b

    def _segments(self, b, B, clips_per_recording, cutoffs):
        """Uploads the recording table (frame ranges and cutoffs) when it differs
        from the one already on the device."""
        if clips_per_recording is None:
            clips_per_recording = (1,) * B
        clips_per_recording = tuple(int(c) for c in clips_per_recording)
        if any(c <= 0 for c in clips_per_recording):
            raise ValueError("every recording must own at least one clip (clips_per_recording has an entry <= 0)")
        if sum(clips_per_recording) != B:
            raise ValueError("clips_per_recording must sum to the number of clips")
        n_seg = len(clips_per_recording)
        seg_len = np.asarray(clips_per_recording, dtype=np.int64) * b["Tp"]
        if cutoffs is None:        # whole recording: int(duration / block) + 1 >= its frame count
            cutoffs = seg_len + 1
        cutoffs = tuple(int(c) for c in cutoffs)
        if len(cutoffs) != n_seg:
            raise ValueError("one cutoff per recording")
        key = (clips_per_recording, cutoffs)
        if b["seg_key"] != key:
            starts = np.zeros(n_seg + 1, dtype=np.int64)
            np.cumsum(seg_len, out=starts[1:])
            b["seg_start"][:n_seg + 1].copy_(torch.from_numpy(starts), non_blocking=False)
            b["seg_cut"][:n_seg].copy_(torch.from_numpy(np.asarray(cutoffs, dtype=np.int64)), non_blocking=False)
            b["seg_key"] = key
        return n_seg

    # -------------------------------------------------------------------- run
    def _model_and_decode(self, b, feats_ptr, B, T, dev, clips_per_recording, cutoffs, alphas):
        lib = _lib.lib()
        model = self.model
        handle = model._get_handle(dev)                     # cached by the model, keyed on its parameters
        n_seg = self._segments(b, B, clips_per_recording, cutoffs)
        ws_bytes = max(model._workspace_bytes(handle, B, T), 1)
        if b["ws"] is None or b["ws_bytes"] < ws_bytes:
            b["ws"] = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            b["ws_bytes"] = ws_bytes
        out = b["out"][:B]
        with torch.cuda.device(dev):
            st = _lib.stream_ptr(dev)
            model._forward_into(handle, feats_ptr, B, T, out, b["ws"], b["ws_bytes"], st, alphas)
            ev = b["ev"]
            _lib.check(lib.sedmi_decode_events(_lib.ptr(out), B * b["Tp"], b["n_cls"], _lib.ptr(b["seg_start"]),
                                               _lib.ptr(b["seg_cut"]), n_seg, self.thr, int(self.apply_sigmoid),
                                               self.median_window, _lib.ptr(ev[0]), _lib.ptr(ev[1]), _lib.ptr(ev[2]),
                                               _lib.ptr(ev[3]), b["cap_ev"], _lib.ptr(b["n_events"]),
                                               _lib.ptr(b["dws"]), b["dws_bytes"], st), "sedmi_decode_events")
        res = dict(b)
        res["probs"] = out       # probabilities (CRNN) or logits (the other models): what the model returns
        res["n_seg"] = n_seg
        # models whose kernels can report a failed run after the fact (ShakeRCNN's single-launch recurrence)
        res["_check"] = (handle, B, T, dev) if hasattr(model, "_check_run") else None
        return res

    def check(self, b):
        """Waits for the run that produced ``b`` and raises if one of its kernels reported that it could not
        finish (``SedmiRecurrenceError``: ShakeRCNN's single-launch recurrence gave up because the GPU was shared;
        the run's predictions are NaN and its event list is empty -- never act on them).  ``events`` calls this;
        callers that read ``b["probs"]`` themselves call it first."""
        c = b.get("_check")
        if c is not None:
            handle, B, T, dev = c
            with torch.cuda.device(dev):
                self.model._check_run(handle, b["ws"], b["ws_bytes"], B, T, _lib.stream_ptr(dev))

    def run(self, y: torch.Tensor, clips_per_recording=None, cutoffs=None, alphas=None):
        """y: float32 (or int16 PCM, taken as sample / 32768) CUDA tensor [B, N] of equally long clips; consecutive clips
        of one recording are decoded as one frame sequence (training.py:350).
        ``alphas``: the three ShakeShake mixing weights for the Shake models
        (drawn as the reference draws them when None).
        Returns a dict: ``probs`` [B, T', n_cls] (the model's output), ``ev``
        int32 [4, cap] (recording, class, onset frame, offset frame) and
        ``n_events`` [1], all on the device and valid once the stream has
        drained; they are overwritten by the next run."""
        if not y.is_cuda or y.dtype not in (torch.float32, torch.int16) or y.dim() != 2 or not y.is_contiguous():
            raise ValueError("y must be a contiguous float32 or int16 CUDA tensor [clips, samples]")
        B, N = y.shape
        dev = y.device
        plan = fe.get_plan(self.sr, self.n_fft, self.win, self.hop, self.n_mels, dev)
        T = plan.num_frames(N)
        b = self._buffers(B, T, dev, True)
        mel = b["mel"][:B]
        entry = "sedmi_melspec_i16" if y.dtype == torch.int16 else "sedmi_melspec_f32"
        lib = _lib.lib()
        flags = _lib.SEDMI_MEL_LOG if self.log else 0
        mean = istd = stats = None
        if self.normalise and self.mean is not None:        # dataset-level statistics known beforehand: one pass
            if self._fixed_stats is None or self._fixed_stats[0] != dev:
                m32 = torch.as_tensor(self.mean).to(dev, torch.float32).contiguous()
                i32 = (1.0 / torch.as_tensor(self.std).to(dev, torch.float64)).to(torch.float32).contiguous()
                if m32.numel() != self.n_mels or i32.numel() != self.n_mels:
                    raise ValueError(f"mean / std must have {self.n_mels} entries")
                self._fixed_stats = (dev, m32, i32)
            _, mean, istd = self._fixed_stats
            flags |= _lib.SEDMI_MEL_NORMALISE
        with torch.cuda.device(dev):
            st = _lib.stream_ptr(dev)
            _lib.check(getattr(lib, entry)(plan.handle, _lib.ptr(y), B, N, flags, _lib.ptr(mean), _lib.ptr(istd),
                                           _lib.ptr(mel), st), entry)
            if self.normalise and self.mean is None:        # statistics of this call's clips: two passes
                stats = self._batch_statistics(mel, B, T, dev)
                _lib.check(lib.sedmi_mel_apply_f32(_lib.ptr(mel), B, self.n_mels, T, _lib.SEDMI_MEL_NORMALISE,
                                                   _lib.ptr(stats[2]), _lib.ptr(stats[3]), st), "sedmi_mel_apply_f32")
        # the [B, n_mels, T] bytes are the [B, T, n_mels] feature (feature_extraction.py:49-50)
        res = self._model_and_decode(b, _lib.ptr(mel), B, T, dev, clips_per_recording, cutoffs, alphas)
        res["mel"] = mel
        if stats is not None:
            res["mel_mean"], res["mel_std"] = stats[0], stats[1]        # float64 [n_mels], on the device
        return res

    def _batch_statistics(self, mel, B, T, dev):
        """float64 (mean, std) per mel bin of ``mel`` over the clips of every rank of ``self.group``, plus the
        float32 (mean, 1 / std) the normalisation kernel reads.  Everything stays on the device; with a group
        this is the path's one all-reduce (2 * n_mels + 1 doubles)."""
        from . import parallel
        M = self.n_mels
        s = torch.zeros(2 * M, dtype=torch.float64, device=dev)
        _lib.check(_lib.lib().sedmi_mel_stats_f64(_lib.ptr(mel), B, M, T, _lib.ptr(s), _lib.ptr(s[M:]),
                                                  _lib.stream_ptr(dev)), "sedmi_mel_stats_f64")
        group = None if self.group is True else self.group
        if self.group is not None:
            mean, std = parallel.all_reduce_mel_stats(s[:M], s[M:], float(B * T), group=group)
        else:
            mean, std = parallel.mel_stats_to_mean_std(s[:M], s[M:], float(B * T))
        return mean, std, mean.to(torch.float32), (1.0 / std).to(torch.float32)

    def run_features(self, feats: torch.Tensor, clips_per_recording=None, cutoffs=None, alphas=None):
        """The same from precomputed features [B, T, width] (float32, CUDA,
        contiguous), e.g. the AST embeddings the reference trained its Shake
        models on (feature_extraction.py:52-58) or a feature cache."""
        if (not feats.is_cuda or feats.dtype != torch.float32 or feats.dim() != 3 or not feats.is_contiguous()
                or feats.shape[2] != self.n_mels):
            raise ValueError(f"feats must be a contiguous float32 CUDA tensor [clips, frames, {self.n_mels}]")
        B, T, _ = feats.shape
        b = self._buffers(B, T, feats.device, False)
        return self._model_and_decode(b, _lib.ptr(feats), B, T, feats.device, clips_per_recording, cutoffs, alphas)

    def events(self, b, id2cls, names):
        """Host-side event list of a finished run, in the reference's format:
        (name, onset_s, offset_s, class) with float64 times (evaluation.py:75-78)."""
        self.check(b)
        n = int(b["n_events"].item())
        if n > b["cap_ev"]:
            raise _lib.SedmiError(f"{n} events exceed the event buffer's capacity {b['cap_ev']}")
        host = b["ev"][:, :n].cpu().numpy()
        block_s = float(self.block_length) / 1000
        return [(names[int(s)], np.int64(on) * block_s, np.int64(off) * block_s, id2cls[int(c)])
                for s, c, on, off in zip(*host)]
