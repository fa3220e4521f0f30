"""Acceptance rule for decoded events.  TEST INFRASTRUCTURE ONLY.

The two implementations being compared (the CPU path of the reference, restated
in this package's oracle, and the HIP path) produce frame predictions that may
differ by float32 rounding.  Thresholding (evaluation.py:52, strict ``>``) turns
a difference of 1e-7 into a different bit when the value sits on the threshold,
so events are compared under this rule:

* every frame whose thresholded bit differs between the two must have its CPU
  probability within ``tol`` of the threshold (|p_cpu - thr| <= tol) — a
  differing frame further away is an out-of-tolerance difference (must be 0);
* with the frames within ``tol`` of the threshold masked (forced to the same
  value on both sides), the two event lists must be identical.
"""
from __future__ import annotations

import numpy as np

from . import decode as odec


def frame_differences(p_cpu: np.ndarray, p_gpu: np.ndarray, thr: float = 0.5, tol: float = 1e-3,
                      apply_sigmoid: bool = False):
    """p_cpu, p_gpu: [frames, n_cls] model outputs (logits when ``apply_sigmoid``).
    Returns (n_differing_frames, n_out_of_tolerance, near_mask) where near_mask marks
    the frames whose CPU probability is within ``tol`` of the threshold."""
    q_cpu = odec.sigmoid32(p_cpu) if apply_sigmoid else np.asarray(p_cpu, np.float32)
    q_gpu = odec.sigmoid32(p_gpu) if apply_sigmoid else np.asarray(p_gpu, np.float32)
    differ = odec.threshold(q_cpu, thr) != odec.threshold(q_gpu, thr)
    near = np.abs(q_cpu.astype(np.float64) - thr) <= tol
    return int(differ.sum()), int((differ & ~near).sum()), near


def masked(p: np.ndarray, near: np.ndarray, thr: float, apply_sigmoid: bool) -> np.ndarray:
    """Copy of ``p`` with the frames of ``near`` forced below the threshold."""
    out = p.copy()
    out[near] = -30.0 if apply_sigmoid else min(0.0, thr - 1.0)
    return out


def cpu_events(p: np.ndarray, n_clips_per_recording, frames_per_clip: int, names, id2cls, durations, block_length,
               thr: float = 0.5, apply_sigmoid: bool = False):
    """The reference's decoder, recording by recording (training.py:350-358)."""
    out, o = [], 0
    for name, n, dur in zip(names, n_clips_per_recording, durations):
        rows = n * frames_per_clip
        out += odec.get_prediction_from_raw_output(p[o:o + rows], id2cls, dur, block_length, name, thr, apply_sigmoid)
        o += rows
    return out
