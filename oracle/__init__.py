"""CPU oracle for the SED inference hot path.  TEST INFRASTRUCTURE ONLY.

A plain numpy / torch-CPU restatement of what the reference computes on the
path  feature_extraction.py:16-70 -> nnet/CRNN.py:191-268 -> evaluation.py:5-80.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker: the shipped
package (``sound_event_detection_amd``) never imports it and has no CPU
fallback.

How each part is pinned
-----------------------
* ``oracle.decode``  — restates evaluation.py:49-80.  Pinned against the
  reference itself (``oracle/make_golden.py`` imports evaluation.py and stores
  its outputs in tests/golden/decode_*.json) and against the float64 onsets
  the reference's own result files hold
  (BirdSED_results/dcase_baseline/test_predictions.tsv:3-4,14).
* ``oracle.crnn``    — restates nnet/CNN.py:33-114, nnet/RNN.py:7-30 and the
  forward glue of nnet/CRNN.py:191-268,141-143.  The CNN and the
  BidirectionalGRU are pinned against the reference classes imported from
  /root/reference (tests/golden/crnn_*.npz).  nnet/CRNN.py itself cannot be
  imported here (it imports torchaudio, which is not installed), so the few
  glue lines between those two modules and the strong head are restated, not
  executed: for that glue alone, parity unpinned.
* ``oracle.mel``     — the arithmetic lives in librosa (version not pinned by
  the reference; ``librosa.get_duration(path=...)`` at
  feature_extraction.py:33 implies >= 0.10), which is not installed and whose
  sources are not on disk.  The restatement follows librosa 0.10's published
  algorithm for ``feature.melspectrogram`` / ``stft`` / ``filters.mel``; the
  reference holds no fixture for it, so: parity unpinned.  It is cross-checked
  in tests against two independent implementations that are installed
  (``torch.stft`` and ``transformers.audio_utils``).
* the optional stages the reference does not have (log, mean/std
  normalisation, median filter) are defined here by numpy/scipy and are
  parity unpinned by construction; they default to off everywhere.
"""
