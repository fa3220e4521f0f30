"""File-level prediction: audio files in, the reference's event table out.

Counterpart of the prediction loop of /root/reference/training.py:328-365, which
per file chops the audio into ``clip_length`` ms pieces (feature_extraction.py:35),
zero-pads the last one (:41-42), runs mel extraction, the model and a D2H copy
clip by clip (training.py:345-348), concatenates the frame predictions
(:350) and decodes them with the file's duration (:353-358).  Here whole files
are grouped into batches of clips that go through mel -> model -> decode on the GPU
in one pass (``SedPipeline``); a file's clips are decoded as one frame sequence
with the reference's cutoff ``int(duration / block) + 1`` (evaluation.py:58).
``model`` is any of the four ``--model`` choices (training.py:200-217);
``apply_sigmoid=None`` follows training.py:358 (False for the baseline CRNN, True
for the models that return logits).

``log`` / ``normalise`` switch on the two feature stages the reference does not
have (its ``--normalize`` flag is never read, training.py:91-94); both default to
off.  ``normalise=True`` without ``mean`` / ``std`` takes the statistics of the
files of the call: a first pass over all files sums them per mel bin in float64
(over all ranks of ``stats_group`` when one is given: one all-reduce of
2 * n_mels + 1 doubles), the second pass predicts with them.
"""
from __future__ import annotations

import numpy as np
import torch

from . import feature_extraction as fe
from .pipeline import SedPipeline


def _file_batches(test_files, dataset_location, dataset, sr, clip_s, loader, get_dur, batch_clips):
    """Groups of whole files whose clips fill a batch: yields lists of (file, duration, [clips])."""
    n_clip = int(sr * clip_s)
    group, n_pending = [], 0
    for test_file in test_files:
        path = dataset_location + dataset + '/' + test_file                      # training.py:352
        duration = get_dur(path)
        clips = []
        for i in range(int(np.ceil(duration / clip_s))):                         # feature_extraction.py:35
            clip = np.asarray(loader(path, sr, i * clip_s, clip_s), dtype=np.float32)
            if len(clip) < n_clip:
                clip = np.concatenate([clip, np.zeros(n_clip - len(clip), np.float32)])
            clips.append(clip[:n_clip])
        if not clips:
            continue
        if n_pending and n_pending + len(clips) > batch_clips:
            yield group
            group, n_pending = [], 0
        group.append((test_file.replace('\\', '/'), duration, clips))
        n_pending += len(clips)
    if group:
        yield group


def dataset_mel_statistics(test_files, dataset_location, dataset, clip_length, n_fft, n_mels, hop_length, win_length,
                           *, log=False, device=None, audio_loader=None, audio_duration=None, batch_clips=1024,
                           stats_group=None):
    """float64 (mean, std) per mel bin over every clip of ``test_files`` (the zero-padded tail of a file's last
    clip included: it is part of the feature the model sees), on ``device``.  With ``stats_group`` (a
    ``torch.distributed`` group, or True for the default one) the sums of all ranks are added first, so every rank
    gets the statistics of the union of the ranks' files."""
    from . import parallel
    sr = fe.sampling_rates[dataset]
    clip_s = 0.001 * clip_length
    device = torch.device(device if device is not None else "cuda")
    sums = torch.zeros(2 * n_mels, dtype=torch.float64, device=device)
    count = 0.0
    for group in _file_batches(test_files, dataset_location, dataset, sr, clip_s, audio_loader or fe._load_wav,
                               audio_duration or fe._wav_duration, batch_clips):
        y = torch.from_numpy(np.stack([c for _, _, clips in group for c in clips])).to(device)
        mel = fe.melspectrogram(y, sr, n_fft, n_mels, hop_length, win_length, log=log)
        s, s2, n = fe.mel_statistics(mel)
        sums[:n_mels] += s
        sums[n_mels:] += s2
        count += n
    if stats_group is not None:
        return parallel.all_reduce_mel_stats(sums[:n_mels], sums[n_mels:], count,
                                             group=None if stats_group is True else stats_group)
    if count == 0:
        raise ValueError("no clips to take statistics from")
    return parallel.mel_stats_to_mean_std(sums[:n_mels], sums[n_mels:], count)


def predict_files(model, test_files, dataset_location, dataset, clip_length, n_fft, n_mels, hop_length, win_length,
                  id2cls, block_length, decision_threshold=0.5, apply_sigmoid=None, *, device=None,
                  audio_loader=None, audio_duration=None, batch_clips=1024, median_window=1, return_probs=False,
                  alphas=None, log=False, normalise=False, mean=None, std=None, stats_group=None):
    """Events of every file as ``(file, onset_s, offset_s, label)`` tuples with
    float64 times, files in input order, class-major inside a file
    (evaluation.py:61-78).  With ``return_probs`` also {file: ndarray
    [clips * frames, n_cls]} of the concatenated frame predictions (what the model
    returns: probabilities for the CRNN, logits otherwise).  ``alphas``: fixed
    ShakeShake mixing weights for the Shake models (default: drawn per batch with
    ``torch.rand(1)`` as the reference's eval mode does, models.py:88-91)."""
    sr = fe.sampling_rates[dataset]
    clip_s = 0.001 * clip_length
    device = torch.device(device if device is not None else "cuda")
    loader = audio_loader or fe._load_wav
    get_dur = audio_duration or fe._wav_duration
    if normalise and mean is None:
        mean, std = dataset_mel_statistics(test_files, dataset_location, dataset, clip_length, n_fft,
                                           int(model.feature_width), hop_length, win_length, log=log, device=device,
                                           audio_loader=audio_loader, audio_duration=audio_duration,
                                           batch_clips=batch_clips, stats_group=stats_group)
    pipe = SedPipeline(model, sr=sr, n_fft=n_fft, n_mels=n_mels, hop_length=hop_length, win_length=win_length,
                       block_length=block_length, decision_threshold=decision_threshold,
                       apply_sigmoid=apply_sigmoid, median_window=median_window, log=log, normalise=normalise,
                       mean=mean, std=std)
    block_s = float(block_length) / 1000
    predictions, probs_out = [], {}
    for group in _file_batches(test_files, dataset_location, dataset, sr, clip_s, loader, get_dur, batch_clips):
        y = torch.from_numpy(np.stack([c for _, _, clips in group for c in clips])).to(device)
        counts = [len(clips) for _, _, clips in group]
        cutoffs = [int(dur / block_s) + 1 for _, dur, _ in group]              # evaluation.py:58
        out = pipe.run(y, clips_per_recording=counts, cutoffs=cutoffs, alphas=alphas)
        names = [f for f, _, _ in group]
        predictions.extend(pipe.events(out, id2cls, names))
        if return_probs:
            p = out["probs"].cpu().numpy()
            o = 0
            for f, n in zip(names, counts):
                probs_out[f] = p[o:o + n].reshape(-1, p.shape[-1])
                o += n
    return (predictions, probs_out) if return_probs else predictions


def write_predictions_tsv(predictions, path):
    """The table of training.py:364-365: header ``filename onset offset event_label``, tab separated."""
    import pandas as pd
    pd.DataFrame(predictions, columns=['filename', 'onset', 'offset', 'event_label']).to_csv(path, sep='\t', index=False)
