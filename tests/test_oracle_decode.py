"""The decode oracle against the reference's own outputs (CPU only).

tests/golden/decode_*.{npz,json} were produced by running evaluation.py:49-80
(oracle/make_golden.py); the two .tsv files are result tables the reference
ships (training.py:364-365 wrote them)."""
import json
import os

import numpy as np
import pytest

from oracle import decode


def _cases(golden_dir):
    inputs = np.load(os.path.join(golden_dir, "decode_inputs.npz"))
    with open(os.path.join(golden_dir, "decode_expected.json")) as f:
        meta = json.load(f)
    return inputs, meta


def test_decode_matches_reference_outputs(golden_dir):
    inputs, meta = _cases(golden_dir)
    assert len(meta) >= 12
    for m in meta:
        id2cls = {i: f"cls{i}" for i in range(m["n_cls"])}
        got = decode.get_prediction_from_raw_output(
            inputs[m["name"]], id2cls, m["audio_duration"], m["block_length"], "file.wav",
            decision_threshold=m["thr"], apply_sigmoid=m["apply_sigmoid"])
        want = [(f, float.fromhex(a), float.fromhex(b), c) for f, a, b, c in m["events"]]
        assert len(got) == len(want), m["name"]
        for g, w in zip(got, want):
            assert g[0] == w[0] and g[3] == w[3], m["name"]
            # bit-exact float64 times
            assert np.float64(g[1]).tobytes() == np.float64(w[1]).tobytes(), m["name"]
            assert np.float64(g[2]).tobytes() == np.float64(w[2]).tobytes(), m["name"]
            assert isinstance(g[1], np.float64) and isinstance(g[2], np.float64)


@pytest.mark.parametrize("name", ["ref_tsv_birdsed_dcase_baseline.tsv", "ref_tsv_tut_dcase_baseline_fold1.tsv"])
def test_time_arithmetic_matches_reference_result_tables(golden_dir, name):
    """Every onset/offset the reference wrote is frame_index * 0.04 evaluated
    in float64 — e.g. 166 * 0.04 prints as 6.640000000000001."""
    with open(os.path.join(golden_dir, name)) as f:
        rows = [line.rstrip("\n").split("\t") for line in f]
    assert rows[0] == ["filename", "onset", "offset", "event_label"]
    bl = float(40) / 1000
    seen = {}
    for fn, on, off, cls in rows[1:]:
        for txt in (on, off):
            k = int(round(float(txt) / bl))
            assert repr(float(np.int64(k) * bl)) == txt
        # within one (file, class) the events are ordered and disjoint
        prev = seen.get((fn, cls))
        assert prev is None or float(on) > prev
        seen[(fn, cls)] = float(off)
    assert repr(float(np.int64(166) * bl)) == "6.640000000000001"


def test_threshold_is_strict_and_int64():
    p = np.array([[0.5, 0.50000006, 0.49999997]], np.float32)
    b = decode.threshold(p, 0.5)
    assert b.dtype == np.int64 and b.tolist() == [[0, 1, 0]]


def test_cutoff_rule():
    assert decode.cutoff_frames(10.0, 40) == 251
    assert decode.cutoff_frames(7.3, 40) == int(7.3 / 0.04) + 1


def test_median_matches_majority_vote_after_threshold():
    rng = np.random.Generator(np.random.PCG64(5))
    p = rng.uniform(0, 1, (300, 4)).astype(np.float32)
    for w in (3, 5, 9):
        a = decode.threshold(decode.median_filter_time(p, w), 0.5)
        b = decode.threshold(p, 0.5)
        pad = w // 2
        bp = np.concatenate([b[:pad][::-1], b, b[-pad:][::-1]], 0)
        votes = sum(bp[k:k + len(b)] for k in range(w))
        assert np.array_equal(a, (votes > w // 2).astype(np.int64))


def test_confusion_counts_agree_with_sklearn():
    from sklearn.metrics import f1_score, precision_score, recall_score
    rng = np.random.Generator(np.random.PCG64(6))
    y = rng.integers(0, 2, (500, 5)); yh = rng.integers(0, 2, (500, 5)); yh[:, 4] = 0
    tp, fp, fn, tn = decode.confusion_counts(y, yh)
    prec, rec, f1 = decode.prf_from_counts(tp, fp, fn)
    assert np.allclose(prec, precision_score(y, yh, average=None, zero_division=0.0))
    assert np.allclose(rec, recall_score(y, yh, average=None, zero_division=0.0))
    assert np.allclose(f1, f1_score(y, yh, average=None, zero_division=0.0))
    p, r, f = decode.prf_from_counts(tp.sum(), fp.sum(), fn.sum())
    assert np.isclose(f, f1_score(y.flatten(), yh.flatten(), zero_division=0.0))
