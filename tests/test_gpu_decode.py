"""HIP threshold / decode kernels vs the reference's own outputs and the oracle
(needs an MI355X).  Integer work: bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import decode as odec

pytestmark = pytest.mark.gpu


def _golden(golden_dir):
    inputs = np.load(os.path.join(golden_dir, "decode_inputs.npz"))
    with open(os.path.join(golden_dir, "decode_expected.json")) as f:
        return inputs, json.load(f)


def test_decode_matches_reference_golden(golden_dir):
    from sound_event_detection_amd import evaluation as ev
    inputs, meta = _golden(golden_dir)
    for m in meta:
        id2cls = {i: f"cls{i}" for i in range(m["n_cls"])}
        pred = torch.from_numpy(inputs[m["name"]])
        if m["apply_sigmoid"]:
            # torch's CPU sigmoid and the device expf may differ in the last ulp: frames whose
            # probability is within tolerance of the threshold are exempt, so move them away
            p = pred.clone()
            near = (torch.sigmoid(p) - m["thr"]).abs() < 1e-5
            p[near] += 0.01
            want = odec.get_prediction_from_raw_output(p.numpy(), id2cls, m["audio_duration"], m["block_length"],
                                                       "file.wav", m["thr"], True)
            pred = p
        else:
            want = [(f, float.fromhex(a), float.fromhex(b), c) for f, a, b, c in m["events"]]
        got = ev.get_prediction_from_raw_output(pred.cuda(), id2cls, m["audio_duration"], m["block_length"],
                                                "file.wav", decision_threshold=m["thr"],
                                                apply_sigmoid=m["apply_sigmoid"])
        assert len(got) == len(want), m["name"]
        for g, w in zip(got, want):
            assert g[0] == w[0] and g[3] == w[3], m["name"]
            assert isinstance(g[1], np.float64) and isinstance(g[2], np.float64)
            assert np.float64(g[1]).tobytes() == np.float64(w[1]).tobytes(), m["name"]
            assert np.float64(g[2]).tobytes() == np.float64(w[2]).tobytes(), m["name"]


def test_cpu_tensor_input_is_accepted(golden_dir):
    """training.py:348-356 hands the decoder a CPU tensor."""
    from sound_event_detection_amd import evaluation as ev
    inputs, meta = _golden(golden_dir)
    m = next(x for x in meta if x["name"] == "rand_10cls")
    id2cls = {i: f"cls{i}" for i in range(10)}
    got = ev.get_prediction_from_raw_output(torch.from_numpy(inputs["rand_10cls"]), id2cls, 10.0, 40, "file.wav")
    assert len(got) == len(m["events"])


def test_batched_decode_equals_per_recording_oracle():
    from sound_event_detection_amd import evaluation as ev
    rng = np.random.Generator(np.random.PCG64(11))
    lens = [250, 0, 1, 63, 64, 65, 128, 129, 750, 2500]
    cut = [251, 5, 1, 63, 10, 66, 200, 129, 700, 2501]
    n_cls = 7
    z = rng.normal(0, 1, (sum(lens), n_cls))
    k = np.ones(5) / 5
    z = np.stack([np.convolve(z[:, c], k, mode="same") for c in range(n_cls)], 1)
    p = (1 / (1 + np.exp(-3 * z))).astype(np.float32)
    seg, cls, on, off = ev.decode_recordings(torch.from_numpy(p).cuda(), lens, cut, 0.5)
    want, o = [], 0
    for s, (n, c) in enumerate(zip(lens, cut)):
        for (cc, a, b) in odec.decode_events(odec.threshold(p[o:o + n]), c):
            want.append((s, cc, a, b))
        o += n
    got = list(zip(seg.tolist(), cls.tolist(), on.tolist(), off.tolist()))
    assert got == want


@pytest.mark.parametrize("window", [3, 7, 21, 129])
def test_median_window_matches_scipy(window):
    from sound_event_detection_amd import evaluation as ev
    rng = np.random.Generator(np.random.PCG64(window))
    lens = [250, 40, 5, 1000]
    n_cls = 4
    p = rng.uniform(0, 1, (sum(lens), n_cls)).astype(np.float32)
    p[np.abs(p - 0.5) < 1e-4] = 0.6
    seg, cls, on, off = ev.decode_recordings(torch.from_numpy(p).cuda(), lens, [10 ** 6] * len(lens), 0.5,
                                             median_window=window)
    want, o = [], 0
    for s, n in enumerate(lens):
        if window <= 2 * n:      # scipy needs the window to fit a double reflection; smaller cases below
            b = odec.threshold(odec.median_filter_time(p[o:o + n], window))
        else:
            b = None
        if b is not None:
            want += [(s, cc, a, e) for (cc, a, e) in odec.decode_events(b, n)]
        o += n
    got = [g for g in zip(seg.tolist(), cls.tolist(), on.tolist(), off.tolist())
           if window <= 2 * lens[g[0]]]
    assert got == want


def test_threshold_int64_strict():
    from sound_event_detection_amd import evaluation as ev
    p = torch.tensor([[0.5, 0.50000006, 0.49999997, 1.0, 0.0]], dtype=torch.float32).cuda()
    out = ev.threshold(p, 0.5)
    assert out.dtype == torch.int64 and out.cpu().tolist() == [[0, 1, 0, 1, 0]]
    big = torch.rand(1000, 250, 10, device="cuda")
    assert torch.equal(ev.threshold(big, 0.3), torch.where(big > 0.3, 1, 0))


def test_confusion_counts_and_evaluate_scores():
    from sklearn.metrics import accuracy_score, f1_score, precision_score, recall_score
    from sound_event_detection_amd import evaluation as ev

    class Fixed(torch.nn.Module):
        def forward(self, x):
            return x

    rng = np.random.Generator(np.random.PCG64(3))
    n_cls = 5
    batches = []
    for _ in range(4):
        logits = torch.from_numpy(rng.normal(0, 2, (8, 250, n_cls)).astype(np.float32))
        logits[..., 4] = -9.0                      # a class that never fires (zero_division path)
        labels = torch.from_numpy(rng.integers(0, 2, (8, 250, n_cls)).astype(np.float32))
        batches.append((logits, labels))
    id2cls = {i: f"c{i}" for i in range(n_cls)}
    res, cres = ev.evaluate(Fixed(), "cuda", batches, id2cls, 0.5, apply_sigmoid=True)
    y = np.concatenate([b[1].numpy() for b in batches]).astype(np.int64).reshape(-1, n_cls)
    yh = np.concatenate([(torch.sigmoid(b[0]) > 0.5).numpy() for b in batches]).astype(np.int64).reshape(-1, n_cls)
    f1s = f1_score(y, yh, average=None)
    ps = precision_score(y, yh, average=None, zero_division=0.0)
    rs = recall_score(y, yh, average=None, zero_division=0.0)
    for i in range(n_cls):
        assert np.isclose(cres[f"c{i}"]["f1"], f1s[i]) and np.isclose(cres[f"c{i}"]["precision"], ps[i])
        assert np.isclose(cres[f"c{i}"]["recall"], rs[i])
    assert np.isclose(res["f1"], f1_score(y.flatten(), yh.flatten(), zero_division=0.0))
    assert np.isclose(res["precision"], precision_score(y.flatten(), yh.flatten(), zero_division=0.0))
    assert np.isclose(res["recall"], recall_score(y.flatten(), yh.flatten(), zero_division=0.0))
    assert np.isclose(res["accuracy"], accuracy_score(y.flatten(), yh.flatten()))


def test_decode_fuzz_many_shapes_vs_oracle():
    """Seeded random recording tables: 1 to 92 classes, recordings of 0 to 1500 frames (empty ones, one-frame ones,
    lengths either side of the 64 frames a wave walks per step), cutoffs of 0, inside, at and beyond the recording,
    thresholds other than 0.5, sticky and flickering maps.  Bit-exact against the reference's decode rule
    (evaluation.py:58-78) restated in the oracle, recording by recording."""
    from sound_event_detection_amd import evaluation as ev
    rng = np.random.Generator(np.random.PCG64(2025))
    for case in range(40):
        n_cls = int(rng.choice([1, 2, 3, 6, 7, 10, 16, 17, 33, 64, 92]))
        n_rec = int(rng.integers(1, 9))
        lens = [int(rng.choice([0, 1, 2, 63, 64, 65, 127, 128, 129, 250, int(rng.integers(3, 1500))])) for _ in range(n_rec)]
        cut = [int(rng.choice([0, 1, max(n // 2, 0), n, n + 1, n + 1000])) for n in lens]
        thr = float(rng.choice([0.5, 0.3, 0.71]))
        total = sum(lens)
        if rng.random() < 0.5:                                      # sticky runs
            z = rng.normal(0, 1, (total + 8, n_cls))
            z = np.stack([np.convolve(z[:, c], np.ones(9) / 9, mode="same") for c in range(n_cls)], 1)[:total]
            p = (1 / (1 + np.exp(-4 * z))).astype(np.float32)
        else:                                                       # flicker, with exact ties at the threshold
            p = rng.uniform(0, 1, (total, n_cls)).astype(np.float32)
            p[rng.random((total, n_cls)) < 0.05] = np.float32(thr)  # strict '>': a tie is off
        seg, cls, on, off = ev.decode_recordings(torch.from_numpy(p).cuda(), lens, cut, thr)
        want, o = [], 0
        for s, (n, c) in enumerate(zip(lens, cut)):
            want += [(s, cc, a, b) for (cc, a, b) in odec.decode_events(odec.threshold(p[o:o + n], thr), c)]
            o += n
        got = list(zip(seg.tolist(), cls.tolist(), on.tolist(), off.tolist()))
        assert got == want, (case, n_cls, lens, cut, thr)
