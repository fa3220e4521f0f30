"""The allocation-free decode entry point (sedmi_decode_events_static: scratch in
a static device array inside the library) against sedmi_decode_events with a
caller-provided workspace, through the C ABI.  Integer work: bit-exact.  Two
calls in a row on the same GPU, the second with a smaller input, so that stale
counters of the first call would show."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(seed, lens, n_cls):
    rng = np.random.Generator(np.random.PCG64(seed))
    z = rng.normal(0, 1, (sum(lens), n_cls))
    k = np.ones(5) / 5
    z = np.stack([np.convolve(z[:, c], k, mode="same") for c in range(n_cls)], 1)
    p = (1 / (1 + np.exp(-3 * z))).astype(np.float32)
    cut = [max(1, n - (i % 3)) for i, n in enumerate(lens)]
    return p, np.asarray(lens, np.int64), np.asarray(cut, np.int64)


def _run(entry, p, lens, cut, median_window, with_ws):
    from sound_event_detection_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    total, n_cls = p.shape
    n_seg = len(lens)
    starts = np.zeros(n_seg + 1, np.int64)
    np.cumsum(lens, out=starts[1:])
    d_p = torch.from_numpy(p).to(dev)
    d_start, d_cut = torch.from_numpy(starts).to(dev), torch.from_numpy(cut).to(dev)
    cap = total * n_cls // 2 + n_seg * n_cls + 1
    ev = torch.full((4, cap), -7, dtype=torch.int32, device=dev)
    n_ev = torch.full((1,), -1, dtype=torch.int64, device=dev)
    args = [_lib.ptr(d_p), total, n_cls, _lib.ptr(d_start), _lib.ptr(d_cut), n_seg, 0.5, 0, median_window,
            _lib.ptr(ev[0]), _lib.ptr(ev[1]), _lib.ptr(ev[2]), _lib.ptr(ev[3]), cap, _lib.ptr(n_ev)]
    with torch.cuda.device(dev):
        if with_ws:
            ws_bytes = lib.sedmi_decode_workspace_bytes(n_seg, n_cls)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            args += [_lib.ptr(ws), ws_bytes]
        args.append(_lib.stream_ptr(dev))
        _lib.check(getattr(lib, entry)(*args), entry)
    torch.cuda.synchronize(dev)
    n = int(n_ev.item())
    return n, ev[:, :n].cpu().numpy()


@pytest.mark.parametrize("median_window", [1, 5])
def test_static_scratch_decode_equals_workspace_decode_twice(median_window):
    big = _case(5, [250, 0, 1, 63, 64, 65, 129, 750, 2500] * 40, 10)        # 3600 (recording, class) pairs
    small = _case(6, [250, 40, 5], 3)                                        # 9 pairs: reuses the scratch's head
    for p, lens, cut in (big, small):
        n_ref, ev_ref = _run("sedmi_decode_events", p, lens, cut, median_window, True)
        n_got, ev_got = _run("sedmi_decode_events_static", p, lens, cut, median_window, False)
        assert n_ref > 0 and n_got == n_ref
        assert np.array_equal(ev_got, ev_ref)


def test_static_scratch_decode_rejects_sizes_beyond_its_maximum():
    from sound_event_detection_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    n_cls, n_seg = 16, 65536 // 16 + 1
    p = torch.zeros((n_seg, n_cls), dtype=torch.float32, device=dev)
    starts = torch.arange(n_seg + 1, dtype=torch.int64, device=dev)
    cut = torch.ones(n_seg, dtype=torch.int64, device=dev)
    ev = torch.empty((4, 8), dtype=torch.int32, device=dev)
    n_ev = torch.zeros(1, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.sedmi_decode_events_static(_lib.ptr(p), n_seg, n_cls, _lib.ptr(starts), _lib.ptr(cut), n_seg, 0.5, 0,
                                            1, _lib.ptr(ev[0]), _lib.ptr(ev[1]), _lib.ptr(ev[2]), _lib.ptr(ev[3]), 8,
                                            _lib.ptr(n_ev), _lib.stream_ptr(dev))
    assert rc == -1
    assert b"static scratch" in lib.sedmi_last_error()
    # exactly at the maximum it runs
    n_seg -= 1
    with torch.cuda.device(dev):
        rc = lib.sedmi_decode_events_static(_lib.ptr(p), n_seg, n_cls, _lib.ptr(starts), _lib.ptr(cut), n_seg, 0.5, 0,
                                            1, _lib.ptr(ev[0]), _lib.ptr(ev[1]), _lib.ptr(ev[2]), _lib.ptr(ev[3]), 8,
                                            _lib.ptr(n_ev), _lib.stream_ptr(dev))
    torch.cuda.synchronize(dev)
    assert rc == 0 and int(n_ev.item()) == 0
