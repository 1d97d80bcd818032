"""Microbenchmark of pfo_segment_sum on synthetic segment shapes (GPU box): which part of the C2 duration is what."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pfotgnrec_amd import _lib

dev = torch.device("cuda:0")
W0, W1 = 704, 172


def run(name, lens, live_frac, shuffle, by_pos=1, reps=20):
    for chunked in (0, 1):
        run1(name + (" [members]" if chunked else " [segments]"), lens, live_frac, shuffle, by_pos, reps, chunked)


def run1(name, lens, live_frac, shuffle, by_pos, reps, chunked):
    lens = np.asarray(lens, np.int64)
    S = len(lens); M = int(lens.sum())
    seg_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(dev)
    mem = np.arange(M, dtype=np.int32)
    if shuffle: mem = np.random.RandomState(0).permutation(M).astype(np.int32)
    members = torch.from_numpy(mem).to(dev)
    live = torch.from_numpy((np.random.RandomState(1).rand(M) < live_frac).astype(np.uint8)).to(dev)
    src0 = torch.randn(M, W0, device=dev); src1 = torch.randn(M, W1, device=dev)
    out = torch.empty(S, W0 + W1, device=dev)
    n_rows = torch.tensor([S], dtype=torch.int32, device=dev)
    so = np.zeros((M // 16 + 2) * 16, np.int32); so[:M] = np.repeat(np.arange(S), lens)
    seg_of = torch.from_numpy(so).to(dev)
    big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    st = _lib.stream_ptr()
    ts = []
    for r in range(reps):
        big.zero_()                                               # flush L2 / MALL
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("pfo_segment_sum", _lib.ptr(src0), W0, _lib.ptr(src1), W1, _lib.ptr(seg_ptr), _lib.ptr(members),
                  _lib.ptr(seg_of) if chunked else None, M, _lib.ptr(n_rows), S, by_pos, _lib.ptr(live), _lib.ptr(out), st)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    nlive = int(live.sum())
    byts = (nlive * W0 + M * W1 + S * (W0 + W1)) * 4
    t = float(np.median(ts))
    # check
    if S * (W0 + W1) < 4e7:
        l = live.bool().cpu().numpy(); s0 = src0.cpu().numpy() * l[:, None]; s1 = src1.cpu().numpy()[mem]
        rows = np.concatenate([s0, s1], 1)
        ref = np.add.reduceat(rows, np.cumsum(lens) - lens)[: S]
        ref[lens == 0] = 0
        err = np.abs(out.cpu().numpy() - ref).max()
    else:
        err = -1
    print("%-44s S %6d M %6d live %6d  %7.1f us  %6.0f GB/s  err %.2e" % (name, S, M, nlive, t, byts / t / 1e3, err), flush=True)


rng = np.random.RandomState(5)
run("calibration: 4 segments", np.ones(4), 1.0, False)
run("all 1-member, live", np.ones(54000), 1.0, False)
run("all 5-member, live", np.full(10800, 5), 1.0, False)
run("all 5-member, 25% live", np.full(10800, 5), 0.25, False)
run("all 5-member, 25% live, shuffled", np.full(10800, 5), 0.25, True)
mix = np.concatenate([rng.randint(1, 5, size=9400), rng.randint(15, 42, size=500)])
run("C2-like mix, 25% live", mix, 0.25, False)
run("C2-like mix, 25% live, shuffled", mix, 0.25, True)
run("C2-like mix, all live, shuffled", mix, 1.0, True)
run("500 x 28 only", np.full(500, 28), 0.25, True)
run("9400 x 1-4 only", rng.randint(1, 5, size=9400), 0.25, True)
