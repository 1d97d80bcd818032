"""Diagnostic: is the 2-rank deterministic run reproducible across process launches, with and without the bucketed all-reduce?"""
import os, sys, tempfile, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import test_gpu_data_parallel as T
import pathlib
def run(tag, buckets, port):
    d = pathlib.Path(tempfile.mkdtemp())
    T._spawn(2, port, d, 3, 48, buckets=buckets, det=True)
    suf = "_buckets" if buckets else ""
    return [dict(np.load(d / ("w2_r%d_B48%s.npz" % (r, suf)))) for r in (0, 1)]
if __name__ == "__main__":
  runs = [("single_%d" % i, False, 29411 + i) for i in range(8)] + [("bucket_%d" % i, True, 29431 + i) for i in range(6)]
  res = {t: run(t, b, p) for t, b, p in runs}
  def cmp(a, b):
    out = []
    for k in res[a][0]:
        x, y = res[a][0][k].astype(np.float64), res[b][0][k].astype(np.float64)
        out.append("%s:%.3g(%d)" % (k, np.abs(x - y).max(), int((x != y).sum())))
    print(a, "vs", b, " ".join(out))
  for t, _, _ in runs[1:]:
    cmp("single_0", t)
