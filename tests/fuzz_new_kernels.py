"""One-off shape fuzz of the kernels added late in round 1 (upconv, input tail, post-processing) against their torch / scipy
statements: odd sizes, H != W, tiny planes, channel counts that leave partial groups.  Not part of the pytest suites."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import kernel_cases as kc

dev = torch.device("cuda:0")
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0


def check(name, fn):
    global bad
    try:
        err, tol = fn(dev)
        torch.cuda.synchronize()
        ok = err <= tol
    except Exception as e:  # noqa: BLE001
        err, tol, ok = repr(e)[:200], 0, False
    if not ok:
        bad += 1
    print("%-70s %s  (%s)" % (name, "ok" if ok else "FAIL", err), flush=True)


for i in range(14):
    N = int(rs.randint(1, 4))
    h, w = int(rs.randint(1, 12)), int(rs.randint(1, 12))
    f = float(rs.choice([1.5, 2.0, 3.0, 4.0, 5.3]))
    H, W = max(h, int(h * f + rs.randint(0, 3))), max(w, int(w * f + rs.randint(0, 3)))
    C = int(rs.choice([4, 8, 12, 64, 100]))
    dil = int(rs.choice([1, 1, 1, 2]))
    rows = int(rs.choice([0, 1, 2]))
    ar = None if rows == 0 else (N * H * W if rows == 1 or N % 2 else N * H * W // 2)
    check("upconv N=%d %dx%d->%dx%d C=%d dil=%d addend_rows=%s" % (N, h, w, H, W, C, dil, ar), kc.case_upconv(N, h, w, H, W, C, dil=dil, addend_rows=ar, seed=100 + i))
for i in range(6):
    B, H, W = int(rs.randint(1, 4)), int(rs.randint(13, 150)), int(rs.randint(13, 150))
    check("normalize_tf B=%d %dx%d" % (B, H, W), kc.case_normalize_tf(B, H, W, seed=200 + i))
for i in range(5):
    B, H, W = int(rs.randint(1, 3)), int(rs.randint(24, 120)), int(rs.randint(24, 120))
    check("elastic B=%d %dx%d" % (B, H, W), kc.case_elastic(B, H, W, seed=300 + i))
for i in range(5):
    B, H, W = int(rs.randint(1, 3)), int(rs.randint(20, 140)), int(rs.randint(20, 140))
    check("postprocess B=%d %dx%d" % (B, H, W), kc.case_postprocess(B, H, W, seed=400 + i))
print("FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
