#!/usr/bin/env python
"""Compare a regenerated fixture directory with tests/golden (the pin check):
    GRIDMM_GOLDEN_OUT=/tmp/gg python -m oracle.gen_golden && python tools/compare_golden.py /tmp/gg
Exit code 1 if any array differs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compare(new_dir, old_dir=os.path.join(ROOT, "tests", "golden")):
    bad = 0
    for f in sorted(os.listdir(new_dir)):
        if not f.endswith(".npz"):
            continue
        a, b = np.load(os.path.join(new_dir, f), allow_pickle=True), np.load(os.path.join(old_dir, f), allow_pickle=True)
        worst, keys_equal = 0.0, set(a.files) == set(b.files)
        same = keys_equal
        for k in (set(a.files) & set(b.files)) - {"versions"}:
            x, y = a[k], b[k]
            if x.shape != y.shape or x.dtype != y.dtype:
                same = False
                continue
            if x.dtype.kind == "f":
                d = np.abs(np.nan_to_num(x.astype(np.float64), posinf=1e30, neginf=-1e30) -
                           np.nan_to_num(y.astype(np.float64), posinf=1e30, neginf=-1e30))
                worst = max(worst, float(d.max()) if d.size else 0.0)
                same &= bool(np.array_equal(x, y, equal_nan=True))
            else:
                same &= bool(np.array_equal(x, y))
        print("%-34s %s  max float diff %.3g" % (f, "identical" if same else "DIFFERS", worst))
        bad += not same
    return bad


if __name__ == "__main__":
    sys.exit(1 if compare(sys.argv[1]) else 0)
