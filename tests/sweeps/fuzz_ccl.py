"""Randomised sweep of mf_label_components against scipy's labelling of every
barcode (tests/test_gpu_ccl.py's checker): random volumes, code counts, fill
fractions (sparse to percolating), blob sizes, both connectivities.  Labels and
both counts must be equal.
    python3 -u tests/sweeps/fuzz_ccl.py [cases] [first_seed]"""
import os
import sys
import time

import numpy as np
import torch
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_ccl import _label_gpu, _label_ref       # noqa: E402


def one_case(seed, dev):
    rng = np.random.default_rng(seed)
    Z = int(rng.integers(1, 5))
    H = int(rng.choice([rng.integers(1, 40), rng.integers(40, 220), 32, 33, 64]))
    W = int(rng.choice([rng.integers(1, 70), rng.integers(70, 330), 64, 65, 128]))
    codes = int(rng.integers(1, 6))
    smooth = float(rng.choice([0.0, 0.0, 1.0, 3.0]))
    fill = float(rng.uniform(0.1, 0.95))
    f = rng.standard_normal((Z, H, W))
    if smooth > 0:
        f = ndimage.gaussian_filter(f, (0, smooth, smooth))
    ranks = f.argsort(axis=None).argsort().reshape(f.shape) / f.size
    dec = np.minimum((ranks * codes).astype(np.int16), codes - 1)
    dec[rng.random(dec.shape) > fill] = -1
    conn3d = bool(rng.integers(2))
    desc = 'seed %d Z %d H %d W %d codes %d smooth %.0f fill %.2f 3d %d' % (
        seed, Z, H, W, codes, smooth, fill, conn3d)
    got, counts = _label_gpu(dec, conn3d, dev)
    exp, n_comp = _label_ref(dec, conn3d)
    if not np.array_equal(got, exp):
        return False, 'labels', desc
    if counts[0] != (dec >= 0).sum() or counts[1] != n_comp:
        return False, 'counts', desc
    return True, '', desc


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    dev = torch.device('cuda:0')
    t0 = time.time()
    for k in range(cases):
        ok, why, desc = one_case(first + k, dev)
        print('%s  %s  [%.0f s]' % ('ok ' if ok else 'FAIL', desc, time.time() - t0), flush=True)
        if not ok:
            print('MISMATCH:', why, flush=True)
            sys.exit(1)
    print('all %d cases equal' % cases)


if __name__ == '__main__':
    main()
