"""Details of one tests/sweeps/fuzz_pp.py case: python3 tests/sweeps/fuzz_pp_case.py <seed>"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'sweeps'))
import preprocess_oracle as po
from merlin_amd import preprocess as pp
import fuzz_pp
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
H = int(rng.integers(1, 300))
W = int(rng.choice([rng.integers(1, 40), rng.integers(40, 400), 112, 113, 224, 225, 128, 256]))
n = int(rng.integers(1, 5))
hp_sigma = float(rng.choice([1.0, 2.0, 3.0])); dsize = int(rng.choice([1, 3, 5, 7, 9, 11]))
dsigma = float(rng.choice([0.6, 1.0, 2.0])); iters = int(rng.choice([0, 1, 2, 3, 4, 5, 8, 20]))
frames = np.stack([fuzz_pp.camera((H, W), int(rng.integers(0, 12)), rng, float(rng.choice([0, 300, 2000]))) for _ in range(n)])
win = po.filter_size_for_sigma(hp_sigma)
for k in range(n):
    f = po.high_pass_filter(frames[k], win, hp_sigma).astype(np.float64)
    for it in (2, 3, 5, 10, iters):
        got = pp.deconvolve_lucyrichardson(f, dsize, dsigma, it)
        want = po.deconvolve_lucyrichardson(f, dsize, dsigma, it)
        err = np.abs(got - want) / np.maximum(np.abs(want), 1.0)
        g16, w16 = po.to_uint16(got), po.to_uint16(want)
        bad = np.nonzero(g16 != w16)
        print('frame', k, 'iters', it, 'max rel err %.3g' % err.max(), 'u16 diffs', len(bad[0]),
              [(int(y), int(x), float(got[y, x]), float(want[y, x])) for y, x in zip(*bad)][:4])
    print('nonzero fraction', np.count_nonzero(f) / f.size, 'max', f.max())
