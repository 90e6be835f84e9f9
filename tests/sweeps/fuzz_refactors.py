"""Randomised parity sweep of extract_refactors (the OptimizeIteration
fragment's reduction: region table with trace x magnitude statistics on the
GPU, ordered per-barcode sums on the host) against the oracle, with and
without the background term, over random codebooks, planes, area thresholds.
Bar: 1e-12 relative, NaNs in the same places (tests/test_gpu_parity.py).
    python3 -u tests/sweeps/fuzz_refactors.py [cases] [first_seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import merfish_oracle as oracle                              # noqa: E402
from conftest import ArrayCodebook, load_golden, synth_stack  # noqa: E402
from merlin_amd.codebook import random_weight_codebook        # noqa: E402
from merlin_amd.decoding import PixelBasedDecoder             # noqa: E402


def one_case(seed, dev):
    rng = np.random.default_rng(seed)
    if rng.integers(3) == 0:
        cb = load_golden('g2_synth192')['codebook']
    else:
        from math import comb
        B = int(rng.choice([8, 10, 12, 16, 18, 22, 24, 32]))
        w = int(rng.integers(2, 6))
        N = min(int(rng.integers(3, 200)), comb(B, w) // 2)
        cb = random_weight_codebook(B, N, w, seed=seed).get_barcodes()
    N, B = cb.shape
    H, W = int(rng.integers(20, 160)), int(rng.integers(20, 260))
    spots = int(rng.choice([0.0, 0.002, 0.01, 0.04]) * H * W)
    img = synth_stack(cb, (1, H, W), spots, seed=seed, amp=(500, 5000),
                      bg=float(rng.choice([0.5, 20, 100])),
                      psf=float(rng.choice([0.8, 1.2, 2.5])))[:, 0]
    sf = rng.uniform(100, 900, B)
    bg = rng.uniform(0, 110, B) if rng.integers(2) else np.zeros(B)
    thr = float(rng.uniform(0.3, 0.65))
    area = int(rng.integers(1, 7))
    desc = 'seed %d B %d N %d H %d W %d spots %d thr %.3f area %d' % (
        seed, B, N, H, W, spots, thr, area)
    dm = oracle.normalized_codebook(cb)
    di, pm, npt, d = oracle.decode_pixels(img, dm, sf, bg, thr, 1, 1)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=dev)
    dec.refactorAreaThreshold = area
    gdi, gpm, gnpt, gd = dec.decode_pixels(img, sf, bg, thr, 1, 1)
    if not (np.array_equal(gdi, di) and np.array_equal(gpm, pm)):
        return False, 'decode', desc
    for with_bg in (False, True):
        with np.errstate(all='ignore'):
            want = oracle.extract_refactors(di, pm, npt, dm, area, with_bg)
            got = dec.extract_refactors(gdi, gpm, gnpt, with_bg)
        for a, b, name in zip(got, want, ('refactors', 'background refactors', 'seen')):
            if not np.array_equal(np.isnan(a), np.isnan(b)):
                return False, '%s: NaN pattern (bg %d)' % (name, with_bg), desc
            if not np.allclose(a, b, rtol=1e-12, atol=0, equal_nan=True):
                return False, '%s beyond 1e-12 (bg %d)' % (name, with_bg), desc
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
