"""Randomised parity sweep of the drop-in decode_pixels (HIP path through the
C-ABI) against the oracle: random codebooks (uniform and mixed weight),
uint16 / float32 / float64 images of odd shapes, every blur radius, distance
and magnitude thresholds, awkward scale factors and backgrounds.  Bars as in
tests/test_gpu_parity.py: decoded image, magnitudes and traces bit-exact;
distances bit-exact wherever the winner is unambiguous or the pixel decodes,
2^-18 = 4e-6 relative where two codewords tie to the search key's resolution.
    python3 -u tests/sweeps/fuzz_decode.py [cases] [first_seed]"""
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
    kind = int(rng.integers(5))
    if kind == 0:
        cb = load_golden('g2_synth192')['codebook']
    elif kind == 4:                       # thousands of codewords: untagged keys, streamed chunks
        B = int(rng.choice([24, 32]))
        N = int(rng.integers(3100, 5000))
        cb = random_weight_codebook(B, N, int(rng.integers(4, 6)), seed=seed).get_barcodes()
    elif kind == 3:                       # mixed weights: the general (non-uniform) kernel
        B = int(rng.choice([8, 13, 16, 21, 32]))
        N = int(rng.integers(2, 120))
        cb = np.zeros((N, B), dtype=np.int64)
        for i in range(N):
            cb[i, rng.choice(B, int(rng.integers(1, min(7, B))), replace=False)] = 1
        cb = np.unique(cb, axis=0)
    else:
        from math import comb
        B = int(rng.choice([8, 10, 12, 14, 16, 18, 22, 27, 32]))
        w = int(rng.integers(2, 6))
        N = min(int(rng.integers(1, 400)), max(1, comb(B, w) // 2))
        cb = random_weight_codebook(B, N, w, seed=seed).get_barcodes()
    N, B = cb.shape
    H, W = int(rng.integers(8, 150)), int(rng.integers(8, 260))   # > the largest blur radius
    if kind == 4:
        H, W = min(H, 50), min(W, 90)     # the oracle's brute force is N x pixels
    spots = int(rng.choice([0.0, 0.005, 0.03]) * H * W)
    img = synth_stack(cb, (1, H, W), spots, seed=seed,
                      amp=(500, float(rng.choice([5000, 60000]))),
                      bg=float(rng.choice([0.5, 50, 100])))[:, 0]
    dtype = rng.choice(['u16', 'f32', 'f64'])
    if dtype != 'u16':
        img = img.astype(np.float64) + rng.uniform(0, 1, img.shape)
        img = img.astype(np.float32 if dtype == 'f32' else np.float64)
    sf = np.exp(rng.uniform(np.log(0.5), np.log(2e4), B))
    bg = rng.uniform(-20, 120, B) if rng.integers(2) else np.zeros(B)
    sigma = float(rng.choice([0, 0.5, 1, 1.5, 2, 2.5, 3]))
    dthr = float(rng.uniform(0.2, 0.7))
    mthr = float(rng.choice([0.0, 1.0, 1.0, 3.0]))
    desc = 'seed %d kind %d B %d N %d H %d W %d %s sigma %.1f dthr %.3f mthr %.0f' % (
        seed, kind, B, N, H, W, dtype, sigma, dthr, mthr)
    dm = oracle.normalized_codebook(cb)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=dev)
    di, pm, npt, d = dec.decode_pixels(img, sf, bg, dthr, mthr, sigma)
    ref = img if dtype == 'u16' else img.astype(np.float32)
    odi, opm, onpt, od, gap = oracle.decode_pixels(ref, dm, sf, bg, dthr, mthr, sigma,
                                                   return_gap=True)
    if not np.array_equal(pm, opm):
        return False, 'magnitudes', desc
    if not np.array_equal(npt, onpt):
        return False, 'traces', desc
    if not np.array_equal(di, odi):
        return False, 'decoded image (%d pixels)' % np.count_nonzero(di != odi), desc
    clear = (gap > 1e-4) | (odi >= 0)
    if not np.array_equal(d[clear], od[clear]):
        return False, 'distances', desc
    if not np.allclose(d[~clear], od[~clear], rtol=4e-6, atol=0):
        return False, 'tied distances', desc
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
