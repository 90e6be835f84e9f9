"""Randomised parity sweep of the preprocess stage (HIP path) against
oracle/preprocess_oracle.py: random frame shapes (including widths that are
odd, narrower than a strip, not a multiple of the strip cover), batches,
high-pass widths, deconvolution windows (odd 1..11) and iteration counts.
Bars as in tests/test_gpu_preprocess.py: high pass bit-exact; deconvolution
bit-exact up to two iterations, 1e-9 relative after; the uint16 hand-over
(a truncation) off by one count only where the oracle's float64 value is within
that 1e-9 of an integer.  A development check:
    python3 -u tests/sweeps/fuzz_pp.py [cases] [first_seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import preprocess_oracle as po                     # noqa: E402
from merlin_amd import preprocess as pp             # noqa: E402


def camera(shape, n_spots, rng, offset):
    h, w = shape
    img = offset + rng.poisson(100, size=shape).astype(np.float64)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_spots):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        img += rng.uniform(500, 6000) * np.exp(
            -((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 1.4 ** 2))
    return np.clip(np.rint(img), 0, 65535).astype(np.uint16)


def one_case(seed, dev):
    rng = np.random.default_rng(seed)
    H = int(rng.integers(1, 300))
    W = int(rng.choice([rng.integers(1, 40), rng.integers(40, 400), 112, 113, 224, 225, 128, 256]))
    n = int(rng.integers(1, 5))
    hp_sigma = float(rng.choice([1.0, 2.0, 3.0]))
    dsize = int(rng.choice([1, 3, 5, 7, 9, 11]))
    dsigma = float(rng.choice([0.6, 1.0, 2.0]))
    iters = int(rng.choice([0, 1, 2, 3, 4, 5, 8, 20]))
    frames = np.stack([camera((H, W), int(rng.integers(0, 12)), rng,
                              float(rng.choice([0, 300, 2000]))) for _ in range(n)])
    desc = 'seed %d n %d H %d W %d hp %.0f decon %d/%.1f x%d' % (
        seed, n, H, W, hp_sigma, dsize, dsigma, iters)
    # 1. high pass, bit-exact
    win = po.filter_size_for_sigma(hp_sigma)
    for f in frames:
        if not np.array_equal(pp.high_pass_filter(f, win, hp_sigma),
                              po.high_pass_filter(f, win, hp_sigma)):
            return False, 'high pass', desc
    # 2. deconvolution of the filtered frame in float64
    f = po.high_pass_filter(frames[0], win, hp_sigma).astype(np.float64)
    got = pp.deconvolve_lucyrichardson(f, dsize, dsigma, iters)
    want = po.deconvolve_lucyrichardson(f, dsize, dsigma, iters)
    if iters <= 2:
        if not np.array_equal(got, want):
            return False, 'deconvolution not bit-exact at %d iterations' % iters, desc
    else:
        err = (np.abs(got - want) / np.maximum(np.abs(want), 1.0)).max()
        if not err <= 1e-9:
            return False, 'deconvolution error %.3g' % err, desc
    # 3. the whole stage on the stack, uint16 hand-over + histogram
    task = pp.DeconvolutionPreprocess(dict(
        highpass_sigma=hp_sigma, decon_sigma=dsigma, decon_filter_size=dsize,
        decon_iterations=iters), batch=int(rng.integers(1, 4)))
    stack = torch.from_numpy(frames.reshape(n, 1, H, W)).to(dev)
    out, hist = task.preprocess_device(stack)
    out, hist = out.cpu().numpy(), hist.cpu().numpy()
    for k in range(n):
        # the hand-over truncates: a pixel may differ by one count only where
        # the oracle's float64 value sits within the deconvolution's 1e-9 of
        # an integer (isolated pixels that Lucy-Richardson leaves at their
        # integer value do, to 1e-14)
        hp = po.high_pass_filter(frames[k], win, hp_sigma).astype(np.float64)
        want64 = po.deconvolve_lucyrichardson(hp, dsize, dsigma, iters)
        want16 = po.to_uint16(want64)
        diff = out[k, 0].astype(np.int64) - want16.astype(np.int64)
        bad = diff != 0
        if bad.any():
            near = np.abs(want64 - np.rint(want64)) <= 1e-9 * np.maximum(np.abs(want64), 1.0)
            if np.abs(diff[bad]).max() > 1 or not near[bad].all():
                return False, 'hand-over: %d pixels differ, max %d' % (
                    bad.sum(), np.abs(diff).max()), desc
        if not np.array_equal(hist[k], po.pixel_histogram(out[k]).astype(np.int64)):
            return False, 'histogram', desc
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
