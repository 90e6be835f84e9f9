"""Time the REFERENCE's own decode path as written (per-pixel Python loops,
sklearn BallTree, skimage label / regionprops) on one CPU core.

Run in the build container only (the reference does not travel):

    /opt/conda/bin/python3.9 oracle/time_reference.py [sizes...]

Same import shims as ``oracle/make_goldens.py`` (reference files untouched;
``cv2.GaussianBlur`` is the oracle's restatement because cv2 is absent here, so
the blur's share of the time is a stand-in -- it is a few % of the total).
Workload: SURVEY 8d config 2 scaled down -- 16-bit MHD4 test-suite codebook
(140 codewords), Poisson(100) background + Gaussian spots (1 per 100 px^2,
sigma 1.2, amplitude 1000-5000), scale factors 500, one z plane of size n x n.
Timed per plane: decode_pixels (decoding.py:46-131), the 140 calls of
extract_barcodes_with_index that Decode makes (analysis/decode.py:225-230) and
extract_refactors (OptimizeIteration).  Writes
``profiles/cpu_reference_baseline.json`` with the per-size timings and the
linear extrapolation in pixel count to a 2048 x 2048 plane and a 7-z FOV.
"""
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402  (shims + fixtures, no side effects)


def synthetic_plane(codebook, n, seed):
    rng = np.random.default_rng(seed)
    N, B = codebook.shape
    img = rng.poisson(100, size=(B, n, n)).astype(np.float64)
    r = 4
    oy, ox = np.mgrid[-r:r + 1, -r:r + 1]
    for _ in range(n * n // 100):
        code = rng.integers(N)
        cy, cx = rng.uniform(r, n - r - 1, 2)
        amp = rng.uniform(1000, 5000)
        iy, ix = int(cy), int(cx)
        psf = amp * np.exp(-((iy + oy - cy) ** 2 + (ix + ox - cx) ** 2)
                           / (2 * 1.2 ** 2))
        for b in np.nonzero(codebook[code])[0]:
            img[b, iy - r:iy + r + 1, ix - r:ix + r + 1] += psf
    return np.clip(np.rint(img), 0, 65535).astype(np.uint16)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024]
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    mg.install_shims()
    from merlin.util import decoding          # the reference, unmodified
    import warnings
    warnings.simplefilter('ignore')
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:          # old threadpoolctl here: OMP_NUM_THREADS=1 is set
        limiter = None
    cb = mg.load_test_codebook()
    dec = decoding.PixelBasedDecoder(mg.ArrayCodebook(cb))
    dec.refactorAreaThreshold = 4
    sf, bg = np.full(16, 500.0), np.zeros(16)
    rows = []
    for n in sizes:
        img = synthetic_plane(cb, n, 1234)
        t0 = time.perf_counter()
        di, pm, npt, d = dec.decode_pixels(img, sf, bg, 0.5167, 1, 1)
        t1 = time.perf_counter()
        count = 0
        for i in range(cb.shape[0]):
            count += len(dec.extract_barcodes_with_index(
                i, di, pm, npt, d, 0, 100 if n > 400 else 10, 0, None, 0))
        t2 = time.perf_counter()
        dec.extract_refactors(di, pm, npt, extractBackgrounds=False)
        t3 = time.perf_counter()
        rows.append({'size': n, 'pixels': n * n,
                     'decode_pixels_s': t1 - t0,
                     'extract_barcodes_140_s': t2 - t1,
                     'extract_refactors_s': t3 - t2,
                     'barcodes': count,
                     'decoded_px': int((di >= 0).sum()),
                     'Mpx_per_s_decode_plus_extract':
                         n * n / (t2 - t0) / 1e6})
        print(json.dumps(rows[-1]), flush=True)
    big = rows[-1]
    scale = 2048 * 2048 / big['pixels']
    plane = (big['decode_pixels_s'] + big['extract_barcodes_140_s']) * scale
    out = {
        'what': 'the reference as written (merlin/util/decoding.py imported '
                'unmodified): decode_pixels + 140 x extract_barcodes_with_index'
                ' per plane, plus extract_refactors',
        'host': platform.processor() or platform.machine(),
        'cpu_model': next((l.split(':', 1)[1].strip()
                           for l in open('/proc/cpuinfo')
                           if l.startswith('model name')), 'unknown'),
        'python': sys.version.split()[0], 'numpy': np.__version__,
        'cores': 1, 'blur': 'stand-in (cv2 absent: the oracle restatement)',
        'command': '/opt/conda/bin/python3.9 oracle/time_reference.py '
                   + ' '.join(str(s) for s in sizes),
        'measured': rows,
        'extrapolated': {
            'basis': 'linear in pixel count from the largest size measured',
            'plane_2048_s': plane,
            'fov_7z_2048_s': 7 * plane,
            'Mpx_per_s': 2048 * 2048 / plane / 1e6,
            'extract_refactors_plane_2048_s':
                big['extract_refactors_s'] * scale},
    }
    os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
    with open(os.path.join(ROOT, 'profiles',
                           'cpu_reference_baseline.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out['extrapolated']))
    del limiter


if __name__ == '__main__':
    main()
