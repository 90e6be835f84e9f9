"""Randomised parity sweep: Decode.run_fov (HIP path) against the oracle (and the
typed record sink against the plain frame cast to the database's types) over
random codebooks, image shapes, blur widths, thresholds, crops, minimum areas,
2-D / 3-D labelling and spot densities (sparse to merged blobs).  A development
check, not part of the product: run on the GPU box as
    python3 -u tests/sweeps/fuzz_fov.py [cases] [first_seed]
Stops at the first mismatch and saves its inputs to gpurun_out/fuzz_fail.npz."""
import os
import sys
import time
import warnings

import numpy as np
import pandas
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import merfish_oracle as oracle                      # noqa: E402
from conftest import (ArrayCodebook, AffineAligner, frame_to_array,   # noqa: E402
                      load_golden, synth_stack)
from merlin_amd.codebook import random_weight_codebook           # noqa: E402
from merlin_amd.decode import Decode                               # noqa: E402


def oracle_fov(cb, img, sf, bg, params, fov, tform, decode3d):
    dm = oracle.normalized_codebook(cb)
    Z = img.shape[1]
    outs = [oracle.decode_pixels(img[:, z], dm, sf, bg,
                                 params['distance_threshold'], 1,
                                 params['lowpass_sigma']) for z in range(Z)]
    frames = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if decode3d:
            di = np.array([o[0] for o in outs])
            pm = np.array([o[1] for o in outs])
            npt = np.array([o[2] for o in outs])
            d = np.array([o[3] for o in outs]).astype(np.float32)
            frames.append(oracle.extract_all_barcodes(
                cb.shape[0], di, pm, npt, d, fov, params['crop_width'], None,
                tform, params['minimum_area']))
        else:
            for z, (di, pm, npt, d) in enumerate(outs):
                frames.append(oracle.extract_all_barcodes(
                    cb.shape[0], di, pm, npt, d, fov, params['crop_width'], z,
                    tform, params['minimum_area']))
        return pandas.concat(frames)


def one_case(seed, dev):
    rng = np.random.default_rng(seed)
    kind = rng.integers(4)
    if kind == 0:
        cb = load_golden('g2_synth192')['codebook']
    else:
        B = int(rng.choice([8, 10, 12, 14, 16, 18, 20, 22, 24, 27, 32]))
        N = int(rng.integers(3, 300))
        w = int(rng.integers(2, 6))
        from math import comb
        N = min(N, comb(B, w) // 2)
        cb = random_weight_codebook(B, N, w, seed=seed).get_barcodes()
    N, B = cb.shape
    decode3d = bool(rng.integers(3) == 0)
    Z = int(rng.integers(1, 6)) if not decode3d else int(rng.integers(2, 9))
    H = int(rng.integers(20, 150))
    W = int(rng.integers(20, 260))
    dens = rng.choice([0.0, 0.002, 0.01, 0.04, 0.12])
    spots = int(dens * H * W)
    amp_hi = float(rng.choice([3000, 5000, 60000]))
    img = synth_stack(cb, (Z, H, W), spots, seed=seed, amp=(500, amp_hi),
                      bg=float(rng.choice([0.5, 20, 100])),
                      psf=float(rng.choice([0.8, 1.2, 2.5])))
    sf = rng.uniform(100, 900, B)
    bg = rng.uniform(0, 110, B) if rng.integers(2) else np.zeros(B)
    crop = int(rng.integers(0, min(9, H // 2 - 1, W // 2 - 1)))
    params = {'crop_width': crop,
              'minimum_area': int(rng.integers(0, 5)),
              'distance_threshold': float(rng.uniform(0.3, 0.65)),
              'lowpass_sigma': float(rng.choice([0, 0.5, 1, 1.5, 2, 2.5, 3])),
              'decode_3d': decode3d}
    tform = None
    aligner = None
    if rng.integers(2):
        tform = np.float32([[0.108, 0, rng.uniform(-5, 5)],
                            [0, 0.108, rng.uniform(-5, 5)], [0, 0, 1]])
        aligner = AffineAligner(tform)
    fov = int(rng.integers(0, 500))
    desc = ('seed %d B %d N %d Z %d H %d W %d 3d %d spots %d crop %d area %d '
            'thr %.3f sigma %.1f' % (seed, B, N, Z, H, W, decode3d, spots, crop,
                                     params['minimum_area'],
                                     params['distance_threshold'],
                                     params['lowpass_sigma']))
    task = Decode(ArrayCodebook(cb), params, device=dev)
    bc, vol = task.run_fov(torch.from_numpy(img).to(dev), sf, bg, fov=fov,
                           globalAligner=aligner)
    exp = oracle_fov(cb, img, sf, bg, task.parameters, fov, tform, decode3d)
    r, e = frame_to_array(bc, B), frame_to_array(exp, B)
    ok = r.shape == e.shape
    why = 'shape %s vs %s' % (r.shape, e.shape)
    if ok and len(r):
        for c in (0, 1, 4, 13):
            if not np.array_equal(r[:, c], e[:, c]):
                ok, why = False, 'integer column %d' % c
        if ok and not np.allclose(r, e, rtol=1e-6, atol=1e-6):
            ok, why = False, 'floats beyond 1e-6'
        if ok and not decode3d:
            if not np.allclose(r[:, :14], e[:, :14], rtol=1e-12, atol=1e-12):
                ok, why = False, '2-D scalars beyond 1e-12'
            elif not np.array_equal(r[:, 14:], e[:, 14:]):
                ok, why = False, '2-D per-bit means not bit-exact'
    if ok:
        # the typed record sink (barcode database column types, written on the
        # device) must hold exactly the plain frame cast by the database
        from merlin_amd import barcodedb
        db = barcodedb.BarcodeDB(ArrayCodebook(cb))
        typed, _ = task.run_fov(torch.from_numpy(img).to(dev), sf, bg, fov=fov,
                                globalAligner=aligner, database=db)
        schema = barcodedb.bc_column_types(B)
        cast = bc.astype(schema) if len(bc) else bc
        if len(typed) != len(bc):
            ok, why = False, 'typed sink: %d rows vs %d' % (len(typed), len(bc))
        elif len(bc):
            for c in schema:
                a, b_ = typed[c].to_numpy(), cast[c].to_numpy()
                if not np.array_equal(a, b_, equal_nan=a.dtype.kind == 'f'):
                    ok, why = False, 'typed sink column %s' % c
                    break
    if not ok:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        np.savez(os.path.join(ROOT, 'gpurun_out', 'fuzz_fail.npz'), cb=cb,
                 img=img, sf=sf, bg=bg, got=r, exp=e)
    return ok, why, desc, len(r)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    dev = torch.device('cuda:0')
    t0 = time.time()
    total = 0
    for k in range(cases):
        ok, why, desc, n = one_case(first + k, dev)
        total += n
        print('%s  %s records %d  [%.0f s]' % ('ok ' if ok else 'FAIL', desc, n,
                                                time.time() - t0), flush=True)
        if not ok:
            print('MISMATCH:', why, flush=True)
            sys.exit(1)
    print('all %d cases equal, %d records compared' % (cases, total))


if __name__ == '__main__':
    main()
