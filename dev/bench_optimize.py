"""OptimizeIteration on one MI355X (the single-GPU share):
3 chained iterations of F = 50 random (fov, z) fragments of 2048 x 2048 x 16-bit
planes, optimize_background off and on.  Prints one JSON line.

    python dev/bench_optimize.py [--fragments 50] [--iterations 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fragments', type=int, default=50)
    ap.add_argument('--iterations', type=int, default=3)
    ap.add_argument('--workers', type=int, default=None)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    from merlin_amd.optimize import OptimizeIteration, initial_scale_factors, pixel_histogram
    from merlin_amd.synthetic import synthetic_fov

    codebook = bench.load_codebook()
    Z, H, W = bench.Z, bench.H, bench.W
    fovs = [synthetic_fov(codebook.get_barcodes(), Z, H, W, bench.SPOTS_PER_PLANE,
                          bench.SEED + i, dev) for i in range(2)]
    hist = sum(pixel_histogram(f.reshape(f.shape[0], -1)) for f in fovs)
    sf0 = initial_scale_factors(hist.cpu().numpy())

    def provider(fov, z):
        return fovs[fov][:, z]

    out = {'fragments': args.fragments, 'iterations': args.iterations,
           'plane': [bench.B, H, W], 'runs': []}
    for background in (False, True):
        for rep in range(2):                   # rep 0 warms up
            prev = None
            times = []
            for it in range(args.iterations):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                cur = OptimizeIteration(
                    codebook, dict({'fov_per_iteration': args.fragments,
                                    'optimize_background': background},
                                   **({} if args.workers is None
                                      else {'fragment_workers': args.workers})),
                    fovs=[0, 1], zCount=Z, previousIteration=prev,
                    initialScaleFactors=sf0, imageProvider=provider, device=dev,
                    seed=99 + it).run()
                sf = cur.get_scale_factors()
                bg = cur.get_backgrounds()
                torch.cuda.synchronize(dev)
                times.append(time.perf_counter() - t0)
                prev = cur
        out['runs'].append({
            'optimize_background': background,
            's_per_iteration': times,
            'ms_per_fragment': [t / args.fragments * 1e3 for t in times],
            'Mpx_per_s': [args.fragments * H * W / t / 1e6 for t in times],
            'scale_factors_last': [float(v) for v in sf],
            'backgrounds_last': [float(v) for v in bg],
            'barcodes_per_fragment': float(np.mean(
                [len(v) for v in cur.barcodes.values()]))})
    print(json.dumps(out))


if __name__ == '__main__':
    main()
