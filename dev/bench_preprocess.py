"""Preprocess stage (high-pass + 20 Lucy-Richardson iterations) on one MI355X:
time per 2048 x 2048 camera frame by batch size, against the traffic roofline.

    python3 dev/bench_preprocess.py [--images 16] [--batches 1,2,4,7] [--cpu]

Algorithmic bytes per pixel and image (DESIGN.md section 4.5,
merlin_amd.preprocess.algorithmic_bytes_per_pixel): 646 B/px at 20 iterations
(r01: 720, every iteration 34 B plus an initialisation pass).  Prints one
JSON line per batch size.
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
from merlin_amd import preprocess as pp  # noqa: E402

HBM_PEAK = 8000.0     # GB/s, MI355X_MICROARCH.md


def camera_frames(n, H, W, seed=1234):
    g = torch.Generator(device='cuda').manual_seed(seed)
    img = 2000.0 + 100.0 + 10.0 * torch.randn((n, H, W), device='cuda', generator=g)
    # ~1 spot per 100 px^2, sigma 1.4 px, amplitude 500..6000
    n_spots = H * W // 100
    yy, xx = torch.meshgrid(torch.arange(-5, 6, device='cuda'),
                            torch.arange(-5, 6, device='cuda'), indexing='ij')
    psf = torch.exp(-(yy ** 2 + xx ** 2) / (2 * 1.4 ** 2)).reshape(-1)
    for i in range(n):
        cy = torch.randint(5, H - 5, (n_spots,), device='cuda', generator=g)
        cx = torch.randint(5, W - 5, (n_spots,), device='cuda', generator=g)
        amp = 500 + 5500 * torch.rand((n_spots,), device='cuda', generator=g)
        idx = ((cy[:, None] + yy.reshape(-1)[None]) * W + cx[:, None] + xx.reshape(-1)[None])
        img[i].view(-1).index_add_(0, idx.reshape(-1), (amp[:, None] * psf[None]).reshape(-1))
    return img.clamp_(0, 65535).round_().to(torch.uint16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=16)
    ap.add_argument('--size', type=int, default=2048)
    ap.add_argument('--batches', default='1,2,4,7')
    ap.add_argument('--iterations', type=int, default=20)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--cpu', action='store_true', help='time the numpy oracle on one frame')
    a = ap.parse_args()
    H = W = a.size
    frames = camera_frames(a.images, H, W)
    bytes_px = pp.algorithmic_bytes_per_pixel(a.iterations)
    cpu = None
    if a.cpu:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import preprocess_oracle as po
        torch.set_num_threads(1)
        f = frames[0].cpu().numpy()
        t0 = time.perf_counter()
        po.preprocess_image(f, decon_iterations=a.iterations)
        cpu = time.perf_counter() - t0
    for batch in [int(b) for b in a.batches.split(',')]:
        task = pp.DeconvolutionPreprocess(dict(decon_iterations=a.iterations), batch=batch)
        task.preprocess_device(frames[:batch])           # warm-up
        torch.cuda.synchronize()
        best = None
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out, hist = task.preprocess_device(frames, hist_group=1)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        per_image = best / a.images
        gbs = bytes_px * H * W / (per_image * 1e-3) / 1e9
        line = dict(metric='preprocessed Mpixels/sec per GPU', value=H * W / per_image / 1e3,
                    unit='Mpx/s', ms_per_image=per_image, batch=batch, images=a.images,
                    iterations=a.iterations, dtype='f64',
                    roofline=dict(bound='hbm', achieved=gbs, peak=HBM_PEAK, unit='GB/s',
                                  frac=gbs / HBM_PEAK, traffic=None),
                    config=dict(workload='{}x{} camera frames, high-pass sigma 3 + {} Lucy-Richardson '
                                'iterations (9x9, sigma 2)'.format(H, W, a.iterations)),
                    nonzero_frac=float((out.view(torch.int16) != 0).float().mean()))
        if cpu is not None:
            line['cpu_baseline'] = dict(value=H * W / cpu / 1e6, unit='Mpx/s', cores=1, kind='port',
                                        sample='one {}x{} frame, {:.1f} s'.format(H, W, cpu))
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
