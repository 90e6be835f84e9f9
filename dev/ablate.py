"""Ablation timings of the decode kernel: codebook size and blur radius."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.codebook import Codebook
from merlin_amd.decoding import PixelBasedDecoder
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = bench.load_codebook()
img = synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234, dev)
rng = np.random.default_rng(0)
sf = bench.initial_scale_factors(img) * rng.uniform(0.9, 1.1, 16); bg = np.zeros(16)
def t(dec, sigma, tr=False, n=5):
    f = lambda: dec.decode_pixels_device(img, sf, bg, 0.5167, 1, sigma, tr, torch.float64)
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for n in (1, 35, 70, 140):
    dec = PixelBasedDecoder(Codebook(cb.get_barcodes()[:n]), device=dev)
    for sigma in (0, 1):
        print('N=%3d sigma=%d  %.3f ms' % (n, sigma, t(dec, sigma)))
