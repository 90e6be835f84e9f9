"""Time the decode kernel alone for the library named by MERFISH_HIP_LIB:
the FOV-path configuration (traces at decoded pixels only) and the plain one,
min / median of 6 blocks of 10 launches after 20 warm-up launches."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = bench.load_codebook()
task = Decode(cb, {'crop_width': 100}, device=dev)
img = synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234, dev)
sf = bench.initial_scale_factors(img); bg = np.zeros(16)
rng = np.random.default_rng(0)
cfgs = [('equal_sf', sf, 'decoded'), ('random_sf', sf * rng.uniform(0.9, 1.1, 16), 'decoded'),
        ('equal_sf', sf, False)]
fs = [(n, tr, (lambda s=s, tr=tr: task.decoder.decode_pixels_device(img, s, bg, 0.5167, 1, 1, tr, torch.float64)))
      for n, s, tr in cfgs]
for _ in range(7):
    for _, _, f in fs: f()
torch.cuda.synchronize()
res = {i: [] for i in range(len(fs))}
for blk in range(6):
    for i, (_, _, f) in enumerate(fs):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): v = f()
        e1.record(); torch.cuda.synchronize()
        res[i].append(e0.elapsed_time(e1) / 10)
name = os.path.basename(os.environ.get('MERFISH_HIP_LIB', 'default'))
for i, (n, tr, _) in enumerate(fs):
    r = sorted(res[i])
    print('%-18s %-9s traces=%-7s min %.3f med %.3f ms' % (name, n, tr, r[0], r[len(r) // 2]))
