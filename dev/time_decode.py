"""Time the decode kernel alone for the library named by MERFISH_HIP_LIB."""
import os, sys, time
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
for name, s in (('equal_sf', sf), ('random_sf', sf * rng.uniform(0.9, 1.1, 16))):
    for tr in (True, False):
        f = lambda: task.decoder.decode_pixels_device(img, s, bg, 0.5167, 1, 1, tr, torch.float64)
        f(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): v = f()
        e1.record(); torch.cuda.synchronize()
        print(os.environ.get('MERFISH_HIP_LIB', 'default'), name, 'traces' if tr else 'notraces', '%.3f ms' % (e0.elapsed_time(e1)/5), 'decoded', int((v.decoded >= 0).sum()))
