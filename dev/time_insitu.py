"""Per-launch decode kernel times inside the pipelined FOV loop (bench.py's
timed region), with and without the side stream (MF_NO_SIDE=1)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = bench.load_codebook()
task = Decode(cb, {'crop_width': 100, 'distance_threshold': 0.5167}, device=dev)
imgs = [synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234 + i, dev) for i in range(2)]
sf = bench.initial_scale_factors(imgs[0]); bg = np.zeros(16)
import time
for rnd in range(2):
    task.kernel_events = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in task.run_fovs((imgs[i % 2] for i in range(40)), sf, bg):
        pass
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ts = [a.elapsed_time(b) for a, b in task.kernel_events]
    print('round', rnd, 'ms/FOV %.3f' % (dt / 40 * 1e3), 'kernel mean %.3f' % (sum(ts) / len(ts)),
          ' '.join('%.2f' % t for t in ts))
