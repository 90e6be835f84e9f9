"""run_fovs period for several pipeline depths."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = bench.load_codebook()
task = Decode(cb, {'crop_width': 100, 'distance_threshold': 0.5167}, device=dev)
fovs = [synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234 + i, dev) for i in range(2)]
sf = bench.initial_scale_factors(fovs[0]); bg = np.zeros(16)
def run(n):
    for bc, vol in task.run_fovs((fovs[i % 2] for i in range(n)), sf, bg):
        pass
for ahead, lag in ((1, 1), (1, 2), (1, 3), (2, 2), (2, 3), (2, 4), (3, 4)):
    task.decodeAhead, task.frameLag = ahead, lag
    run(12); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); run(120); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 120 * 1e3)
    print('decodeAhead %d frameLag %d: %.3f ms/FOV  mem %.1f GB' % (ahead, lag, best, torch.cuda.memory_reserved() / 1e9))
