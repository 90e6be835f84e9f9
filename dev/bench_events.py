"""Does recording HIP events around the decode launch change the pipelined
period?  Same loop as bench.py's timed region, with and without events."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = bench.load_codebook()
params = {'crop_width': 100, 'distance_threshold': 0.5167, 'lowpass_sigma': 1, 'minimum_area': 0}
task = Decode(cb, params, device=dev)
fovs = [synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234 + i, dev) for i in range(2)]
sf = bench.initial_scale_factors(fovs[0]); bg = np.zeros(16)
def run(n):
    for bc, vol in task.run_fovs((fovs[i % 2] for i in range(n)), sf, bg):
        pass
run(5); torch.cuda.synchronize()
for label, ev in (('no events', None), ('events', []), ('no events', None), ('events', [])):
    task.kernel_events = ev
    ts = []
    for _ in range(3):
        if ev is not None: task.kernel_events = []
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(50); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 50 * 1e3)
    print(label, ' '.join('%.3f' % t for t in ts))
