"""bench.py's two clocks side by side, several repeats (which one is slower, and
by how much, at the driver's 20 + 5 and at the default 200 + 20)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = bench.load_codebook()
task = Decode(cb, {'crop_width': 100, 'distance_threshold': 0.5167, 'lowpass_sigma': 1,
                   'minimum_area': 0}, device=dev)
fovs = [synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234 + i, dev) for i in range(2)]
sf = bench.initial_scale_factors(fovs[0]); bg = np.zeros(16)
def barrier():
    torch.cuda.synchronize(dev)
if '--reserve' in sys.argv:
    task.reserve_memory(fovs[0].shape)
for steps, warm in ((20, 5), (20, 5), (200, 20), (20, 5), (200, 20)):
    dt, ev, bc, vol = bench.timed_steady(task, fovs, sf, bg, steps, warm, barrier)
    k = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    dc, ev2, bc, vol = bench.timed_cold(task, fovs, sf, bg, steps, barrier)
    k2 = sum(a.elapsed_time(b) for a, b in ev2) / len(ev2)
    print('steps %3d warm %2d: steady %.3f ms/FOV (decode kernel %.3f)   cold %.3f (%.3f)'
          % (steps, warm, dt / steps * 1e3, k, dc / steps * 1e3, k2), flush=True)
