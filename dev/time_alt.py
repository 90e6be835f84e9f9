"""Decode kernel time: one resident FOV repeatedly vs two FOVs alternated."""
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
def timeit(pick, reps, warm):
    for r in range(warm): v = task.decode_fov_device(imgs[pick(r)], sf, bg)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps): v = task.decode_fov_device(imgs[pick(r)], sf, bg)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, pick in (('same0', lambda r: 0), ('same1', lambda r: 1), ('alternate', lambda r: r % 2)):
    for reps, warm in ((10, 2), (30, 10)):
        print(name, reps, warm, '%.3f ms' % timeit(pick, reps, warm))
