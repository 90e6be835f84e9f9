"""Pipelined FOV period (run_fovs, 3 x 120 FOVs after 20) and the decode kernel alone, for
the library MERFISH_HIP_LIB names."""
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
task.reserve_memory(fovs[0].shape, expectedBarcodes=280000)
def run(n):
    for bc, vol in task.run_fovs((fovs[i % 2] for i in range(n)), sf, bg):
        pass
run(20); torch.cuda.synchronize()
res = []
for _ in range(3):
    t0 = time.perf_counter(); run(120); torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 120 * 1e3)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for _ in range(5): v = task.decode_fov_device(fovs[0], sf, bg)
e0.record()
for _ in range(20): v = task.decode_fov_device(fovs[0], sf, bg)
e1.record(); torch.cuda.synchronize()
print(os.path.basename(os.environ.get('MERFISH_HIP_LIB', 'default')), 'period', ' '.join('%.3f' % r for r in res),
      'ms/FOV; decode alone %.3f ms' % (e0.elapsed_time(e1) / 20), flush=True)
