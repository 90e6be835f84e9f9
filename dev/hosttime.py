"""Host-side cost of the pipelined FOV loop: wall time per FOV of run_fovs, the
same loop under cProfile, and how long the host spends blocked on the GPU."""
import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = bench.load_codebook()
task = Decode(cb, {'crop_width': 100}, device=dev)
fovs = [synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234 + i, dev) for i in range(2)]
sf = bench.initial_scale_factors(fovs[0]); bg = np.zeros(16)
def run(n):
    for bc, vol in task.run_fovs((fovs[i % 2] for i in range(n)), sf, bg):
        pass
run(6); torch.cuda.synchronize()
t0 = time.perf_counter(); run(30); torch.cuda.synchronize()
print('run_fovs %.3f ms/FOV' % ((time.perf_counter() - t0) / 30 * 1e3))
# time blocked in event.synchronize
import torch.cuda
blocked = [0.0]
orig = torch.cuda.Event.synchronize
def timed(self):
    t = time.perf_counter(); orig(self); blocked[0] += time.perf_counter() - t
torch.cuda.Event.synchronize = timed
t0 = time.perf_counter(); run(30); torch.cuda.synchronize()
tot = time.perf_counter() - t0
print('total %.3f ms/FOV, host blocked on events %.3f ms/FOV, host busy %.3f ms/FOV' % (tot / 30 * 1e3, blocked[0] / 30 * 1e3, (tot - blocked[0]) / 30 * 1e3))
torch.cuda.Event.synchronize = orig
pr = cProfile.Profile(); pr.enable(); run(30); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
