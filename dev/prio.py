"""run_fovs period with the decode kernels on a high-priority stream."""
import os, sys, time
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
def bench_it(tag):
    run(8); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); run(40); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
    print('%s: %.3f ms/FOV' % (tag, best))
print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else '')
bench_it('default stream')
hi = torch.cuda.Stream(dev, priority=-1)
with torch.cuda.stream(hi):
    bench_it('decode on priority -1 stream')
lo = torch.cuda.Stream(dev, priority=0)
with torch.cuda.stream(lo):
    bench_it('decode on a priority 0 side stream')
task._tbl = torch.cuda.Stream(dev, priority=1) if False else task._tbl
