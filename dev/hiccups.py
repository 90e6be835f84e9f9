"""Per-FOV completion gaps of the pipelined loop, with GC events marked."""
import gc, os, sys, time
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
gcs = []
def cb_gc(phase, info):
    gcs.append((time.perf_counter(), phase, info.get('generation')))
gc.callbacks.append(cb_gc)
def run(n, stamps):
    for bc, vol in task.run_fovs((fovs[i % 2] for i in range(n)), sf, bg):
        stamps.append(time.perf_counter())
run(5, []); torch.cuda.synchronize()
for rep in range(4):
    if rep == 2: gc.collect(); gc.freeze(); gc.disable(); print('gc disabled')
    stamps = []; del gcs[:]
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(50, stamps); torch.cuda.synchronize()
    t1 = time.perf_counter()
    gaps = np.diff(np.array([t0] + stamps)) * 1e3
    big = [(i, round(g, 2)) for i, g in enumerate(gaps) if g > 3.0]
    g2 = [(round((t - t0) * 1e3, 1), ph, gen) for t, ph, gen in gcs if gen and gen >= 1]
    print('rep %d: %.3f ms/FOV  median gap %.2f  big gaps %s  gc(gen>=1) %s  reserved %.1f GB' % (
        rep, (t1 - t0) / 50 * 1e3, np.median(gaps), big, g2[:12], torch.cuda.memory_reserved() / 1e9))
