"""The driver's 20+5 window, yield by yield: one generator, 5 warm-up FOVs,
sync, 20 FOVs with their time stamps, sync."""
import os, sys, time, itertools
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
task.reserve_memory(fovs[0].shape, expectedBarcodes=280000)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
PRIME = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # decode launches before anything else
for _ in range(PRIME):
    v = task.decode_fov_device(fovs[0], sf, bg)
torch.cuda.synchronize()
for rep in range(2):
    gen = task.run_fovs((fovs[i % 2] for i in itertools.count()), sf, bg)
    for _ in range(W):
        next(gen)
    torch.cuda.synchronize()
    task.kernel_events = []
    t0 = time.perf_counter(); st = []
    for _ in range(20):
        next(gen); st.append(time.perf_counter())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ev, task.kernel_events = task.kernel_events, None
    gen.close()
    print('rep %d: %.3f ms/FOV; yields at' % (rep, (t1 - t0) / 20 * 1e3),
          ' '.join('%.1f' % ((t - t0) * 1e3) for t in st), '| end %.1f' % ((t1 - t0) * 1e3))
    print('   kernel ms', ' '.join('%.2f' % a.elapsed_time(b) for a, b in ev))
    print('   kernel starts', ' '.join('%.1f' % ev[0][0].elapsed_time(a) for a, b in ev))
