"""run_fovs period with the labelling on the table stream / on the main stream."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = bench.load_codebook()
fovs = [synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234 + i, dev) for i in range(2)]
sf = bench.initial_scale_factors(fovs[0]); bg = np.zeros(16)
for name in ('three side streams', 'three side streams', 'labelling on the table stream',
             'rows on the table stream too', 'three side streams'):
    task = Decode(cb, {'crop_width': 100, 'distance_threshold': 0.5167}, device=dev)
    task.reserve_memory(fovs[0].shape, expectedBarcodes=280000)
    if name != 'three side streams':
        task._label_stream = task._table_stream
    if name == 'rows on the table stream too':
        task._side_stream = task._table_stream
    def run(n):
        for bc, vol in task.run_fovs((fovs[i % 2] for i in range(n)), sf, bg):
            pass
    run(20); torch.cuda.synchronize()
    res = []
    for _ in range(3):
        t0 = time.perf_counter(); run(100); torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 100 * 1e3)
    print('%-32s %s ms/FOV' % (name, ' '.join('%.3f' % r for r in res)), flush=True)
