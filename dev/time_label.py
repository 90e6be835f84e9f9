"""mf_label_components alone on the bench FOV's decoded image (HIP events, 20 calls),
for the library MERFISH_HIP_LIB names."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = bench.load_codebook()
task = Decode(cb, {'crop_width': 100, 'distance_threshold': 0.5167}, device=dev)
img = synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234, dev)
sf = bench.initial_scale_factors(img); bg = np.zeros(16)
vol = task.decode_fov_device(img, sf, bg)
for _ in range(5):
    task.decoder._label_enqueue(vol, False)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lab = task.decoder._label_enqueue(vol, False)
e1.record(); torch.cuda.synchronize()
print(os.path.basename(os.environ.get('MERFISH_HIP_LIB', 'default')), 'label %.1f us' % (e0.elapsed_time(e1) / 20 * 1e3),
      'counts', lab[2].tolist(), flush=True)
