"""run_fov on one stream, nothing overlapped (profiling target: the per-kernel
times of one FOV with the GPU to themselves)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = bench.load_codebook()
task = Decode(cb, {'crop_width': 100}, device=dev, overlap=False)
img = synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234, dev)
sf = bench.initial_scale_factors(img); bg = np.zeros(16)
for i in range(12):
    bc, vol = task.run_fov(img, sf, bg, fov=i)
    torch.cuda.synchronize()
print(len(bc))
