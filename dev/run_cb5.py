"""Three decode launches of config 5 (32 bits x 10000 codewords, 10 z) for the profiler."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlin_amd.codebook import random_weight_codebook
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = random_weight_codebook(32, 10000, 4, seed=11)
task = Decode(cb, {'crop_width': 100}, device=dev)
img = synthetic_fov(cb.get_barcodes(), 10, 2048, 2048, 40000, 1234, dev)
sf = np.full(32, 500.0); bg = np.zeros(32)
for _ in range(3):
    v = task.decode_fov_device(img, sf, bg)
torch.cuda.synchronize()
