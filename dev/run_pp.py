"""Fixed small preprocess workload for rocprofv3 passes:
python3 dev/run_pp.py [images] [batch]  (2048 x 2048 frames, 20 iterations)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'dev'))
from merlin_amd import preprocess as pp  # noqa: E402
from bench_preprocess import camera_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
frames = camera_frames(n, 2048, 2048)
task = pp.DeconvolutionPreprocess(batch=batch)
out, hist = task.preprocess_device(frames, hist_group=1)
torch.cuda.synchronize()
print('done', int(hist.sum()))
