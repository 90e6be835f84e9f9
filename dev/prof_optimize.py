"""cProfile of one OptimizeIteration (10 fragments) on the GPU box."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from merlin_amd.optimize import OptimizeIteration, initial_scale_factors, pixel_histogram  # noqa
from merlin_amd.synthetic import synthetic_fov  # noqa

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
cb = bench.load_codebook()
fov = synthetic_fov(cb.get_barcodes(), bench.Z, bench.H, bench.W, bench.SPOTS_PER_PLANE,
                    bench.SEED, dev)
sf0 = initial_scale_factors(pixel_histogram(fov.reshape(16, -1)).cpu().numpy())
bgopt = len(sys.argv) > 1 and sys.argv[1] == 'bg'


def go(n):
    it = OptimizeIteration(cb, {'fov_per_iteration': n, 'optimize_background': bgopt, 'fragment_workers': int(os.environ.get('MF_WORKERS', '4'))},
                           fovs=[0], zCount=bench.Z, initialScaleFactors=sf0,
                           imageProvider=lambda f, z: fov[:, z], device=dev, seed=1).run()
    it.get_scale_factors()
    torch.cuda.synchronize()


go(2)
pr = cProfile.Profile()
pr.enable()
go(10)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
