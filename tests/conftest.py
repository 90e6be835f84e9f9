import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def golden_names():
    return sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, 'g*.npz')))


def load_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    d = {k: g[k] for k in g.files}
    p = d['params']
    d['p'] = dict(dthr=float(p[0]), mthr=float(p[1]), sigma=float(p[2]),
                  crop=int(p[3]), min_area=int(p[4]), fov=int(p[5]),
                  z_index=None if p[8] else int(p[6]),
                  refactor_area=int(p[7]), decode3d=bool(p[8]))
    return d


class ArrayCodebook:
    """Minimal stand-in for merlin.data.codebook.Codebook: the decoder only
    calls get_barcodes() (merlin/util/decoding.py:270)."""

    def __init__(self, barcodes):
        self._b = np.asarray(barcodes)

    def get_barcodes(self, ignoreBlanks=False):
        return self._b

    def get_barcode_count(self):
        return self._b.shape[0]

    def get_bit_count(self):
        return self._b.shape[1]


class AffineAligner:
    """merlin/analysis/globalalign.py:141-148 with a fixed 3x3 transform."""

    def __init__(self, tform):
        self.t = tform

    def fov_coordinate_array_to_global(self, fov, fovCoordArray):
        toGlobal = np.ones(fovCoordArray.shape)
        toGlobal[:, [0, 1]] = fovCoordArray[:, [1, 2]]
        g = np.matmul(self.t, toGlobal.T).T[:, [2, 0, 1]]
        g[:, 0] = fovCoordArray[:, 0]
        return g


def frame_to_array(df, B):
    cols = ['barcode_id', 'fov', 'mean_intensity', 'max_intensity', 'area',
            'mean_distance', 'min_distance', 'x', 'y', 'z', 'global_x',
            'global_y', 'global_z', 'cell_index'] + \
        ['intensity_{}'.format(i) for i in range(B)]
    if len(df) == 0:
        return np.zeros((0, len(cols)))
    return df[cols].to_numpy().astype(np.float64)


@pytest.fixture(scope='session')
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def synth_stack(codebook, shape, n_spots, seed, amp=(1000, 5000), bg=100,
                psf=1.2):
    """numpy Poisson background + Gaussian spots, uint16 [B, (Z,) H, W]."""
    rng = np.random.default_rng(seed)
    N, B = codebook.shape
    img = rng.poisson(bg, size=(B,) + tuple(shape)).astype(np.float64)
    H, W = shape[-2:]
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(n_spots):
        b = rng.integers(N)
        cy, cx = rng.uniform(2, H - 2), rng.uniform(2, W - 2)
        a = rng.uniform(*amp)
        spot = a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * psf ** 2))
        on = np.nonzero(codebook[b])[0]
        if len(shape) == 3:
            zc = rng.integers(shape[0])
            for zz in range(max(0, zc - 1), min(shape[0], zc + 2)):
                img[on, zz] += (1.0 if zz == zc else 0.5) * spot
        else:
            img[on] += spot
    return np.clip(np.rint(img), 0, 65535).astype(np.uint16)
