"""GPU: BASELINE.json's full-size configuration (2048 x 2048 x 7 z, 16-bit MHD4
codebook) through size-independent properties, plus oracle checks on crops."""
import os

import numpy as np
import pytest

import merfish_oracle as oracle
from conftest import ArrayCodebook, GOLDEN_DIR

pytestmark = pytest.mark.gpu

Z, H, W = 7, 2048, 2048


@pytest.fixture(scope='module')
def fov(gpu_device):
    import torch
    from merlin_amd.decode import Decode
    from merlin_amd.synthetic import synthetic_fov
    cb = np.load(os.path.join(GOLDEN_DIR, 'testsuite_inputs.npz'))['codebook']
    img = synthetic_fov(cb, Z, H, W, 40000, 4321, gpu_device)
    rng = np.random.default_rng(9)
    sf = rng.uniform(110, 150, 16)
    bg = rng.uniform(0, 5, 16)
    task = Decode(ArrayCodebook(cb), {'crop_width': 100}, device=gpu_device)
    bc, vol = task.run_fov(img, sf, bg, fov=3)
    # run_fov keeps the traces of decoded pixels only (pixel-major); a second
    # launch with the reference's full plane-major traces
    full = task.decode_fov_device(img, sf, bg, returnTraces=True)
    torch.cuda.synchronize()
    return dict(cb=cb, img=img, sf=sf, bg=bg, task=task, bc=bc, vol=vol,
                full=full)


def test_crops_match_oracle(fov):
    """The blur is local: a crop decoded by the oracle equals the full decode
    away from the crop's own border (interior, edge and corner crops)."""
    dm = oracle.normalized_codebook(fov['cb'])
    vol = fov['vol']
    for z, y0, x0 in ((0, 0, 0), (3, 1000, 900), (6, H - 160, W - 200),
                      (2, 0, 1500), (5, 1890, 0)):
        y1, x1 = min(y0 + 160, H), min(x0 + 200, W)
        crop = fov['img'][:, z, y0:y1, x0:x1].cpu().numpy()
        di, pm, npt, d, gap = oracle.decode_pixels(
            crop, dm, fov['sf'], fov['bg'], 0.5167, 1, 1, return_gap=True)
        # rows/cols whose 5x5 blur window lies inside the crop or on a real
        # image border (where crop and image reflect identically)
        ya = 0 if y0 == 0 else 2
        yb = (y1 - y0) if y1 == H else (y1 - y0 - 2)
        xa = 0 if x0 == 0 else 2
        xb = (x1 - x0) if x1 == W else (x1 - x0 - 2)
        sl = (slice(ya, yb), slice(xa, xb))
        gs = (z, slice(y0 + ya, y0 + yb), slice(x0 + xa, x0 + xb))
        assert np.array_equal(vol.decoded[gs].cpu().numpy(), di[sl])
        assert np.array_equal(vol.magnitudes[gs].cpu().numpy(), pm[sl])
        want = npt[(slice(None),) + sl]
        assert np.array_equal(
            fov['full'].traces[z, :, gs[1], gs[2]].cpu().numpy(), want)
        hit = di[sl] >= 0
        assert vol.traces_decoded_only and hit.any()
        assert np.array_equal(
            vol.traces[z, :, gs[1], gs[2]].cpu().numpy()[:, hit], want[:, hit])
        got = vol.distances[gs].cpu().numpy()
        clear = (gap[sl] > 1e-4) | (di[sl] >= 0)
        assert np.array_equal(got[clear], d[sl][clear])
        np.testing.assert_allclose(got, d[sl], rtol=4e-6)


def test_planes_are_independent(fov):
    """One launch over 7 planes == 7 single-plane launches (2-D mode)."""
    import torch
    dec = fov['task'].decoder
    for z in (0, 4):
        one = dec.decode_pixels_device(fov['img'][:, z], fov['sf'], fov['bg'],
                                       0.5167, 1, 1)
        assert torch.equal(one.decoded[0], fov['vol'].decoded[z])
        assert torch.equal(one.magnitudes[0], fov['vol'].magnitudes[z])
        assert torch.equal(one.distances[0], fov['vol'].distances[z])
        assert torch.equal(one.traces[0], fov['full'].traces[z])
        assert torch.equal(one.decoded[0], fov['full'].decoded[z])
        hit = one.decoded[0] >= 0
        assert torch.equal(one.traces[0][:, hit], fov['vol'].traces[z][:, hit])


def test_component_and_record_invariants(fov):
    import torch
    vol, dec = fov['vol'], fov['task'].decoder
    table = dec.region_table(vol, False)
    labels = vol.labels
    decoded = vol.decoded
    fg = decoded >= 0
    # every assigned pixel is labelled, the label is a pixel of the same code
    # that is not after it in raster order, and roots label themselves
    assert torch.equal(labels >= 0, fg)
    lin = torch.arange(decoded.numel(), device=decoded.device).view_as(decoded)
    lab = labels[fg].long()
    assert bool((lab <= lin[fg]).all())
    assert torch.equal(decoded.view(-1)[lab], decoded[fg])
    assert torch.equal(labels.view(-1)[lab].long(), lab)        # idempotent
    assert bool((lab // (H * W) == lin[fg] // (H * W)).all())   # 2-D: same plane
    # 8-neighbours with the same code share a label
    for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):
        a = decoded[:, :H - dy, max(0, -dx):W - max(0, dx)]
        b = decoded[:, dy:, max(0, dx):W + min(0, dx)]
        la = labels[:, :H - dy, max(0, -dx):W - max(0, dx)]
        lb = labels[:, dy:, max(0, dx):W + min(0, dx)]
        same = (a == b) & (a >= 0)
        assert torch.equal(la[same], lb[same])
    rec = table.rec
    # areas add up to the assigned pixels; one row per root; reference order
    assert int(rec[:, 2].sum()) == int(fg.sum())
    roots = torch.nonzero(labels.view(-1) == lin.view(-1)).squeeze(1)
    assert len(rec) == roots.numel()
    order = np.lexsort((rec[:, 10], rec[:, 0], rec[:, 1]))
    assert np.array_equal(order, np.arange(len(rec)))
    assert np.array_equal(np.sort(rec[:, 10]).astype(np.int64),
                          roots.cpu().numpy())
    first = torch.from_numpy(rec[:, 10].astype(np.int64)).to(decoded.device)
    assert np.array_equal(decoded.view(-1)[first].cpu().numpy(),
                          rec[:, 0].astype(np.int16))
    area = torch.bincount(lab, minlength=1)[first].cpu().numpy()
    assert np.array_equal(area, rec[:, 2].astype(np.int64))
    # centroids inside the image, intensities consistent
    assert (rec[:, 9] >= 0).all() and (rec[:, 9] <= W - 1).all()
    assert (rec[:, 8] >= 0).all() and (rec[:, 8] <= H - 1).all()
    assert (rec[:, 3] <= rec[:, 4] * (1 + 1e-6)).all()          # mean <= max
    assert (rec[:, 6] <= rec[:, 5] + 1e-12).all()               # min <= mean
    assert (rec[:, 6] <= 0.5167).all()
    # the DataFrame is the crop/area filter of the table
    bc = fov['bc']
    keep = ((rec[:, 9] > 100) & (rec[:, 9] < H - 100) & (rec[:, 8] > 100)
            & (rec[:, 8] < W - 100))
    assert len(bc) == int(keep.sum())
    assert np.array_equal(bc['area'].to_numpy(), rec[keep, 2])
    assert np.array_equal(bc['z'].to_numpy(), rec[keep, 1])
    assert np.array_equal(bc['intensity_5'].to_numpy(), table.inten[keep, 5])
    assert (bc['fov'] == 3).all() and (bc['cell_index'] == -1).all()


def test_spot_recall(fov):
    """Sanity of the whole path: the number of barcodes found is of the order
    of the 40 000 spots drawn per plane."""
    per_plane = fov['bc'].groupby('z').size()
    assert len(per_plane) == Z
    assert per_plane.min() > 20000 and per_plane.max() < 60000
