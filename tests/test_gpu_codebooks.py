"""GPU: BASELINE.json configs 3 and 5 -- the large-codebook (MFMA, codebook
streamed through LDS) nearest-codeword path of decode_pixels
(merlin/util/decoding.py:112-121) at the codebooks SURVEY 8d names:

  config 3: 22 bits x 1 000 weight-4 codewords (seed 7), one 2048 x 2048 plane
  config 5: 32 bits x 10 000 random weight-4 codewords (seed 11, minimum
            distance NOT enforced), 2048 x 2048 x 10 z planes

Small images against the oracle's brute force (bit-exact bar, as in
test_gpu_parity), exact-tie images (a random codebook has many codewords at
equal distance: the exact-scan branch of the kernel decides, lowest index
wins, as the oracle), and the full sizes through size-independent properties
(planes independent, crops vs the oracle, region-table invariants)."""
import warnings

import numpy as np
import pytest

import merfish_oracle as oracle
from conftest import ArrayCodebook, frame_to_array, synth_stack
from test_gpu_parity import assert_distances

pytestmark = pytest.mark.gpu

CONFIGS = {
    'cfg3': dict(bits=22, words=1000, seed=7, z=1),
    'cfg5': dict(bits=32, words=10000, seed=11, z=10),
}


def _codebook(name):
    from merlin_amd.codebook import random_weight_codebook
    c = CONFIGS[name]
    return random_weight_codebook(c['bits'], c['words'], 4,
                                  seed=c['seed']).get_barcodes()


@pytest.mark.parametrize('name', ['cfg3', 'cfg5'])
def test_small_image_against_oracle(name, gpu_device):
    """decode_pixels + one barcode's records, 48 x 80, brute-force oracle."""
    from merlin_amd.decoding import PixelBasedDecoder
    cb = _codebook(name)
    N, B = cb.shape
    rng = np.random.default_rng(3)
    img = synth_stack(cb, (48, 80), 40, seed=17)
    sf = rng.uniform(200, 600, B)
    bg = rng.uniform(50, 120, B)
    dm = oracle.normalized_codebook(cb)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    di, pm, npt, d = dec.decode_pixels(img, sf, bg, 0.5176, 1, 1)
    odi, opm, onpt, od, gap = oracle.decode_pixels(img, dm, sf, bg, 0.5176, 1,
                                                   1, return_gap=True)
    assert np.array_equal(pm, opm)
    assert np.array_equal(npt, onpt)
    assert np.array_equal(di, odi)
    assert (odi >= 0).sum() > 50
    assert_distances(d, od, gap, odi)
    # the records of the five most frequent barcodes (the crop compares x with
    # shape[0] = 48, the reference's quirk, so some of them lose every row)
    rows = 0
    for i in np.argsort(-np.bincount(odi[odi >= 0]))[:5]:
        got = dec.extract_barcodes_with_index(int(i), di, pm, npt, d, 0, 0, 0)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            exp = oracle.extract_barcodes_with_index(int(i), odi, opm, onpt, od,
                                                     0, 0, 0)
        r, e = frame_to_array(got, B), frame_to_array(exp, B)
        assert r.shape == e.shape
        np.testing.assert_allclose(r[:, :14], e[:, :14], rtol=1e-12, atol=1e-12)
        assert np.array_equal(r[:, 14:], e[:, 14:])
        rows += len(e)
    assert rows > 0


def _mixed_codebook(bits, words, seed):
    """Distinct random codewords of weight 3, 4 or 5 (a non-uniform codebook:
    the per-codeword value and half-norm tables of the kernel), sorted."""
    rng = np.random.default_rng(seed)
    seen = set()
    while len(seen) < words:
        w = int(rng.integers(3, 6))
        seen.add(tuple(sorted(rng.choice(bits, w, replace=False))))
    mat = np.zeros((words, bits), dtype=np.int64)
    for i, w in enumerate(sorted(seen)):
        mat[i, list(w)] = 1
    return mat


@pytest.mark.parametrize('bits,words', [(32, 4000), (27, 3100)])
def test_mixed_weight_large_codebook_against_oracle(bits, words, gpu_device):
    """More than 3 072 codewords of mixed weight: the untagged search of the
    kernel (the winner is re-scored among the 16 codewords of the recorded
    tile) on the non-uniform tables, with a codebook streamed in chunks whose
    last tile is partly padding; negative and all-zero traces included (a
    padding row must not win)."""
    from merlin_amd.decoding import PixelBasedDecoder
    cb = _mixed_codebook(bits, words, seed=bits)
    N, B = cb.shape
    rng = np.random.default_rng(4)
    img = synth_stack(cb, (40, 72), 40, seed=23)
    img[:, :6, :8] = 0                    # nothing there: every sum is zero
    sf = rng.uniform(200, 600, B)
    bg = rng.uniform(50, 120, B)          # dark pixels go negative
    dm = oracle.normalized_codebook(cb)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    di, pm, npt, d = dec.decode_pixels(img, sf, bg, 0.5176, 1, 1)
    odi, opm, onpt, od, gap = oracle.decode_pixels(img, dm, sf, bg, 0.5176, 1,
                                                   1, return_gap=True)
    assert np.array_equal(pm, opm)
    assert np.array_equal(npt, onpt)
    assert np.array_equal(di, odi)
    assert (odi >= 0).sum() > 50
    assert_distances(d, od, gap, odi)


def test_uniform_large_codebook_dark_pixels(gpu_device):
    """Config 5's codebook on an image whose background exceeds the signal
    almost everywhere (negative sums: the zero padding rows of the table hold
    the smallest key) and on empty pixels."""
    from merlin_amd.decoding import PixelBasedDecoder
    cb = _codebook('cfg5')
    N, B = cb.shape
    img = synth_stack(cb, (40, 72), 25, seed=29)
    img[:, 20:, :] = 0
    sf = np.full(B, 300.0)
    bg = np.full(B, 400.0)
    dm = oracle.normalized_codebook(cb)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    di, pm, npt, d = dec.decode_pixels(img, sf, bg, 0.5176, 1, 1)
    odi, opm, onpt, od, gap = oracle.decode_pixels(img, dm, sf, bg, 0.5176, 1,
                                                   1, return_gap=True)
    assert np.array_equal(pm, opm)
    assert np.array_equal(npt, onpt)
    assert np.array_equal(di, odi)
    assert_distances(d, od, gap, odi)


def _tie_image(cb, rng, blocks=6, size=10):
    """uint16 [B, H, W]: flat blocks whose on-bits all carry the SAME value,
    five on-bits per block chosen so that at least two codewords are subsets
    of them: after the (identical, exact integer) blur of every on-bit plane
    the traces have five equal components and those codewords are at exactly
    the same distance (0.4595 < 0.5176, so the pixel decodes)."""
    N, B = cb.shape
    words = {tuple(np.nonzero(r)[0]) for r in cb}
    H = W = blocks * size
    img = np.zeros((B, H, W), dtype=np.uint16)
    n_tied = []
    for by in range(blocks):
        for bx in range(blocks):
            for _ in range(1000):
                five = tuple(sorted(rng.choice(B, 5, replace=False)))
                subs = [tuple(b for b in five if b != drop) for drop in five]
                k = sum(s in words for s in subs)
                if k >= 2:
                    break
            else:
                raise AssertionError('no tied 5-bit set found')
            n_tied.append(k)
            v = int(rng.integers(800, 40000))
            ys, xs = by * size + 2, bx * size + 2
            img[list(five), ys:ys + size - 4, xs:xs + size - 4] = v
    return img, n_tied


@pytest.mark.parametrize('name', ['cfg3', 'cfg5'])
def test_exact_ties_lowest_index_wins(name, gpu_device):
    """Exactly tied codewords within the distance threshold: the decoded
    index is the oracle's (lowest index), distances bit-exact."""
    from merlin_amd.decoding import PixelBasedDecoder
    cb = _codebook(name)
    N, B = cb.shape
    rng = np.random.default_rng(5)
    img, n_tied = _tie_image(cb, rng)
    sf = np.full(B, 300.0)
    bg = np.zeros(B)
    dm = oracle.normalized_codebook(cb)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    di, pm, npt, d = dec.decode_pixels(img, sf, bg, 0.5176, 1, 1)
    odi, opm, onpt, od, gap = oracle.decode_pixels(img, dm, sf, bg, 0.5176, 1,
                                                   1, return_gap=True)
    tied = (gap == 0) & (odi >= 0)
    assert tied.sum() > 200            # the case is what it claims to be
    assert np.array_equal(pm, opm)
    assert np.array_equal(npt, onpt)
    assert np.array_equal(di, odi)
    assert np.array_equal(d[odi >= 0], od[odi >= 0])
    assert_distances(d, od, gap, odi)


@pytest.fixture(scope='module', params=['cfg3', 'cfg5'])
def full(request, gpu_device):
    import torch
    from merlin_amd.decode import Decode
    from merlin_amd.synthetic import synthetic_fov
    name = request.param
    c = CONFIGS[name]
    cb = _codebook(name)
    Z, H, W = c['z'], 2048, 2048
    img = synthetic_fov(cb, Z, H, W, 40000, 1234 + c['seed'], gpu_device)
    B = c['bits']
    rng = np.random.default_rng(c['seed'])
    sf = rng.uniform(400, 600, B)
    bg = rng.uniform(0, 5, B)
    task = Decode(ArrayCodebook(cb), {'crop_width': 100}, device=gpu_device)
    bc, vol = task.run_fov(img, sf, bg, fov=1)
    torch.cuda.synchronize()
    yield dict(name=name, cb=cb, img=img, sf=sf, bg=bg, task=task, bc=bc,
               vol=vol, Z=Z, H=H, W=W)
    del img, vol, bc, task
    torch.cuda.empty_cache()


def test_full_size_crops_match_oracle(full):
    """The blur is local: a crop decoded by the oracle equals the full-size
    decode away from the crop's own border."""
    dm = oracle.normalized_codebook(full['cb'])
    vol, H, W, Z = full['vol'], full['H'], full['W'], full['Z']
    crops = [(0, 0, 0), (Z - 1, H - 40, W - 72)] if full['name'] == 'cfg5' \
        else [(0, 0, 0), (0, 1000, 900), (0, H - 40, W - 72)]
    decoded_any = 0
    for z, y0, x0 in crops:
        y1, x1 = min(y0 + 40, H), min(x0 + 72, W)
        crop = full['img'][:, z, y0:y1, x0:x1].cpu().numpy()
        di, pm, npt, d, gap = oracle.decode_pixels(
            crop, dm, full['sf'], full['bg'], 0.5167, 1, 1, return_gap=True)
        ya = 0 if y0 == 0 else 2
        yb = (y1 - y0) if y1 == H else (y1 - y0 - 2)
        xa = 0 if x0 == 0 else 2
        xb = (x1 - x0) if x1 == W else (x1 - x0 - 2)
        sl = (slice(ya, yb), slice(xa, xb))
        gs = (z, slice(y0 + ya, y0 + yb), slice(x0 + xa, x0 + xb))
        assert np.array_equal(vol.decoded[gs].cpu().numpy(), di[sl])
        assert np.array_equal(vol.magnitudes[gs].cpu().numpy(), pm[sl])
        hit = di[sl] >= 0
        decoded_any += int(hit.sum())
        want = npt[(slice(None),) + sl]
        assert np.array_equal(
            vol.traces[z, :, gs[1], gs[2]].cpu().numpy()[:, hit], want[:, hit])
        got = vol.distances[gs].cpu().numpy()
        clear = (gap[sl] > 1e-4) | hit
        assert np.array_equal(got[clear], d[sl][clear])
        np.testing.assert_allclose(got, d[sl], rtol=4e-6)
    assert decoded_any > 20


def test_full_size_planes_independent(full):
    """One launch over Z planes == single-plane launches (2-D mode)."""
    import torch
    dec = full['task'].decoder
    for z in sorted({0, full['Z'] - 1}):
        one = dec.decode_pixels_device(full['img'][:, z], full['sf'],
                                       full['bg'], 0.5167, 1, 1,
                                       returnTraces=False)
        assert torch.equal(one.decoded[0], full['vol'].decoded[z])
        assert torch.equal(one.magnitudes[0], full['vol'].magnitudes[z])
        assert torch.equal(one.distances[0], full['vol'].distances[z])


def test_full_size_region_table_invariants(full):
    import torch
    vol, dec = full['vol'], full['task'].decoder
    H, W = full['H'], full['W']
    table = dec.region_table(vol, False)
    decoded = vol.decoded
    fg = decoded >= 0
    rec = table.rec
    assert int(rec[:, 2].sum()) == int(fg.sum())
    order = np.lexsort((rec[:, 10], rec[:, 0], rec[:, 1]))
    assert np.array_equal(order, np.arange(len(rec)))
    first = torch.from_numpy(rec[:, 10].astype(np.int64)).to(decoded.device)
    assert np.array_equal(decoded.view(-1)[first].cpu().numpy(),
                          rec[:, 0].astype(np.int16))
    assert (rec[:, 6] <= 0.5167).all() and (rec[:, 6] <= rec[:, 5] + 1e-12).all()
    bc = full['bc']
    keep = ((rec[:, 9] > 100) & (rec[:, 9] < H - 100) & (rec[:, 8] > 100)
            & (rec[:, 8] < W - 100))
    assert len(bc) == int(keep.sum()) and len(bc) > 1000 * full['Z']
    assert np.array_equal(bc['area'].to_numpy(), rec[keep, 2])
    assert np.array_equal(bc['barcode_id'].to_numpy(),
                          rec[keep, 0].astype(np.int64))
