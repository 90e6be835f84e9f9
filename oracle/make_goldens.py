"""Generate golden vectors by running the REFERENCE's own decoding.py.

Run in the build container only (the reference does not travel):

    /opt/conda/bin/python3.9 oracle/make_goldens.py

The reference module ``merlin/util/decoding.py`` is imported unmodified from
/root/reference with harness-level shims for what this container lacks:

* ``merlin`` is registered as a bare namespace package so that
  ``merlin/__init__.py`` (needs python-dotenv) is not executed;
* ``cv2`` is absent from every interpreter here, so a stub module exposes
  ``GaussianBlur`` = the oracle's restatement (``oracle/merfish_oracle.py``).
  Consequence: the goldens pin everything AFTER the blur against the real
  reference code; the blur itself is pinned only by self-consistency
  ("parity unpinned" for that stage);
* ``np.int`` (removed in numpy 1.24), ``pandas.concat(objs, 1)`` positional
  axis and ``Series.between(inclusive=False)`` are mapped to their modern
  spellings.

Outputs: ``tests/golden/*.npz`` (inputs + expected outputs; data only).
"""
import os
import sys
import types

import numpy as np
import pandas

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, HERE)
import merfish_oracle as oracle  # noqa: E402


def install_shims():
    pkg = types.ModuleType('merlin')
    pkg.__path__ = [os.path.join(REF, 'merlin')]
    sys.modules['merlin'] = pkg

    cv2 = types.ModuleType('cv2')

    def GaussianBlur(src, ksize, sigmaX):
        assert ksize[0] == ksize[1]
        if src.dtype == np.uint16:
            return oracle.gaussian_blur_u16(np.ascontiguousarray(src),
                                            ksize[0], sigmaX)
        return oracle.gaussian_blur_f32(
            np.ascontiguousarray(src, dtype=np.float32), ksize[0], sigmaX)
    cv2.GaussianBlur = GaussianBlur
    sys.modules['cv2'] = cv2

    if not hasattr(np, 'int'):
        np.int = int

    _concat = pandas.concat

    def concat(objs, axis=0, *args, **kwargs):
        return _concat(objs, axis=axis, *args, **kwargs)
    pandas.concat = concat

    _between = pandas.Series.between

    def between(self, left, right, inclusive='both'):
        if inclusive is False:
            inclusive = 'neither'
        elif inclusive is True:
            inclusive = 'both'
        return _between(self, left, right, inclusive=inclusive)
    pandas.Series.between = between


class ArrayCodebook:
    """Stands in for merlin.data.codebook.Codebook: the decoder only calls
    get_barcodes() (decoding.py:270)."""

    def __init__(self, barcodes):
        self._b = np.asarray(barcodes)

    def get_barcodes(self, ignoreBlanks=False):
        return self._b


class AffineAligner:
    """globalalign.py:141-148 with a fixed float32 3x3 transform."""

    def __init__(self, tform):
        self.t = tform

    def fov_coordinate_array_to_global(self, fov, fovCoordArray):
        toGlobal = np.ones(fovCoordArray.shape)
        toGlobal[:, [0, 1]] = fovCoordArray[:, [1, 2]]
        g = np.matmul(self.t, toGlobal.T).T[:, [2, 0, 1]]
        g[:, 0] = fovCoordArray[:, 0]
        return g


def load_test_codebook(name='test_codebook.csv'):
    df = pandas.read_csv(os.path.join(REF, 'test', 'auxiliary_files', name))
    bits = [c for c in df.columns if c not in ('name', 'id')]
    return df[bits].to_numpy().astype(np.int64)


def load_test_fov(fov):
    """Assemble u16[16,128,128] from test_{fov}_{round}.tif following
    test_data_organization.csv (bit -> imagingRound, frame) and
    test_microscope_parameters.json (transpose, flip_vertical)."""
    import tifffile
    org = pandas.read_csv(os.path.join(
        REF, 'test', 'auxiliary_files', 'test_data_organization.csv'))
    planes = []
    for i in range(16):
        row = org[org['channelName'] == 'bit{}'.format(i + 1)].iloc[0]
        img = tifffile.imread(os.path.join(
            REF, 'test', 'auxiliary_files',
            'test_{}_{}.tif'.format(fov, int(row['imagingRound']))))
        frame = img[int(row['frame'])]
        frame = np.flip(np.transpose(frame), axis=0)
        planes.append(frame)
    return np.ascontiguousarray(np.array(planes, dtype=np.uint16))


def synthetic_stack(codebook, shape, n_spots, seed, z=None, amp=(1000, 5000),
                    bg=100, psf=1.2):
    """Poisson background + Gaussian spots on the on-bits of a random barcode
    (parameters after util/simulator.py:18-28)."""
    rng = np.random.default_rng(seed)
    N, B = codebook.shape
    full = (B,) + tuple(shape)
    img = rng.poisson(bg, size=full).astype(np.float64)
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
                f = 1.0 if zz == zc else 0.5
                img[on, zz] += f * spot * rng.uniform(0.7, 1.3, size=(len(on), 1, 1))
        else:
            img[on] += spot * rng.uniform(0.7, 1.3, size=(len(on), 1, 1))
    return np.clip(np.rint(img), 0, 65535).astype(np.uint16)


def frame_to_arrays(df, B):
    cols = oracle.COLUMN_NAMES + ['intensity_{}'.format(i) for i in range(B)]
    if len(df) == 0:
        return np.zeros((0, len(cols)), dtype=np.float64)
    return df[cols].to_numpy().astype(np.float64)


def run_case(name, decoding, codebook, image, sf=None, bg=None, dthr=0.5176,
             mthr=1, sigma=1, crop=0, min_area=0, fov=0, z_index=0,
             tform=None, refactor_area=4, with_backgrounds=True,
             decode3d=False):
    N, B = codebook.shape
    dec = decoding.PixelBasedDecoder(ArrayCodebook(codebook))
    dec.refactorAreaThreshold = refactor_area
    aligner = AffineAligner(tform) if tform is not None else None
    out = dict(codebook=codebook, image=image,
               sf=np.ones(B) if sf is None else sf,
               bg=np.zeros(B) if bg is None else bg,
               params=np.array([dthr, mthr, sigma, crop, min_area, fov,
                                -1 if decode3d else z_index, refactor_area,
                                int(decode3d)], dtype=np.float64))
    if tform is not None:
        out['tform'] = tform
    if not decode3d:
        di, pm, npt, d = dec.decode_pixels(image, sf, bg, dthr, mthr, sigma)
        tables = [dec.extract_barcodes_with_index(
            i, di, pm, npt, d, fov, crop, z_index, aligner, min_area)
            for i in range(N)]
        rf, bgr, seen = dec.extract_refactors(di, pm, npt, False)
        out.update(refactors=rf, barcodes_seen=seen)
        if with_backgrounds:
            rf2, bgr2, seen2 = dec.extract_refactors(di, pm, npt, True)
            out.update(refactors_bg=rf2, background_refactors=bgr2,
                       barcodes_seen_bg=seen2)
    else:
        Z = image.shape[1]
        di = np.zeros((Z,) + image.shape[2:], dtype=np.int16)
        pm = np.zeros(di.shape, dtype=np.float32)
        d = np.zeros(di.shape, dtype=np.float32)     # analysis/decode.py:116
        npt = np.zeros((Z, B) + image.shape[2:], dtype=np.float32)
        for z in range(Z):
            a, b_, c, e = dec.decode_pixels(image[:, z], sf, bg, dthr, mthr,
                                            sigma)
            di[z], pm[z], npt[z], d[z] = a, b_, c, e
        tables = [dec.extract_barcodes_with_index(
            i, di, pm, npt, d, fov, crop, None, aligner, min_area)
            for i in range(N)]
    records = np.concatenate([frame_to_arrays(t, B) for t in tables])
    out.update(decoded=di, magnitudes=pm, traces=npt, distances=d,
               records=records)
    assert di.dtype == np.int16 and pm.dtype == np.float32
    assert npt.dtype == np.float32
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print('{:28s} records={:5d} decoded_px={:7d} size={:.0f} KB'.format(
        name, len(records), int((di >= 0).sum()),
        os.path.getsize(path) / 1024))


def main():
    install_shims()
    from merlin.util import decoding
    os.makedirs(OUT, exist_ok=True)
    cb = load_test_codebook()

    # raw reference fixture inputs, shipped as data for the GPU-side tests
    fovs = np.array([load_test_fov(0), load_test_fov(1)])
    np.savez_compressed(os.path.join(OUT, 'testsuite_inputs.npz'),
                        images=fovs, codebook=cb,
                        codebook2=load_test_codebook('test_codebook2.csv'))

    # G1: test-suite FOVs, initial scale factors from the summed histograms
    hist = np.zeros((16, 65534))
    for f in fovs:
        for b in range(16):
            hist[b] += np.histogram(f[b], bins=np.arange(0, 65535, 1))[0]
    sf0 = np.zeros(16)
    for b in range(16):
        c = np.cumsum(hist[b])
        c = c / c[-1]
        sf0[b] = np.argmin(np.abs(c - 0.9)) + 2
    tform = np.float32([[0.108, 0, 15.5], [0, 0.108, -7.25], [0, 0, 1]])
    # raw frames carry a ~2000-count camera offset (no high-pass here, the
    # preprocess stage is out of scope): hand it to the decoder as backgrounds
    bg0 = np.array([np.median(fovs[:, b]) for b in range(16)])
    for f in range(2):
        run_case('g1_testsuite_fov{}'.format(f), decoding, cb, fovs[f],
                 sf=sf0 - bg0, bg=bg0, dthr=0.5167, crop=10, fov=f,
                 tform=tform)
    run_case('g1_testsuite_fov0_defaults', decoding, cb, fovs[0], fov=0,
             with_backgrounds=False)

    # G2: synthetic 192^2, sf = 500-ish, bg != 0
    rng = np.random.default_rng(5)
    img = synthetic_stack(cb, (192, 192), 260, seed=0)
    run_case('g2_synth192', decoding, cb, img,
             sf=rng.uniform(300, 700, 16), bg=rng.uniform(80, 120, 16),
             dthr=0.5167, crop=5, min_area=2, fov=3, z_index=2,
             refactor_area=5)

    # G3: edge cases: zero pixels, single-pixel regions, diagonal contacts,
    # regions on the crop boundary, saturated values
    img = np.zeros((16, 64, 64), dtype=np.uint16)
    on = [np.nonzero(cb[i])[0] for i in range(cb.shape[0])]
    img[on[3], 10, 10] = 4000                       # isolated pixel
    for k in range(6):                              # diagonal chain
        img[on[7], 20 + 2 * k, 20 + 2 * k] = 3000 + 100 * k
    img[on[7][:, None], 40:44, 5:9] = 6000          # block near the edge
    img[on[9][:, None], 60:64, 60:64] = 65535       # saturated corner block
    img[on[11][:, None], 30:33, 0:3] = 2500         # touching the border
    img[:, 50:54, 30:34] = 1200                     # uniform: far from any code
    run_case('g3_edges', decoding, cb, img, sf=np.full(16, 100.0),
             dthr=0.5176, crop=2, fov=1, z_index=0, refactor_area=1)

    # G4: 22-bit, 300 weight-4 codewords (not MHD4: ties/near-codes allowed)
    rng = np.random.default_rng(7)
    words = set()
    while len(words) < 300:
        words.add(tuple(sorted(rng.choice(22, 4, replace=False))))
    cb22 = np.zeros((300, 22), dtype=np.int64)
    for i, w in enumerate(sorted(words)):
        cb22[i, list(w)] = 1
    img = synthetic_stack(cb22, (96, 96), 120, seed=3)
    run_case('g4_22bit', decoding, cb22, img, sf=np.full(22, 400.0),
             dthr=0.5167, crop=0, fov=0, z_index=1, with_backgrounds=False)

    # G5: decode_3d, Z=3, 64^2 (analysis/decode.py:129-161)
    # NB the reference crops x against shape[0] = Z in 3-D (decoding.py:245),
    # so Z is made large enough for records to survive that quirk.
    img = synthetic_stack(cb, (20, 48, 48), 60, seed=11)
    run_case('g5_decode3d', decoding, cb, img, sf=np.full(16, 450.0),
             dthr=0.5167, crop=1, fov=2, decode3d=True)

    # G6: mixed-weight codebook (rows of weight 3..5 and one all-zero row)
    rng = np.random.default_rng(13)
    cbm = np.zeros((40, 12), dtype=np.int64)
    for i in range(1, 40):
        cbm[i, rng.choice(12, rng.integers(3, 6), replace=False)] = 1
    img = synthetic_stack(cbm[1:], (64, 64), 50, seed=17)
    run_case('g6_mixed_weight', decoding, cbm, img, sf=np.full(12, 300.0),
             sigma=1.5, dthr=0.6, with_backgrounds=False)


if __name__ == '__main__':
    main()
