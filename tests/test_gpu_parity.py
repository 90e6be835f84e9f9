"""GPU: the HIP path, called through the C-ABI (via the PixelBasedDecoder
drop-in), against the oracle and the reference-generated golden vectors.

Bar: decoded indexes, connected components, areas and every integer column are
bit-exact; float32 magnitudes/traces and float64 distances are bit-exact
against the oracle (same operation order) and within float32 rounding of the
reference goldens (whose norm goes through BLAS); record floats within 1e-12
relative of the oracle's float64 sums (summation order) and 1e-6 for float32
means."""
import warnings

import numpy as np
import pytest

import merfish_oracle as oracle
from conftest import (ArrayCodebook, AffineAligner, frame_to_array,
                      golden_names, load_golden, synth_stack)

pytestmark = pytest.mark.gpu


def _decoder(g, gpu_device):
    from merlin_amd.decoding import PixelBasedDecoder
    dec = PixelBasedDecoder(ArrayCodebook(g['codebook']), device=gpu_device)
    dec.refactorAreaThreshold = g['p']['refactor_area']
    return dec


def _decode_gpu(dec, g):
    p = g['p']
    img = g['image']
    if not p['decode3d']:
        return dec.decode_pixels(img, g['sf'], g['bg'], p['dthr'], p['mthr'],
                                 p['sigma'])
    outs = [dec.decode_pixels(img[:, z], g['sf'], g['bg'], p['dthr'],
                              p['mthr'], p['sigma'])
            for z in range(img.shape[1])]
    return (np.array([o[0] for o in outs]), np.array([o[1] for o in outs]),
            np.array([o[2] for o in outs]),
            np.array([o[3] for o in outs]).astype(np.float32))


def _decode_oracle(g):
    """oracle decode incl. the squared-distance gap to the runner-up."""
    p = g['p']
    dm = oracle.normalized_codebook(g['codebook'])
    img = g['image']
    if not p['decode3d']:
        return oracle.decode_pixels(img, dm, g['sf'], g['bg'], p['dthr'],
                                    p['mthr'], p['sigma'], return_gap=True)
    outs = [oracle.decode_pixels(img[:, z], dm, g['sf'], g['bg'], p['dthr'],
                                 p['mthr'], p['sigma'], return_gap=True)
            for z in range(img.shape[1])]
    return (np.array([o[0] for o in outs]), np.array([o[1] for o in outs]),
            np.array([o[2] for o in outs]),
            np.array([o[3] for o in outs]).astype(np.float32),
            np.array([o[4] for o in outs]))


def assert_distances(d, od, gap, decoded):
    """Distances are bit-exact wherever the winner is unambiguous (runner-up
    more than 1e-4 away in squared distance, i.e. 5e-5 in score) or the pixel
    decodes; an ambiguous far pixel may report either near-tied codeword, so
    its distance agrees to 2^-18 = 4e-6 relative: the slot bits of the search
    key plus float32-score rounding (documented in include/merfish_hip.h)."""
    clear = (gap > 1e-4) | (decoded >= 0)
    assert np.array_equal(d[clear], od[clear])
    np.testing.assert_allclose(d[~clear], od[~clear], rtol=4e-6, atol=0)


@pytest.mark.parametrize('name', golden_names())
def test_decode_pixels_golden(name, gpu_device):
    g = load_golden(name)
    dec = _decoder(g, gpu_device)
    di, pm, npt, d = _decode_gpu(dec, g)
    assert di.dtype == np.int16 and pm.dtype == np.float32
    assert npt.dtype == np.float32
    # vs the reference's own output
    assert np.array_equal(di, g['decoded'])
    np.testing.assert_allclose(pm, g['magnitudes'], rtol=1e-6, atol=0)
    np.testing.assert_allclose(npt, g['traces'], rtol=0, atol=5e-7)
    np.testing.assert_allclose(d, g['distances'], rtol=1e-6, atol=2e-7)
    # vs the oracle: same operation order, so bit-exact
    odi, opm, onpt, od, gap = _decode_oracle(g)
    assert np.array_equal(di, odi)
    assert np.array_equal(pm, opm)
    assert np.array_equal(npt, onpt)
    assert_distances(d, od, gap, odi)


@pytest.mark.parametrize('name', golden_names())
def test_extract_barcodes_golden(name, gpu_device):
    """extract_barcodes_with_index over every index on the GOLDEN images
    (uploaded as numpy arrays, as the Decode task passes them)."""
    g = load_golden(name)
    p = g['p']
    N, B = g['codebook'].shape
    dec = _decoder(g, gpu_device)
    al = AffineAligner(g['tform']) if 'tform' in g else None
    frames = [dec.extract_barcodes_with_index(
        i, g['decoded'], g['magnitudes'], g['traces'], g['distances'],
        p['fov'], p['crop'], p['z_index'], al, p['min_area'])
        for i in range(N)]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        import pandas
        r = frame_to_array(pandas.concat(frames), B)
    exp = g['records']
    assert r.shape == exp.shape
    # integer-valued columns: barcode_id, fov, area, cell_index
    for c in (0, 1, 4, 13):
        assert np.array_equal(r[:, c], exp[:, c])
    # max_intensity, min_distance are selections: exact
    assert np.array_equal(r[:, 3], exp[:, 3])
    assert np.array_equal(r[:, 6], exp[:, 6])
    np.testing.assert_allclose(r[:, :14], exp[:, :14], rtol=1e-6, atol=1e-6)
    if p['decode3d']:
        np.testing.assert_allclose(r[:, 14:], exp[:, 14:], rtol=1e-6, atol=1e-7)
    else:
        # 2-D: float64 sums differ from numpy's pairwise order by rounding
        np.testing.assert_allclose(r[:, :14], exp[:, :14], rtol=1e-12, atol=1e-12)
        # per-bit float32 means: same sequential order => bit-exact
        assert np.array_equal(r[:, 14:], exp[:, 14:])
    # empty index gives the reference's empty frame
    if len(exp) < N:
        empty = [f for f in frames if len(f) == 0][0]
        assert list(empty.columns)[:3] == ['barcode_id', 'fov', 'mean_intensity']
        assert len(empty.columns) == 14 + B


@pytest.mark.parametrize('name', [n for n in golden_names()
                                  if 'decode3d' not in n])
def test_extract_refactors_golden(name, gpu_device):
    g = load_golden(name)
    dec = _decoder(g, gpu_device)
    rf, bgr, seen = dec.extract_refactors(
        g['decoded'], g['magnitudes'], g['traces'], False)
    assert np.array_equal(seen, g['barcodes_seen'])
    np.testing.assert_allclose(rf, g['refactors'], rtol=1e-12, equal_nan=True)
    assert np.array_equal(bgr, np.zeros(g['codebook'].shape[1]))
    if 'refactors_bg' in g:
        rf, bgr, seen = dec.extract_refactors(
            g['decoded'], g['magnitudes'], g['traces'], True)
        np.testing.assert_allclose(rf, g['refactors_bg'], rtol=1e-12,
                                   equal_nan=True)
        np.testing.assert_allclose(bgr, g['background_refactors'],
                                   rtol=1e-12, equal_nan=True)


def test_cached_device_path_equals_upload_path(gpu_device):
    """decode_pixels -> extract (device-cached) == extract on re-uploaded
    copies of the same arrays."""
    g = load_golden('g2_synth192')
    p = g['p']
    dec = _decoder(g, gpu_device)
    di, pm, npt, d = _decode_gpu(dec, g)
    a = dec.extract_barcodes_with_index(5, di, pm, npt, d, 0, 0, 0)
    dec2 = _decoder(g, gpu_device)
    b = dec2.extract_barcodes_with_index(5, di.copy(), pm.copy(), npt.copy(),
                                         d.copy(), 0, 0, 0)
    assert a.equals(b)


@pytest.mark.parametrize('seed', range(6))
def test_decode_random_scale_factors(seed, gpu_device):
    """Bit-exactness of the reciprocal-based division path (and of the plain
    IEEE path for extreme scale factors) over awkward scale/background values,
    odd image sizes (edge tiles, unaligned rows) and every blur radius."""
    from conftest import synth_stack
    from merlin_amd.decoding import PixelBasedDecoder
    rng = np.random.default_rng(100 + seed)
    cb = load_golden('g2_synth192')['codebook']
    shape = [(70, 130), (64, 64), (33, 77), (128, 66), (17, 200), (96, 96)][seed]
    img = synth_stack(cb, shape, 25, seed=seed, amp=(500, 60000), bg=50)
    sf = np.exp(rng.uniform(np.log(0.3), np.log(3e4), 16))
    bg = rng.uniform(-20, 120, 16)
    sigma = [1, 0.5, 1.5, 2, 0, 1][seed]
    if seed == 5:
        sf[3] = 1e25          # disables the reciprocal path for the launch
    dm = oracle.normalized_codebook(cb)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    di, pm, npt, d = dec.decode_pixels(img, sf, bg, 0.5176, 1, sigma)
    odi, opm, onpt, od, gap = oracle.decode_pixels(img, dm, sf, bg, 0.5176, 1,
                                                   sigma, return_gap=True)
    assert np.array_equal(pm, opm)
    assert np.array_equal(npt, onpt)
    assert np.array_equal(di, odi)
    assert_distances(d, od, gap, odi)


@pytest.mark.parametrize('sigma', [0.1, 0.15, 0.2, 0.21, 0.25, 0.6])
def test_narrow_gaussian_u16(sigma, gpu_device):
    """lowPassSigma below ~0.21 gives ksize 3 with side taps that round to 0
    and a centre tap of exactly 65536 in 16.16 fixed point (the identity);
    just above it the side taps are tiny.  Both must match the oracle's exact
    integer blur on uint16 input (the packed 16-bit taps of the row pass cannot
    hold 65536)."""
    from conftest import synth_stack
    from merlin_amd.decoding import PixelBasedDecoder
    cb = load_golden('g2_synth192')['codebook']
    img = synth_stack(cb, (70, 130), 25, seed=3, amp=(500, 60000), bg=50)
    img[0, :4, :4] = 65535            # saturated pixels at a border
    sf = np.full(16, 900.0)
    bg = np.zeros(16)
    dm = oracle.normalized_codebook(cb)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    di, pm, npt, d = dec.decode_pixels(img, sf, bg, 0.5176, 1, sigma)
    odi, opm, onpt, od, gap = oracle.decode_pixels(img, dm, sf, bg, 0.5176, 1,
                                                   sigma, return_gap=True)
    assert np.array_equal(pm, opm)
    assert np.array_equal(npt, onpt)
    assert np.array_equal(di, odi)
    assert_distances(d, od, gap, odi)


@pytest.mark.parametrize('case', ['b22_mhd4', 'b22_n1000_w4', 'b32_n3000_w4',
                                  'b32_n1500_mixed', 'b12_n1'])
def test_large_and_odd_codebooks(case, gpu_device):
    """BASELINE configs 3 and 5 at oracle-sized images: 22-bit codebooks (MHD4,
    and 1000 distinct weight-4 words: two LDS chunks), 32-bit codebooks that exceed the LDS
    budget and stream through it in chunks (uniform weight and mixed weight),
    and the degenerate one-codeword codebook."""
    from conftest import synth_stack
    from merlin_amd.codebook import (constant_weight_codebook,
                                     random_weight_codebook)
    from merlin_amd.decoding import PixelBasedDecoder
    rng = np.random.default_rng(11)
    if case == 'b22_mhd4':
        # a 22-bit weight-4 code with Hamming distance 4 has at most
        # C(22,3)/4 = 385 words; the greedy construction finds 263
        cb = constant_weight_codebook(22, 4, 4).get_barcodes()
        assert cb.shape[0] >= 200
    elif case == 'b22_n1000_w4':
        cb = random_weight_codebook(22, 1000, 4, seed=7).get_barcodes()
    elif case == 'b32_n3000_w4':
        cb = random_weight_codebook(32, 3000, 4, seed=11).get_barcodes()
    elif case == 'b32_n1500_mixed':
        cb = np.zeros((1500, 32), dtype=np.int64)
        for i in range(1500):
            cb[i, rng.choice(32, rng.integers(2, 7), replace=False)] = 1
        cb = np.unique(cb, axis=0)
    else:
        cb = np.zeros((1, 12), dtype=np.int64)
        cb[0, [1, 4, 7, 9]] = 1
    N, B = cb.shape
    img = synth_stack(cb, (48, 80), 40, seed=7)
    sf = rng.uniform(200, 600, B)
    bg = rng.uniform(50, 120, B)
    dm = oracle.normalized_codebook(cb)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    thr = 0.5176 if case != 'b32_n1500_mixed' else 0.45
    di, pm, npt, d = dec.decode_pixels(img, sf, bg, thr, 1, 1)
    odi, opm, onpt, od, gap = oracle.decode_pixels(img, dm, sf, bg, thr, 1, 1,
                                                   return_gap=True)
    assert np.array_equal(pm, opm)
    assert np.array_equal(npt, onpt)
    assert np.array_equal(di, odi)
    assert (odi >= 0).sum() > 50
    assert_distances(d, od, gap, odi)
    # the region table of the decoded image, against the oracle
    i = int(np.bincount(odi[odi >= 0]).argmax())
    got = dec.extract_barcodes_with_index(i, di, pm, npt, d, 0, 0, 0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        exp = oracle.extract_barcodes_with_index(i, odi, opm, onpt, od, 0, 0, 0)
    r, e = frame_to_array(got, B), frame_to_array(exp, B)
    assert r.shape == e.shape
    np.testing.assert_allclose(r[:, :14], e[:, :14], rtol=1e-12, atol=1e-12)
    assert np.array_equal(r[:, 14:], e[:, 14:])


@pytest.mark.parametrize('sigma', [2.5, 3])
@pytest.mark.parametrize('bits', [16, 22])
def test_wide_gaussian_u16(sigma, bits, gpu_device):
    """lowPassSigma 2.5 and 3: ksize 11 and 13 (decoding.py:94), blur radius 5
    and 6 -- the widest the library builds (MF_MAX_RADIUS); odd image size, so
    edge tiles and reflected borders on every side, 16-bit and wider
    codebooks."""
    from conftest import synth_stack
    from merlin_amd.decoding import PixelBasedDecoder
    if bits == 16:
        cb = load_golden('g2_synth192')['codebook']
    else:
        from merlin_amd.codebook import random_weight_codebook
        cb = random_weight_codebook(bits, 300, 4, seed=2).get_barcodes()
    B = cb.shape[1]
    rng = np.random.default_rng(int(sigma * 10) + bits)
    img = synth_stack(cb, (75, 141), 30, seed=5, amp=(500, 60000), bg=50)
    sf = rng.uniform(200, 600, B)
    bg = rng.uniform(0, 60, B)
    dm = oracle.normalized_codebook(cb)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    di, pm, npt, d = dec.decode_pixels(img, sf, bg, 0.5176, 1, sigma)
    odi, opm, onpt, od, gap = oracle.decode_pixels(img, dm, sf, bg, 0.5176, 1,
                                                   sigma, return_gap=True)
    assert np.array_equal(pm, opm)
    assert np.array_equal(npt, onpt)
    assert np.array_equal(di, odi)
    assert_distances(d, od, gap, odi)


@pytest.mark.parametrize('sigma', [3.5, 5, 8])
@pytest.mark.parametrize('dtype', ['u16', 'f32'])
def test_sigma_beyond_the_fused_kernel(sigma, dtype, gpu_device):
    """lowPassSigma > 3 (ksize 15, 21, 33): the stack is blurred by
    mf_lowpass_images and decoded with sigma = 0 -- the same results as the
    oracle's blur + decode, uint16 (exact fixed point) and float input."""
    from conftest import synth_stack
    from merlin_amd.decoding import PixelBasedDecoder
    cb = load_golden('g2_synth192')['codebook']
    rng = np.random.default_rng(int(sigma * 10))
    img = synth_stack(cb, (61, 83), 12, seed=7, amp=(2000, 60000), bg=50)
    ref = img
    if dtype == 'f32':
        img = img.astype(np.float64) + rng.uniform(0, 1, img.shape)
        ref = img.astype(np.float32)
    sf = rng.uniform(20, 60, 16)
    bg = rng.uniform(0, 30, 16)
    dm = oracle.normalized_codebook(cb)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    di, pm, npt, d = dec.decode_pixels(img, sf, bg, 0.65, 1, sigma)
    odi, opm, onpt, od, gap = oracle.decode_pixels(ref, dm, sf, bg, 0.65, 1,
                                                   sigma, return_gap=True)
    assert np.array_equal(pm, opm)
    assert np.array_equal(npt, onpt)
    assert np.array_equal(di, odi)
    assert_distances(d, od, gap, odi)


def test_lowpass_images_arguments(gpu_device):
    """The C-ABI refuses what it cannot do: an image smaller than the blur
    radius (numpy's reflect padding refuses it too), sigma beyond ksize 65,
    a short workspace; the fused entry point names the way out."""
    import torch
    from merlin_amd import _lib
    L = _lib.lib()
    img = torch.zeros((2, 8, 8), dtype=torch.uint16, device=gpu_device)
    out = torch.empty_like(img)
    ws = torch.empty(2 * 8 * 8 * 4, dtype=torch.uint8, device=gpu_device)
    args = (img.data_ptr(), _lib.MF_DTYPE_U16, 2, 8, 8, 64)
    assert L.mf_lowpass_images(*args, 5.0, out.data_ptr(), ws.data_ptr(), ws.numel(), None) == -1
    assert b'smaller than blur radius' in L.mf_last_error_string()
    assert L.mf_lowpass_images(*args, 17.0, out.data_ptr(), ws.data_ptr(), ws.numel(), None) == -2
    assert L.mf_lowpass_images(*args, 1.0, out.data_ptr(), ws.data_ptr(), 16, None) < 0
    assert L.mf_lowpass_images(*args, 1.0, out.data_ptr(), ws.data_ptr(), ws.numel(), None) == 0
    from merlin_amd.decoding import PixelBasedDecoder
    cb = load_golden('g2_synth192')['codebook']
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    with pytest.raises(NotImplementedError, match='lowPassSigma'):
        dec.decode_pixels(np.zeros((16, 80, 80), dtype=np.uint16), np.ones(16),
                          np.zeros(16), 0.5176, 1, 17)


@pytest.mark.parametrize('sigma', [0, 0.5, 1, 1.5, 2, 2.5, 3])
def test_float_input_path(sigma, gpu_device):
    """Non-uint16 input is blurred in float32 (cv2's float path; parity with
    cv2 itself is unpinned, the oracle fixes the operation order).  Also runs
    every blur radius through an odd-sized image."""
    from conftest import synth_stack
    from merlin_amd.decoding import PixelBasedDecoder
    cb = load_golden('g2_synth192')['codebook']
    rng = np.random.default_rng(int(sigma * 10))
    img = synth_stack(cb, (75, 141), 30, seed=3).astype(np.float64)
    img += rng.uniform(0, 1, img.shape)            # genuinely fractional
    sf = rng.uniform(200, 600, 16)
    bg = rng.uniform(50, 120, 16)
    dm = oracle.normalized_codebook(cb)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    di, pm, npt, d = dec.decode_pixels(img, sf, bg, 0.5176, 1, sigma)
    odi, opm, onpt, od, gap = oracle.decode_pixels(
        img.astype(np.float32), dm, sf, bg, 0.5176, 1, sigma, return_gap=True)
    assert np.array_equal(pm, opm)
    assert np.array_equal(npt, onpt)
    assert np.array_equal(di, odi)
    assert_distances(d, od, gap, odi)


def test_argument_errors(gpu_device):
    from merlin_amd.decoding import PixelBasedDecoder
    cb = load_golden('g2_synth192')['codebook']
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    img = np.zeros((16, 32, 32), dtype=np.uint16)
    with pytest.raises(ValueError):
        dec.decode_pixels(img[:15])                      # wrong bit count
    with pytest.raises(ValueError):
        dec.decode_pixels(img, scaleFactors=np.ones(3))
    with pytest.raises(NotImplementedError):
        dec.decode_pixels(img, lowPassSigma=17)          # ksize 69 > 65
    with pytest.raises(ValueError):
        dec.decode_pixels(np.zeros((16, 2, 2), dtype=np.uint16))
    bad = cb.astype(np.float64)
    bad[0, 0] = 2                                        # not binary
    with pytest.raises(NotImplementedError):
        PixelBasedDecoder(ArrayCodebook(bad), device=gpu_device).decode_pixels(img)
    with pytest.raises(NotImplementedError):
        PixelBasedDecoder(ArrayCodebook(np.ones((3, 40), dtype=int)))


def test_exclusive_codebook_path_matches_runner_up_path(gpu_device):
    """MHD4 codebook + threshold <= 0.5176: the kernel drops the runner-up
    (no two codewords can both be within reach, decode.hip codebook_exclusive).
    Same decoded image and magnitudes as the top-2 path; distances identical
    wherever a pixel decodes and within the documented 4e-6 elsewhere."""
    import torch
    from merlin_amd import _lib
    from merlin_amd.decoding import PixelBasedDecoder
    cb = load_golden('g2_synth192')['codebook']
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    dec._codebook_device()
    flags = dec._cb_flags
    assert flags & 1 and (flags >> 8) & 0xff == 4 and (flags >> 16) & 0xff == 4
    img = torch.from_numpy(synth_stack(cb, (2, 384, 512), 900, seed=77)).to(gpu_device)
    rng = np.random.default_rng(1)
    sf = rng.uniform(300, 600, 16)
    bg = np.zeros(16)
    for thr in (0.5176, 0.5167, 0.3):
        a = dec.decode_pixels_device(img, sf, bg, thr, 1, 1)
        b = dec.decode_pixels_device(img, sf, bg, thr, 1, 1,
                                     keepRunnerUp=True)
        da, db = a.decoded.cpu().numpy(), b.decoded.cpu().numpy()
        assert np.array_equal(da, db) and (da >= 0).sum() > 1000
        assert torch.equal(a.magnitudes, b.magnitudes)
        xa, xb = a.distances.cpu().numpy(), b.distances.cpu().numpy()
        assert np.array_equal(xa[da >= 0], xb[da >= 0])
        np.testing.assert_allclose(xa, xb, rtol=4e-6, atol=0)


@pytest.mark.parametrize('spots', [0, 3])
def test_extract_refactors_with_few_or_no_regions(spots, gpu_device):
    """A plane in which nothing (or nearly nothing) decodes: barcodes that were
    never seen leave 0/0 = NaN exactly where the reference's loop does
    (decoding.py:329-340), with and without the background term."""
    g = load_golden('g2_synth192')
    cb = g['codebook']
    img = synth_stack(cb, (1, 64, 96), spots, seed=77)[:, 0]
    if spots == 0:
        img[:] = 0
    sf, bg = np.full(16, 300.0), np.zeros(16)
    dm = oracle.normalized_codebook(cb)
    di, pm, npt, d = oracle.decode_pixels(img, dm, sf, bg, 0.5176, 1, 1)
    from merlin_amd.decoding import PixelBasedDecoder
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    gdi, gpm, gnpt, gd = dec.decode_pixels(img, sf, bg)
    assert np.array_equal(gdi, di) and np.array_equal(gpm, pm)
    if spots == 0:
        assert (gdi == -1).all()
    for with_bg in (False, True):
        with np.errstate(all='ignore'):
            want = oracle.extract_refactors(di, pm, npt, dm, 4, with_bg)
            got = dec.extract_refactors(gdi, gpm, gnpt, with_bg)
        for a, b in zip(got, want):
            np.testing.assert_allclose(a, b, rtol=1e-12, equal_nan=True)
            assert np.array_equal(np.isnan(a), np.isnan(b))
