"""GPU parity of the preprocess stage (csrc/preprocess.hip through the C-ABI)
against oracle/preprocess_oracle.py and the reference-made goldens.

Bars: the uint16 high-pass filter is bit-exact.  The float64 deconvolution is
bit-exact while the acceleration parameter is still 0 (the first two
iterations: same operations in the same order); from the third iteration on
lambda = sum(T1 T2) / sum(T2 T2) is reduced in a fixed tile order instead of
numpy's pairwise order, so float64 results agree to 1e-9 relative and the
uint16 hand-over may differ by one count on <= 1e-5 of the pixels."""
import numpy as np
import pytest

from test_preprocess_oracle import PP_NAMES, load_pp
import preprocess_oracle as po

pytestmark = pytest.mark.gpu


def camera(shape, n_spots, seed, offset=2000):
    rng = np.random.default_rng(seed)
    h, w = shape
    img = offset + rng.poisson(100, size=shape).astype(np.float64)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_spots):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        img += rng.uniform(500, 6000) * np.exp(
            -((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 1.4 ** 2))
    return np.clip(np.rint(img), 0, 65535).astype(np.uint16)


def check_f64(got, want, rtol=1e-9):
    scale = np.maximum(np.abs(want), 1.0)
    err = np.abs(got - want) / scale
    assert err.max() <= rtol, 'max relative error {:.3g}'.format(err.max())


def check_u16(got, want, frac=1e-5):
    diff = got.astype(np.int64) - want.astype(np.int64)
    bad = np.count_nonzero(diff)
    assert bad <= max(1, frac * got.size) and (bad == 0 or np.abs(diff).max() == 1), \
        '{} of {} pixels differ, max |diff| {}'.format(bad, got.size, np.abs(diff).max())


@pytest.mark.parametrize('name', PP_NAMES)
def test_high_pass_golden(gpu_device, name):
    from merlin_amd import preprocess as pp
    g = load_pp(name)
    s = g['p']['hp_sigma']
    hp = pp.high_pass_filter(g['image'], po.filter_size_for_sigma(s), s)
    assert hp.dtype == np.uint16 and np.array_equal(hp, g['highpass'])


@pytest.mark.parametrize('shape,window,sigma', [
    ((1, 1), 13, 3.0), ((5, 3), 13, 3.0), ((33, 65), 13, 3.0), ((70, 130), 5, 1.0),
    ((64, 64), 17, 4.0), ((40, 200), 1, 0.0), ((97, 31), 9, 3.0)])
def test_high_pass_shapes(gpu_device, shape, window, sigma):
    from merlin_amd import preprocess as pp
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    img = rng.integers(0, 65536, size=shape, dtype=np.uint16)
    assert np.array_equal(pp.high_pass_filter(img, window, sigma),
                          po.high_pass_filter(img, window, sigma))


def test_high_pass_batch_of_images(gpu_device):
    import torch
    from merlin_amd import preprocess as pp
    stack = np.stack([camera((50, 90), 10, s) for s in range(6)]).reshape(3, 2, 50, 90)
    got = pp.high_pass_filter_device(torch.from_numpy(stack).to(gpu_device), 13, 3).cpu().numpy()
    for b in range(3):
        for z in range(2):
            assert np.array_equal(got[b, z], po.high_pass_filter(stack[b, z], 13, 3))


@pytest.mark.parametrize('name', PP_NAMES)
def test_lucy_richardson_golden(gpu_device, name):
    import torch
    from merlin_amd import preprocess as pp
    g = load_pp(name)
    p = g['p']
    hp = torch.from_numpy(g['highpass']).to(gpu_device)
    dec, lam = pp.deconvolve_lucyrichardson_device(
        hp, p['decon_size'], p['decon_sigma'], p['iterations'], return_lambdas=True)
    trace = []
    want = po.deconvolve_lucyrichardson(g['highpass'].astype(np.float64), p['decon_size'],
                                        p['decon_sigma'], p['iterations'], trace)
    assert np.array_equal(want, g['deconvolved'])
    np.testing.assert_allclose(lam.cpu().numpy()[0], np.array(trace), rtol=1e-11, atol=1e-14)
    check_f64(dec.cpu().numpy(), g['deconvolved'])
    out16 = pp.deconvolve_lucyrichardson_device(
        hp, p['decon_size'], p['decon_sigma'], p['iterations'], out_dtype=torch.uint16)
    check_u16(out16.cpu().numpy(), g['output'])
    # the numpy drop-in takes the reference's float64 image
    dec2 = pp.deconvolve_lucyrichardson(g['highpass'].astype(np.float64), p['decon_size'],
                                        p['decon_sigma'], p['iterations'])
    assert np.array_equal(dec2, dec.cpu().numpy())


@pytest.mark.parametrize('iterations', [0, 1, 2])
@pytest.mark.parametrize('window,sigma', [(9, 2.0), (5, 1.0), (13, 3.0), (1, 0.5), (3, 0.8)])
def test_unaccelerated_iterations_bit_exact(gpu_device, iterations, window, sigma):
    from merlin_amd import preprocess as pp
    img = po.high_pass_filter(camera((75, 141), 30, 5), 13, 3).astype(np.float64)
    got = pp.deconvolve_lucyrichardson(img, window, sigma, iterations)
    want = po.deconvolve_lucyrichardson(img, window, sigma, iterations)
    assert np.array_equal(got, want)


def test_float_input_with_negatives_and_zeros(gpu_device):
    from merlin_amd import preprocess as pp
    rng = np.random.default_rng(3)
    img = rng.normal(50, 40, size=(64, 100))
    img[10:20, 10:20] = 0.0
    got = pp.deconvolve_lucyrichardson(img, 9, 2, 2)
    assert np.array_equal(got, po.deconvolve_lucyrichardson(img, 9, 2, 2))
    check_f64(pp.deconvolve_lucyrichardson(img, 9, 2, 8),
              po.deconvolve_lucyrichardson(img, 9, 2, 8))
    zero = np.zeros((40, 70))
    assert np.array_equal(pp.deconvolve_lucyrichardson(zero, 9, 2, 20), zero)


@pytest.mark.parametrize('name', PP_NAMES)
def test_preprocess_image_golden(gpu_device, name):
    from merlin_amd import preprocess as pp
    g = load_pp(name)
    p = g['p']
    task = pp.DeconvolutionPreprocess(dict(
        highpass_sigma=p['hp_sigma'], decon_sigma=p['decon_sigma'],
        decon_filter_size=p['decon_size'], decon_iterations=p['iterations']))
    out = task._preprocess_image(g['image'])
    assert out.dtype == np.uint16
    check_u16(out, g['output'])
    hp = task._high_pass_filter(g['image'])
    assert hp.dtype == np.float64 and np.array_equal(hp, g['highpass'].astype(np.float64))


def test_stack_histogram_batches_and_determinism(gpu_device):
    import torch
    from merlin_amd import preprocess as pp
    B, Z, H, W = 3, 2, 70, 100
    stack = np.stack([camera((H, W), 15, 10 + s) for s in range(B * Z)]).reshape(B, Z, H, W)
    stack[0, 0, :4, :4] = 65535                 # saturated corner: wraps past 65535
    dev_stack = torch.from_numpy(stack).to(gpu_device)
    outs = []
    for batch in (1, 4, 6, None):
        out, hist = pp.DeconvolutionPreprocess(batch=batch).preprocess_device(dev_stack)
        out, hist = out.cpu().numpy(), hist.cpu().numpy()
        assert out.shape == stack.shape and hist.shape == (B, 65534)
        for b in range(B):      # histogram of the GPU's own output, exactly
            assert np.array_equal(hist[b], po.pixel_histogram(out[b]).astype(np.int64))
        outs.append(out)
    for o in outs[1:]:          # batching and repetition change nothing
        assert np.array_equal(o, outs[0])
    for b in range(B):
        for z in range(Z):
            check_u16(outs[0][b, z], po.preprocess_image(stack[b, z]))


def test_errors(gpu_device):
    import torch
    from merlin_amd import preprocess as pp
    img = np.zeros((32, 32))
    with pytest.raises(NotImplementedError):
        pp.deconvolve_lucyrichardson(img, 18, 2, 3)          # ndimage branch: <= 16
    with pytest.raises(ValueError):
        pp.deconvolve_lucyrichardson(img, 8, 2, 3, gaussianFilter=np.ones((6, 6)))
    with pytest.raises(ValueError):
        pp.deconvolve_lucyrichardson(img, 9, 2, 3, gaussianFilter=np.ones((9, 9)))
    with pytest.raises(NotImplementedError):
        pp.deconvolve_lucyrichardson(img, 15, 2, 3)
    with pytest.raises(NotImplementedError):
        pp.high_pass_filter(img.astype(np.uint16), 19, 4.5)
    with pytest.raises(NotImplementedError):
        pp.high_pass_filter(img.astype(np.float32), 13, 3)
    with pytest.raises(ValueError):
        pp.deconvolve_lucyrichardson(img, 9, 2, -1)
    with pytest.raises(ValueError):
        pp.deconvolve_lucyrichardson_device(
            torch.zeros((2, 8, 8), dtype=torch.uint16, device=gpu_device), 9, 2, 2,
            out_dtype=torch.uint16,
            histogram=torch.zeros((1, 10), dtype=torch.int64, device=gpu_device))


def test_full_size_image(gpu_device):
    """2048 x 2048 (BASELINE config sizes): one image against the oracle, and
    independence from the images it is batched with."""
    import torch
    from merlin_amd import preprocess as pp
    rng = np.random.default_rng(77)
    H = W = 2048
    img = (2000 + rng.poisson(100, size=(H, W))).astype(np.float64)
    ys, xs = rng.integers(8, H - 8, 4000), rng.integers(8, W - 8, 4000)
    yy, xx = np.mgrid[-6:7, -6:7]
    psf = np.exp(-(yy ** 2 + xx ** 2) / (2 * 1.4 ** 2))
    for y, x, a in zip(ys, xs, rng.uniform(500, 6000, 4000)):
        img[y - 6:y + 7, x - 6:x + 7] += a * psf
    img = np.clip(np.rint(img), 0, 65535).astype(np.uint16)
    want = po.preprocess_image(img)
    pair = torch.from_numpy(np.stack([img, img[::-1].copy()])).to(gpu_device)
    out, hist = pp.DeconvolutionPreprocess(batch=2).preprocess_device(pair, hist_group=1)
    out = out.cpu().numpy()
    check_u16(out[0], want)
    assert np.array_equal(hist[0].cpu().numpy(), po.pixel_histogram(out[0]).astype(np.int64))
    alone, _ = pp.DeconvolutionPreprocess(batch=1).preprocess_device(pair[:1], histogram=False)
    assert np.array_equal(alone.cpu().numpy()[0], out[0])


def test_preprocess_feeds_decode(gpu_device):
    """Camera frames -> preprocess -> initial scale factors from the pixel
    histogram (optimize.py:140-155) -> Decode.run_fov, against the same chain
    through the two oracles."""
    import warnings
    import pandas
    import torch
    import merfish_oracle as oracle
    from conftest import ArrayCodebook, frame_to_array, load_golden, synth_stack
    from merlin_amd import preprocess as pp
    from merlin_amd.decode import Decode
    from merlin_amd.optimize import initial_scale_factors
    cb = load_golden('g2_synth192')['codebook']
    B, Z, H, W = 16, 2, 96, 128
    raw = synth_stack(cb, (Z, H, W), 70, seed=8, amp=(800, 4000), psf=1.6)
    raw = (raw.astype(np.int64) + 1900).clip(0, 65535).astype(np.uint16)     # camera offset
    stack, hist = pp.DeconvolutionPreprocess().preprocess_device(
        torch.from_numpy(raw).to(gpu_device))
    want = np.array([[po.preprocess_image(raw[b, z]) for z in range(Z)] for b in range(B)])
    got = stack.cpu().numpy()
    check_u16(got, want)
    want_hist = np.array([po.pixel_histogram(want[b]) for b in range(B)])
    if np.array_equal(got, want):
        assert np.array_equal(hist.cpu().numpy(), want_hist.astype(np.int64))
    sf = initial_scale_factors(hist.cpu().numpy().astype(np.float64))
    assert np.array_equal(sf, initial_scale_factors(want_hist))
    bg = np.zeros(B)
    params = {'crop_width': 4, 'distance_threshold': 0.5167, 'lowpass_sigma': 1}
    task = Decode(ArrayCodebook(cb), params, device=gpu_device)
    bc, vol = task.run_fov(stack, sf, bg, fov=0)
    dm = oracle.normalized_codebook(cb)
    frames = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for z in range(Z):
            di, pm, npt, d = oracle.decode_pixels(got[:, z], dm, sf, bg, 0.5167, 1, 1)
            assert np.array_equal(vol.decoded[z].cpu().numpy(), di)
            frames.append(oracle.extract_all_barcodes(cb.shape[0], di, pm, npt, d, 0, 4, z, None, 0))
    exp = pandas.concat(frames)
    r, e = frame_to_array(bc, B), frame_to_array(exp, B)
    assert r.shape == e.shape and len(r) > 10
    np.testing.assert_allclose(r, e, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('shape', [(1, 1), (3, 5), (1, 200), (200, 1), (9, 9), (130, 47), (257, 64)])
def test_lucy_richardson_small_and_odd_shapes(gpu_device, shape):
    """Strips narrower / shorter than a wave's 64 x 128 window, images smaller
    than the blur radius: every border path of the fused kernel."""
    from merlin_amd import preprocess as pp
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    img = rng.integers(0, 3000, size=shape).astype(np.float64)
    img[rng.random(shape) < 0.5] = 0.0
    for window, sigma in ((9, 2.0), (13, 3.0)):
        assert np.array_equal(pp.deconvolve_lucyrichardson(img, window, sigma, 2),
                              po.deconvolve_lucyrichardson(img, window, sigma, 2))
        check_f64(pp.deconvolve_lucyrichardson(img, window, sigma, 9),
                  po.deconvolve_lucyrichardson(img, window, sigma, 9))


def test_guo_variant(gpu_device):
    """deconvolve_lucyrichardson_guo / DeconvolutionPreprocessGuo against the
    reference-made goldens and the oracle.  The projectors come out of an FFT
    (different builds differ in the last bits), so the comparison with the
    golden is to 1e-9; with the oracle fed the same projectors the non-separable
    filter is the same sum in the same order: bit-exact."""
    import glob, os
    import torch
    from conftest import GOLDEN_DIR
    from merlin_amd import preprocess as pp
    names = sorted(os.path.basename(p)[:-4]
                   for p in glob.glob(os.path.join(GOLDEN_DIR, 'ppguo_*.npz')))
    assert len(names) == 3
    for name in names:
        g = load_pp(name)
        p = g['p']
        hp = g['highpass'].astype(np.float64)
        got = pp.deconvolve_lucyrichardson_guo(hp, p['decon_size'], p['decon_sigma'],
                                               p['iterations'])
        check_f64(got, g['deconvolved'])
        assert np.array_equal(got, po.deconvolve_lucyrichardson_guo(
            hp, p['decon_size'], p['decon_sigma'], p['iterations']))
    g = load_pp('ppguo_synth')
    task = pp.DeconvolutionPreprocessGuo()
    assert task.parameters['decon_iterations'] == 2 and task.parameters['decon_filter_size'] == 9
    stack = np.stack([g['image'], g['image'][::-1].copy()])[None]           # [1, 2, H, W]
    out, hist = task.preprocess_device(torch.from_numpy(stack).to(gpu_device))
    out = out.cpu().numpy()
    check_u16(out[0, 0], g['output'])
    assert np.array_equal(hist[0].cpu().numpy(), po.pixel_histogram(out[0]).astype(np.int64))
    with pytest.raises(NotImplementedError):
        pp.deconvolve_lucyrichardson_guo(np.zeros((20, 20)), 15, 2, 2)
    with pytest.raises(NotImplementedError):
        pp.deconvolve_lucyrichardson_guo(np.zeros((20, 20)), 8, 2, 2)


@pytest.mark.parametrize('hist_group', [1, 3])
def test_fused_handover_histogram_counts_every_pixel(gpu_device, hist_group):
    """The last iteration writes the uint16 hand-over and the histogram itself
    (zeros per lane, values below 256 per wave in LDS, the rest by global
    atomics): on noisy frames with many non-zero and some large values the
    histogram must be exactly the histogram of the output, per row group, and
    the output must equal the unfused path's (float64 result cast on the host)."""
    import torch
    from merlin_amd import preprocess as pp
    rng = np.random.default_rng(21)
    frames = np.stack([camera((96, 200), 60, 70 + i) for i in range(6)])
    frames = np.clip(frames.astype(np.int64) + rng.integers(-40, 40, frames.shape), 0, 65535
                     ).astype(np.uint16)
    task = pp.DeconvolutionPreprocess({'decon_iterations': 6}, batch=4)
    out, hist = task.preprocess_device(torch.from_numpy(frames).to(gpu_device),
                                       hist_group=hist_group)
    out, hist = out.cpu().numpy(), hist.cpu().numpy()
    assert hist.shape == (6 // hist_group, 65534)
    for row in range(6 // hist_group):
        block = out[row * hist_group:(row + 1) * hist_group]
        assert np.array_equal(hist[row], po.pixel_histogram(block).astype(np.int64))
    assert (out > 0).mean() > 0.05 and (out >= 256).sum() > 100
    hp = pp.high_pass_filter_device(torch.from_numpy(frames).to(gpu_device), 13, 3)
    dec = pp.deconvolve_lucyrichardson_device(hp, 9, 2, 6).cpu().numpy()
    assert np.array_equal(out, po.to_uint16(dec))


def test_even_window_branch(gpu_device):
    """Even windowSize: the reference's scipy.ndimage.convolve branch
    (deconvolve.py:102-104, 118-119, 126-128) through
    mf_deconvolve_lucyrichardson_kernel2d.  Against the reference-made goldens
    with the kernel the reference used: bit-exact while the acceleration
    parameter is 0, then to 1e-9 (lambda's sums are added in tile order);
    against the oracle on other shapes, batches and both input dtypes."""
    import glob, os
    import torch
    from conftest import GOLDEN_DIR
    from merlin_amd import preprocess as pp
    names = sorted(os.path.basename(p)[:-4]
                   for p in glob.glob(os.path.join(GOLDEN_DIR, 'ppeven_*.npz')))
    assert len(names) == 3
    for name in names:
        g = load_pp(name)
        p = g['p']
        hp = g['highpass'].astype(np.float64)
        two = pp.deconvolve_lucyrichardson(hp, p['decon_size'], p['decon_sigma'], 2,
                                           gaussianFilter=g['kernel'])
        assert np.array_equal(two, po.deconvolve_lucyrichardson(
            hp, p['decon_size'], p['decon_sigma'], 2, gaussianFilter=g['kernel']))
        got = pp.deconvolve_lucyrichardson(hp, p['decon_size'], p['decon_sigma'],
                                           p['iterations'], gaussianFilter=g['kernel'])
        check_f64(got, g['deconvolved'])
        dec16, lam = pp.deconvolve_lucyrichardson_device(
            torch.from_numpy(g['highpass']).to(gpu_device), p['decon_size'], p['decon_sigma'],
            p['iterations'], out_dtype=torch.uint16, return_lambdas=True,
            gaussianFilter=g['kernel'])
        check_u16(dec16.cpu().numpy(), g['output'])
        trace = []
        po.deconvolve_lucyrichardson(hp, p['decon_size'], p['decon_sigma'], p['iterations'],
                                     trace, gaussianFilter=g['kernel'])
        np.testing.assert_allclose(lam.cpu().numpy()[0], np.array(trace), rtol=1e-11, atol=1e-14)
    # odd shapes, a batch of different images, this interpreter's own kernel
    rng = np.random.default_rng(8)
    for shape, window, sigma in (((75, 141), 6, 1.5), ((33, 70), 16, 3.0), ((5, 7), 8, 2.0),
                                 ((64, 200), 2, 0.6)):
        stack = np.stack([po.high_pass_filter(camera(shape, 12, 40 + i), 13, 3) for i in range(3)])
        got = pp.deconvolve_lucyrichardson_device(
            torch.from_numpy(stack).to(gpu_device), window, sigma, 2, batch=2).cpu().numpy()
        for i in range(3):
            assert np.array_equal(got[i], po.deconvolve_lucyrichardson(
                stack[i].astype(np.float64), window, sigma, 2))
        check_f64(pp.deconvolve_lucyrichardson(stack[0].astype(np.float64), window, sigma, 7),
                  po.deconvolve_lucyrichardson(stack[0].astype(np.float64), window, sigma, 7))
    # the preprocess task with an even decon_filter_size
    g = load_pp('ppeven_w8')
    task = pp.DeconvolutionPreprocess({'decon_filter_size': 8, 'decon_iterations': 8})
    out, hist = task.preprocess_device(torch.from_numpy(g['image'][None]).to(gpu_device))
    check_u16(out[0].cpu().numpy(), g['output'], frac=1e-3)
    assert np.array_equal(hist[0].cpu().numpy(),
                          po.pixel_histogram(out.cpu().numpy()[0]).astype(np.int64))


def test_release_workspaces(gpu_device):
    """The per-stream deconvolution workspaces are cached for reuse and can be
    handed back."""
    from merlin_amd import preprocess
    img = np.random.default_rng(0).integers(0, 4000, (2, 64, 64)).astype(np.uint16)
    a = preprocess.deconvolve_lucyrichardson(img[0], 9, 2, 3)
    assert preprocess.release_workspaces(gpu_device) > 0
    assert preprocess.release_workspaces() == 0
    b = preprocess.deconvolve_lucyrichardson(img[0], 9, 2, 3)
    assert np.array_equal(a, b)
