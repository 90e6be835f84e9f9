"""GPU: the FOV runner (device-resident, all planes in one launch, device-side
record filter) against the per-index drop-in API and the oracle."""
import warnings

import numpy as np
import pandas
import pytest

import merfish_oracle as oracle
from conftest import (ArrayCodebook, AffineAligner, frame_to_array,
                      load_golden, synth_stack)

pytestmark = pytest.mark.gpu


def _oracle_fov(cb, img, sf, bg, params, fov, tform, decode3d):
    dm = oracle.normalized_codebook(cb)
    B, Z = img.shape[:2]
    outs = [oracle.decode_pixels(img[:, z], dm, sf, bg,
                                 params['distance_threshold'], 1,
                                 params['lowpass_sigma']) for z in range(Z)]
    frames = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if decode3d:
            di = np.array([o[0] for o in outs])
            pm = np.array([o[1] for o in outs])
            npt = np.array([o[2] for o in outs])
            d = np.array([o[3] for o in outs]).astype(np.float32)
            frames.append(oracle.extract_all_barcodes(
                cb.shape[0], di, pm, npt, d, fov, params['crop_width'], None,
                tform, params['minimum_area']))
        else:
            for z, (di, pm, npt, d) in enumerate(outs):
                frames.append(oracle.extract_all_barcodes(
                    cb.shape[0], di, pm, npt, d, fov, params['crop_width'], z,
                    tform, params['minimum_area']))
        return pandas.concat(frames)


@pytest.mark.parametrize('decode3d', [False, True])
def test_run_fov_matches_oracle(gpu_device, decode3d):
    import torch
    from merlin_amd.decode import Decode
    cb = load_golden('g2_synth192')['codebook']
    shape = (24, 40, 72) if decode3d else (3, 80, 136)
    img = synth_stack(cb, shape, 90, seed=21)
    rng = np.random.default_rng(3)
    sf = rng.uniform(300, 600, 16)
    bg = rng.uniform(90, 110, 16)
    tform = np.float32([[0.108, 0, 3.5], [0, 0.108, -2.25], [0, 0, 1]])
    params = {'crop_width': 4, 'minimum_area': 2, 'distance_threshold': 0.5167,
              'lowpass_sigma': 1, 'decode_3d': decode3d}
    task = Decode(ArrayCodebook(cb), params, device=gpu_device)
    bc, vol = task.run_fov(torch.from_numpy(img).to(gpu_device), sf, bg,
                           fov=5, globalAligner=AffineAligner(tform))
    exp = _oracle_fov(cb, img, sf, bg, task.parameters, 5, tform, decode3d)
    r, e = frame_to_array(bc, 16), frame_to_array(exp, 16)
    assert r.shape == e.shape and len(r) > 10
    for c in (0, 1, 4, 13):
        assert np.array_equal(r[:, c], e[:, c])
    np.testing.assert_allclose(r, e, rtol=1e-6, atol=1e-6)
    if not decode3d:
        np.testing.assert_allclose(r[:, :14], e[:, :14], rtol=1e-12, atol=1e-12)
        assert np.array_equal(r[:, 14:], e[:, 14:])
        assert np.array_equal(bc.index.to_numpy(), exp.index.to_numpy())
    assert list(bc.columns) == list(exp.columns)
    assert bc['barcode_id'].dtype == np.int64
    assert bc['intensity_0'].dtype == np.float32


def test_run_fov_matches_drop_in_api(gpu_device):
    """run_fov == Decode's own loop over planes and barcode indexes through
    decode_pixels / extract_barcodes_with_index (decode.py:118-127, 225-230)."""
    import torch
    from merlin_amd.decode import Decode
    from merlin_amd.decoding import PixelBasedDecoder
    cb = load_golden('g2_synth192')['codebook']
    img = synth_stack(cb, (2, 96, 128), 80, seed=4)
    sf = np.full(16, 400.0)
    bg = np.zeros(16)
    params = {'crop_width': 3}
    task = Decode(ArrayCodebook(cb), params, device=gpu_device)
    bc, vol = task.run_fov(torch.from_numpy(img).to(gpu_device), sf, bg, fov=1)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    frames = []
    for z in range(2):
        di, pm, npt, d = dec.decode_pixels(img[:, z], sf, bg,
                                           distanceThreshold=0.5167)
        assert np.array_equal(di, vol.decoded[z].cpu().numpy())
        assert np.array_equal(d, vol.distances[z].cpu().numpy())
        frames += [dec.extract_barcodes_with_index(i, di, pm, npt, d, 1, 3, z)
                   for i in range(cb.shape[0])]
    exp = pandas.concat([f for f in frames if len(f)])
    assert np.array_equal(frame_to_array(bc, 16), frame_to_array(exp, 16))
    assert np.array_equal(bc.index.to_numpy(), exp.index.to_numpy())


def test_run_fovs_pipeline_equals_run_fov(gpu_device):
    """The pipelined FOV loop (frame rows of FOV k cross PCIe on a side stream
    while FOV k+1 decodes) yields exactly run_fov's tables, in order,
    including an FOV with no barcode at all and z-duplicate removal."""
    import torch
    from merlin_amd.decode import Decode
    cb = load_golden('g2_synth192')['codebook']
    stacks = [synth_stack(cb, (3, 96, 128), 70, seed=30 + k) for k in range(4)]
    stacks[2] = np.zeros_like(stacks[2])            # nothing decodes
    dev_stacks = [torch.from_numpy(s).to(gpu_device) for s in stacks]
    sf = np.full(16, 380.0)
    bg = np.zeros(16)
    tform = np.float32([[0.1, 0, 7.0], [0, 0.1, -3.0], [0, 0, 1]])
    for extra in ({}, {'remove_z_duplicated_barcodes': True}):
        params = dict({'crop_width': 5, 'minimum_area': 1}, **extra)
        task = Decode(ArrayCodebook(cb), params, device=gpu_device)
        fovs = [11, 12, 13, 14]
        got = list(task.run_fovs(iter(dev_stacks), sf, bg, fovs=fovs,
                                 globalAligner=AffineAligner(tform)))
        assert len(got) == 4
        for k, (bc, vol) in enumerate(got):
            ebc, evol = task.run_fov(dev_stacks[k], sf, bg, fovs[k],
                                     globalAligner=AffineAligner(tform))
            assert torch.equal(vol.decoded, evol.decoded)
            assert list(bc.columns) == list(ebc.columns)
            assert np.array_equal(frame_to_array(bc, 16),
                                  frame_to_array(ebc, 16))
            assert np.array_equal(bc.index.to_numpy(), ebc.index.to_numpy())
            assert (len(bc) == 0) == (k == 2)
            if len(bc):
                assert (bc['fov'] == fovs[k]).all()


def test_optimize_iterations_match_oracle(gpu_device):
    """Two chained OptimizeIteration passes over the reference test-suite FOVs
    (test_analysis_parameters.json:17-41: fov_per_iteration 2) against the
    oracle's decode -> extract_refactors -> nanmedian loop."""
    import torch
    from merlin_amd import optimize
    g = np.load(__import__('os').path.join(
        __import__('conftest').GOLDEN_DIR, 'testsuite_inputs.npz'))
    images, cb = g['images'], g['codebook']
    dm = oracle.normalized_codebook(cb)
    dev_imgs = [torch.from_numpy(images[f]).to(gpu_device) for f in range(2)]
    hist = sum(optimize.pixel_histogram(d) for d in dev_imgs)
    init = optimize.initial_scale_factors(optimize.summed_pixel_histogram(hist))
    ref_hist = np.array([sum(np.histogram(images[f][b],
                                          bins=np.arange(0, 65535, 1))[0]
                             for f in range(2)) for b in range(16)])
    assert np.array_equal(init, oracle.initial_scale_factors(ref_hist))
    # the raw frames keep their camera offset: start from a background guess
    bg0 = np.array([np.median(images[:, b]) for b in range(16)])

    class Warm(object):          # "previous iteration" carrying the start
        def get_scale_factors(self): return init - bg0
        def get_backgrounds(self): return bg0
        def get_scale_factor_history(self): return np.array([init - bg0])
        def get_barcode_count_history(self): return np.zeros((1, 140))

    frags = [[0, 0], [1, 0]]
    prev = Warm()
    sf, bg = init - bg0, bg0
    for it in range(2):
        cur = optimize.OptimizeIteration(
            ArrayCodebook(cb), {'fov_index': frags, 'area_threshold': 3,
                                'optimize_background': True},
            previousIteration=prev,
            imageProvider=lambda f, z: dev_imgs[f], device=gpu_device).run()
        rfs, bgrs = [], []
        for f, z in frags:
            di, pm, npt, d = oracle.decode_pixels(images[f], dm, sf, bg)
            rf, bgr, seen = oracle.extract_refactors(di, pm, npt, dm, 3, True)
            rfs.append(rf)
            bgrs.append(bgr)
            # the fragment's barcode table (optimize.py:116-122): crop 0,
            # z = the fragment's z index, minimum area = area_threshold
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                exp = oracle.extract_all_barcodes(140, di, pm, npt, d, f, 0, z,
                                                  None, 3)
            got = cur.barcodes[frags.index([f, z])]
            assert list(got.columns) == list(exp.columns) and len(exp) > 5
            r, e = frame_to_array(got, 16), frame_to_array(exp, 16)
            assert r.shape == e.shape
            np.testing.assert_allclose(r[:, :14], e[:, :14], rtol=1e-12,
                                       atol=1e-12)
            assert np.array_equal(r[:, 14:], e[:, 14:])
            assert np.array_equal(got.index.to_numpy(), exp.index.to_numpy())
        prevf = np.tile(sf, (2, 1))
        esf = oracle.reduce_scale_factors(np.array(rfs), prevf)
        ebg = oracle.reduce_backgrounds(np.array(bgrs), np.tile(bg, (2, 1)),
                                        prevf)
        np.testing.assert_allclose(cur.get_scale_factors(), esf, rtol=1e-12,
                                   equal_nan=True)
        np.testing.assert_allclose(cur.get_backgrounds(), ebg, rtol=1e-12,
                                   atol=1e-12, equal_nan=True)
        assert set(cur.barcodes) == {0, 1}
        sf, bg, prev = esf, ebg, cur
    assert cur.get_scale_factor_history().shape == (3, 16)


def test_optimize_fragment_workers_do_not_change_results(gpu_device):
    """OptimizeIteration runs its fragments on worker threads (four in flight
    by default); one at a time gives the same refactors, counts and barcode
    tables, bit for bit."""
    import torch
    from merlin_amd import optimize
    from merlin_amd.synthetic import synthetic_fov
    rng = np.random.default_rng(3)
    cb = (rng.random((40, 16)) < 0).astype(int)
    for row in cb:
        row[rng.choice(16, 4, replace=False)] = 1
    cb = np.unique(cb, axis=0)
    fov = synthetic_fov(cb, 3, 256, 320, 300, 77, gpu_device)
    init = np.full(16, 400.0)
    frags = [[0, z] for z in (0, 1, 2, 1, 0, 2, 2)]
    runs = []
    for workers in (1, 3):
        it = optimize.OptimizeIteration(
            ArrayCodebook(cb), {'fov_index': frags, 'optimize_background': True,
                                'fragment_workers': workers},
            initialScaleFactors=init, imageProvider=lambda f, z: fov[:, z],
            device=gpu_device).run()
        runs.append(it)
    a, b = runs
    assert np.array_equal(a.get_scale_factors(), b.get_scale_factors(), equal_nan=True)
    assert np.array_equal(a.get_backgrounds(), b.get_backgrounds(), equal_nan=True)
    assert np.array_equal(a._counts, b._counts) and a._counts.sum() > 50
    assert np.array_equal(a._refactors, b._refactors, equal_nan=True)
    assert set(a.barcodes) == set(b.barcodes) == set(range(len(frags)))
    for k in a.barcodes:
        assert len(a.barcodes[k]) > 5
        assert a.barcodes[k].equals(b.barcodes[k])
    # the same plane drawn twice gives the same table
    assert a.barcodes[1].equals(a.barcodes[3])


def test_run_fovs_from_host_matches_run_fov(gpu_device):
    """Host-resident stacks through the pinned multi-buffered uploader (more
    FOVs than device buffers; pageable numpy, pinned and plain CPU tensors)
    give the tables run_fov gives, with and without the preprocess stage."""
    import torch
    from merlin_amd.decode import Decode
    from merlin_amd.preprocess import DeconvolutionPreprocess
    cb = load_golden('g2_synth192')['codebook']
    sf = np.full(16, 350.0)
    bg = np.zeros(16)
    task = Decode(ArrayCodebook(cb), {'crop_width': 3}, device=gpu_device)
    stacks = [synth_stack(cb, (2, 64, 96), 40, seed=30 + i) for i in range(7)]
    hosts = []
    for i, s in enumerate(stacks):
        if i % 3 == 0:
            hosts.append(s)
        elif i % 3 == 1:
            hosts.append(torch.from_numpy(s).pin_memory())
        else:
            hosts.append(torch.from_numpy(s))
    want = [task.run_fov(torch.from_numpy(s).to(gpu_device), sf, bg, fov=10 + i)[0]
            for i, s in enumerate(stacks)]
    got = [bc for bc, vol in task.run_fovs_from_host(hosts, sf, bg,
                                                     fovs=list(range(10, 17)), depth=2)]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert len(w) > 0 and g.equals(w)

    pre = DeconvolutionPreprocess(device=gpu_device)
    raws = [(s.astype(np.int64) + 1500).clip(0, 65535).astype(np.uint16) for s in stacks[:4]]
    sf2 = np.full(16, 60.0)
    want = [task.run_fov(pre.preprocess_device(torch.from_numpy(r).to(gpu_device),
                                               histogram=False)[0], sf2, bg, fov=i)[0]
            for i, r in enumerate(raws)]
    got = [bc for bc, vol in task.run_fovs_from_host(raws, sf2, bg, preprocess=pre)]
    assert sum(len(w) for w in want) > 0
    for g, w in zip(got, want):
        assert g.equals(w)


def test_side_streams_are_shared_between_decode_objects(gpu_device):
    """The loop's three side streams exist once per device and process (HIP maps
    streams onto four hardware queues: a second Decode object with streams of its
    own ran at 1.75 ms per FOV instead of 1.46), and a second object gives the
    same barcodes as the first."""
    from merlin_amd.decode import Decode
    from merlin_amd.synthetic import synthetic_fov
    cb = ArrayCodebook(load_golden('g2_synth192')['codebook'])
    a = Decode(cb, {'crop_width': 4}, device=gpu_device)
    b = Decode(cb, {'crop_width': 4}, device=gpu_device)
    for role in ('_side_stream', '_table_stream', '_label_stream'):
        assert getattr(a, role)(gpu_device) is getattr(b, role)(gpu_device)
    stacks = [synthetic_fov(cb.get_barcodes(), 2, 96, 128, 60, 50 + i, gpu_device)
              for i in range(3)]
    sf, bg = np.full(16, 400.0), np.zeros(16)
    ra = [bc for bc, _ in a.run_fovs(stacks, sf, bg)]
    rb = [bc for bc, _ in b.run_fovs(stacks, sf, bg)]
    assert sum(len(x) for x in ra) > 20
    for x, y in zip(ra, rb):
        assert x.equals(y)
