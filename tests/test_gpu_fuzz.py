"""GPU: a fixed sample of the randomised sweep tests/sweeps/fuzz_fov.py -- Decode.run_fov
(HIP path) against the oracle over random codebooks (8-32 bits), shapes, blur
widths, thresholds, crops, minimum areas, 2-D / 3-D labelling and spot
densities.  Bars as in test_gpu_fov.py: integer columns and record sets equal;
2-D scalars within 1e-12 and per-bit means bit-exact; 3-D floats within 1e-6.
The sweep itself (thousands of cases, profiles/r02/fuzz_fov.txt) is run by
hand; these 40 seeds (and 60 of the preprocess sweep tests/sweeps/fuzz_pp.py) ride with
every GPU run."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

_HERE = os.path.dirname(os.path.abspath(__file__))


def _sweep():
    spec = importlib.util.spec_from_file_location(
        'fuzz_fov', os.path.join(_HERE, 'sweeps', 'fuzz_fov.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('block', range(4))
def test_random_fovs_match_oracle(block, gpu_device):
    sweep = _sweep()
    records = 0
    for seed in range(3000 + 10 * block, 3010 + 10 * block):
        ok, why, desc, n = sweep.one_case(seed, gpu_device)
        assert ok, '%s: %s' % (desc, why)
        records += n
    assert records > 0


def _pp_sweep():
    spec = importlib.util.spec_from_file_location(
        'fuzz_pp', os.path.join(_HERE, 'sweeps', 'fuzz_pp.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('block', range(2))
def test_random_preprocess_cases_match_oracle(block, gpu_device):
    """60 seeds of tests/sweeps/fuzz_pp.py: high pass bit-exact, Lucy-Richardson
    bit-exact up to two iterations and within 1e-9 after, uint16 hand-over
    equal except where the oracle's float64 value is within that 1e-9 of an
    integer, histogram equal to the histogram of the output."""
    sweep = _pp_sweep()
    for seed in range(4000 + 30 * block, 4030 + 30 * block):
        ok, why, desc = sweep.one_case(seed, gpu_device)
        assert ok, '%s: %s' % (desc, why)


@pytest.mark.parametrize('block', range(2))
def test_random_refactor_cases_match_oracle(block, gpu_device):
    """40 seeds of tests/sweeps/fuzz_refactors.py: extract_refactors with and without
    the background term, 1e-12 relative, NaNs in the same places."""
    spec = importlib.util.spec_from_file_location(
        'fuzz_refactors', os.path.join(_HERE, 'sweeps', 'fuzz_refactors.py'))
    sweep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sweep)
    for seed in range(8000 + 20 * block, 8020 + 20 * block):
        ok, why, desc = sweep.one_case(seed, gpu_device)
        assert ok, '%s: %s' % (desc, why)


def test_random_label_volumes_match_scipy(gpu_device):
    """300 seeds of tests/sweeps/fuzz_ccl.py: mf_label_components against scipy's
    labelling of every code, both connectivities."""
    spec = importlib.util.spec_from_file_location(
        'fuzz_ccl', os.path.join(_HERE, 'sweeps', 'fuzz_ccl.py'))
    sweep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sweep)
    for seed in range(40000, 40300):
        ok, why, desc = sweep.one_case(seed, gpu_device)
        assert ok, '%s: %s' % (desc, why)


@pytest.mark.parametrize('block', range(2))
def test_random_decode_pixels_calls_match_oracle(block, gpu_device):
    """80 seeds of tests/sweeps/fuzz_decode.py: u16 / f32 / f64 images, uniform
    and mixed-weight codebooks, every blur radius; decoded image, magnitudes,
    traces bit-exact, distances bit-exact except at far tied pixels (4e-6)."""
    spec = importlib.util.spec_from_file_location(
        'fuzz_decode', os.path.join(_HERE, 'sweeps', 'fuzz_decode.py'))
    sweep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sweep)
    for seed in range(50000 + 40 * block, 50040 + 40 * block):
        ok, why, desc = sweep.one_case(seed, gpu_device)
        assert ok, '%s: %s' % (desc, why)
