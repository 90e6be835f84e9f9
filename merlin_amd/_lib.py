"""ctypes binding of libmerfish_hip.so (the C-ABI in include/merfish_hip.h).

The library is the product path; there is no CPU fallback.  Importing this
module never touches the GPU; ``lib()`` loads the shared object and raises a
RuntimeError naming the build command when it is missing.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MERFISH_HIP_LIB: development override to time kernel variants
LIB_PATH = os.environ.get('MERFISH_HIP_LIB') or os.path.join(
    _HERE, 'libmerfish_hip.so')

MF_DTYPE_U16 = 0
MF_DTYPE_F32 = 1
MF_DTYPE_F64 = 2
MF_MAX_BITS = 32
MF_MAX_RADIUS = 6
MF_LOWPASS_MAX_RADIUS = 32
MF_REC_COLS = 12
MF_FRAME_COLS = 10
MF_TYPED_F32_COLS = 7
MF_DECODE_TRACES_DECODED_ONLY = 1
MF_DECODE_KEEP_RUNNER_UP = 2
MF_PP_MAX_RADIUS = 8
MF_HIST_BINS = 65534
MF_LABEL_COUNTS_LEN = 2 + 2 * 64
MF_FRAME_STATIC_MAX_ROWS = 1048576

# every symbol include/merfish_hip.h declares
EXPORTS = ('mf_version', 'mf_last_error_string', 'mf_gaussian_taps',
           'mf_lowpass_workspace_bytes', 'mf_lowpass_images',
           'mf_codebook_table_bytes', 'mf_codebook_prepare',
           'mf_decode_pixels', 'mf_label_components',
           'mf_region_workspace_bytes', 'mf_region_table', 'mf_refactor_sums',
           'mf_region_frame_workspace_bytes', 'mf_region_frame',
           'mf_region_frame_static', 'mf_region_frame_typed',
           'mf_high_pass_filter', 'mf_preprocess_workspace_bytes',
           'mf_deconvolve_lucyrichardson', 'mf_preprocess_images',
           'mf_deconvolve_lucyrichardson_guo',
           'mf_deconvolve_lucyrichardson_kernel2d')

_lib = None


class MerfishHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MerfishHipError(
            'libmerfish_hip.so is not built ({}). Build it with '
            '`python -m merlin_amd._build` (needs hipcc); there is no CPU '
            'fallback.'.format(LIB_PATH))
    L = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, i32, i64, f64, f32 = c.c_void_p, c.c_int, c.c_int64, c.c_double, c.c_float
    L.mf_version.restype = i32
    L.mf_version.argtypes = []
    L.mf_last_error_string.restype = c.c_char_p
    L.mf_last_error_string.argtypes = []
    L.mf_lowpass_workspace_bytes.restype = i64
    L.mf_lowpass_workspace_bytes.argtypes = [i32, i64, i32, i32]
    L.mf_lowpass_images.restype = i32
    L.mf_lowpass_images.argtypes = [vp, i32, i64, i32, i32, i64, f64, vp, vp, i64, vp]
    L.mf_gaussian_taps.restype = i32
    L.mf_gaussian_taps.argtypes = [i32, f64, c.POINTER(c.c_uint32),
                                   c.POINTER(c.c_float)]
    L.mf_codebook_table_bytes.restype = i64
    L.mf_codebook_table_bytes.argtypes = [i32, i32]
    L.mf_codebook_prepare.restype = i32
    L.mf_codebook_prepare.argtypes = [c.POINTER(c.c_double), i32, i32, vp, i64]
    L.mf_decode_pixels.restype = i32
    L.mf_decode_pixels.argtypes = [
        vp, i32, i32, i32, i32, i32, i64, i64, vp, i32, i32,
        c.POINTER(c.c_float), c.POINTER(c.c_float), f64, f64, f32,
        vp, vp, vp, i32, vp, i64, i64, i64, i32, vp]
    L.mf_label_components.restype = i32
    L.mf_label_components.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp]
    L.mf_region_workspace_bytes.restype = i64
    L.mf_region_workspace_bytes.argtypes = [i32, i32, i32, i64, i64]
    L.mf_region_table.restype = i32
    L.mf_region_table.argtypes = [
        vp, vp, vp, i32, vp, i64, i64, i64, vp, i32, i32, i32, i32, i32, i32,
        i64, i64, vp, vp, vp, vp, vp, i64, vp]
    L.mf_refactor_sums.restype = i32
    L.mf_refactor_sums.argtypes = [vp, vp, i64, i32, i32, i32, i32, f64, vp, i32,
                                   vp, vp, vp]
    L.mf_region_frame_workspace_bytes.restype = i64
    L.mf_region_frame_workspace_bytes.argtypes = [i64]
    L.mf_region_frame.restype = i32
    L.mf_region_frame.argtypes = [vp, vp, i64, i32, i32, f64, f64, f64, f64,
                                  vp, vp, vp, vp, i64, vp]
    L.mf_region_frame_static.restype = i32
    L.mf_region_frame_static.argtypes = [vp, vp, i64, i32, i32, f64, f64, f64,
                                         f64, vp, vp, vp, vp]
    L.mf_region_frame_typed.restype = i32
    L.mf_region_frame_typed.argtypes = [vp, vp, i64, i32, i32, f64, f64, f64,
                                        f64, vp, vp, vp, vp, vp, vp, vp, i64,
                                        vp]
    L.mf_high_pass_filter.restype = i32
    L.mf_high_pass_filter.argtypes = [vp, i32, i32, i32, i32, f64, vp, vp]
    L.mf_preprocess_workspace_bytes.restype = i64
    L.mf_preprocess_workspace_bytes.argtypes = [i32, i32, i32]
    L.mf_deconvolve_lucyrichardson.restype = i32
    L.mf_deconvolve_lucyrichardson.argtypes = [
        vp, i32, i32, i32, i32, i32, f64, i32, vp, i32, vp, i32, vp, i32, vp,
        i64, vp]
    L.mf_deconvolve_lucyrichardson_guo.restype = i32
    L.mf_deconvolve_lucyrichardson_guo.argtypes = [
        vp, i32, i32, i32, i32, i32, c.POINTER(c.c_double),
        c.POINTER(c.c_double), i32, vp, i32, vp, i32, i32, vp, i64, vp]
    L.mf_deconvolve_lucyrichardson_kernel2d.restype = i32
    L.mf_deconvolve_lucyrichardson_kernel2d.argtypes = [
        vp, i32, i32, i32, i32, i32, c.POINTER(c.c_double), i32, vp, i32, vp,
        i32, vp, i32, vp, i64, vp]
    L.mf_preprocess_images.restype = i32
    L.mf_preprocess_images.argtypes = [
        vp, i32, i32, i32, f64, i32, f64, i32, vp, vp, i32, i32, vp, i64, vp]
    _lib = L
    return L


def check(rc):
    if rc < 0:
        msg = lib().mf_last_error_string().decode('utf-8', 'replace')
        if rc == -2:
            raise NotImplementedError('libmerfish_hip: ' + msg)
        if rc == -1:
            raise ValueError('libmerfish_hip: ' + msg)
        raise MerfishHipError('libmerfish_hip error {}: {}'.format(rc, msg))
    return rc


def gaussian_taps(ksize, sigma):
    """(fixed-point taps uint32[ksize], float taps float32[ksize])."""
    import numpy as np
    fx = (ctypes.c_uint32 * ksize)()
    fl = (ctypes.c_float * ksize)()
    check(lib().mf_gaussian_taps(int(ksize), float(sigma), fx, fl))
    return np.array(fx[:], dtype=np.uint32), np.array(fl[:], dtype=np.float32)
