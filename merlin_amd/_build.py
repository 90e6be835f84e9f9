"""Build libmerfish_hip.so in-tree with hipcc for gfx950.

Cross-compiles without a GPU.  ``python -m merlin_amd._build`` or
``__graft_entry__.build()``.  Objects are rebuilt only when a source or header
is newer than the object.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(ROOT, 'include')
BUILD = os.path.join(HERE, 'csrc', 'build')
LIB = os.path.join(HERE, 'libmerfish_hip.so')

# (source, object stem, extra flags).  decode.hip is compiled once per group of
# blur radii (2g, 2g + 1; group 0 carries the host code): four objects built
# side by side instead of one five-minute translation unit.
DECODE_GROUPS = 4
SOURCES = [('common.cpp', 'common', [])] + \
    [('decode.hip', 'decode_g{}'.format(g), ['-DMF_INST_GROUP={}'.format(g)])
     for g in range(DECODE_GROUPS)] + \
    [('regions.hip', 'regions', []), ('preprocess.hip', 'preprocess', []),
     ('lowpass.hip', 'lowpass', [])]
# -ffp-contract=off: the kernels reproduce numpy's unfused float arithmetic;
# fused multiply-adds are written explicitly where they are wanted.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
         '-ffp-contract=off', '-I' + INCLUDE, '-I' + CSRC,
         '-Wno-unused-result']
# Per-file extras.  preprocess.hip: the float64 filter chains are long runs of
# dependent multiply / add; the default scheduler (minimum registers at the
# occupancy bound) leaves every instruction waiting on the one before it, the
# ILP-first strategy interleaves the independent chains (two columns per lane,
# products ahead of sums) and needs fewer registers doing so.
FILE_FLAGS = {'preprocess.hip': ['-mllvm', '-amdgpu-sched-strategy=max-ilp']}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.exists(c) or c == 'hipcc'):
            return c
    raise RuntimeError('hipcc not found')


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(item):
    src, stem, extra = item
    obj = os.path.join(BUILD, stem + '.o')
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, 'common.h'),
            os.path.join(INCLUDE, 'merfish_hip.h'), os.path.abspath(__file__)]
    if _stale(obj, deps):
        cmd = [_hipcc(), '-x', 'hip'] + FLAGS + FILE_FLAGS.get(src, []) + extra \
            + ['-c', os.path.join(CSRC, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for {}:\n{}'.format(src, r.stderr))
    return obj


def build(verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB, objs):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr)
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(verbose=True)
    sys.exit(0)
