"""bench.py -- decoded Mpixels/s for the pixel-decode hot path on MI355X.

One "step" = one full field of view (BASELINE.json configs[1]: 2048x2048,
7 z-planes, 16-bit MHD4 codebook with 140 codewords) taken from a uint16 stack
already resident in HBM to the barcode table on the host:
decode_pixels for every plane (one launch) -> connected components -> region
table -> D2H of the records -> the reference's DataFrame.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N          # no launcher: starts the N ranks itself

With N ranks every rank decodes its own FOVs (FOVs are independent units,
analysis/decode.py:71-72): weak scaling, no data-path collective.  The ranks
RCCL saw are what the line reports as ``n_gpus``; ``--gpus`` must agree with
the launcher's WORLD_SIZE, and without a launcher ``--gpus N`` spawns N fresh
rank processes before anything touches a GPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

Z, H, W, B = 7, 2048, 2048, 16
SPOTS_PER_PLANE = 40000
SEED = 1234
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def load_codebook():
    from merlin_amd.codebook import Codebook
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'testsuite_inputs.npz'))
    return Codebook(d['codebook'])


def initial_scale_factors(img):
    """optimize.py:140-155 on the device: argmin|cdf - 0.9| + 2 per bit."""
    out = np.zeros(img.shape[0])
    for b in range(img.shape[0]):
        h = torch.bincount(img[b].reshape(-1).to(torch.int32),
                           minlength=65535)[:65534].double()
        c = torch.cumsum(h, 0)
        c = c / c[-1]
        out[b] = float(torch.argmin(torch.abs(c - 0.9))) + 2
    return out


def cpu_baseline(codebook, img_plane, sf, bg, params):
    """The oracle (numpy port of the reference algorithm) timed on one
    2048x2048 plane of the same workload (1/7 of a FOV, ~10-15 s), single
    core."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import warnings
    import merfish_oracle as oracle
    from threadpoolctl import threadpool_limits
    dm = oracle.normalized_codebook(codebook.get_barcodes())
    with threadpool_limits(limits=1):        # cores: 1 means one thread
        t0 = time.perf_counter()
        di, pm, npt, d = oracle.decode_pixels(
            img_plane, dm, sf, bg, params['distance_threshold'], 1,
            params['lowpass_sigma'])
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            oracle.extract_all_barcodes(
                dm.shape[0], di, pm, npt, d, 0, params['crop_width'], 0, None,
                params['minimum_area'])
        dt = time.perf_counter() - t0
    px = img_plane.shape[1] * img_plane.shape[2]
    out = {'value': px / dt / 1e6, 'unit': 'Mpx/s', 'cores': 1,
           'kind': 'port',
           'sample': 'one {}x{} plane x 16 bits of the same synthetic '
                     'FOV: oracle decode_pixels + 140x extract_barcodes '
                     '({:.1f} s)'.format(img_plane.shape[1],
                                         img_plane.shape[2], dt)}
    # The reference as written (per-pixel Python loops + sklearn + skimage)
    # cannot travel to the GPU box; it was timed in the build container by
    # oracle/time_reference.py (1 core, 256^2..1024^2 planes, linear in pixels)
    ref = os.path.join(ROOT, 'profiles', 'cpu_reference_baseline.json')
    if os.path.exists(ref):
        r = json.load(open(ref))
        out['reference_as_written'] = {
            'value': r['extrapolated']['Mpx_per_s'], 'unit': 'Mpx/s',
            'fov_7z_s': r['extrapolated']['fov_7z_2048_s'],
            'extrapolated': True, 'cores': r['cores'],
            'host': r['cpu_model'] + ' (build container, not this host)',
            'blur': r['blur'], 'source': 'profiles/cpu_reference_baseline.json',
            # the reference's extraction time grows with the number of regions:
            # the plane it was timed on is sparser than the bench FOV's planes
            'sample_barcodes_per_Mpx': r['measured'][-1]['barcodes']
            / (r['measured'][-1]['pixels'] / 1e6),
            'sample_decoded_px_fraction': r['measured'][-1]['decoded_px']
            / r['measured'][-1]['pixels']}
    return out


BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 MFMA


def codebook_configs(dev):
    """BASELINE configs 3 and 5: the large-codebook nearest-codeword
    path, timed with HIP events around back-to-back decode launches.
    flops: algorithmic = 2*N*B per pixel (the [px,B] x [B,N] contraction of
    decoding.py:112-116); issued = 2 float16 terms of the trace x 2 x Npad x K
    (v_mfma_f32_32x32x16_f16, the same dense peak as bf16)."""
    from merlin_amd.codebook import random_weight_codebook
    from merlin_amd.decode import Decode
    from merlin_amd.synthetic import synthetic_fov
    out = {}
    for name, bits, words, z, seed, reps in (
            ('config3_22bit_1000cw_1z', 22, 1000, 1, 7, 10),
            ('config5_32bit_10000cw_10z', 32, 10000, 10, 11, 3)):
        cb = random_weight_codebook(bits, words, 4, seed=seed)
        task = Decode(cb, {'crop_width': 100}, device=dev)
        img = synthetic_fov(cb.get_barcodes(), z, H, W, SPOTS_PER_PLANE,
                            SEED, dev)
        sf, bg = np.full(bits, 500.0), np.zeros(bits)
        for _ in range(2):
            v = task.decode_fov_device(img, sf, bg)
        torch.cuda.synchronize(dev)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            v = task.decode_fov_device(img, sf, bg)
        e1.record()
        torch.cuda.synchronize(dev)
        k_ms = e0.elapsed_time(e1) / reps
        del v
        t0 = time.perf_counter()
        for i in range(reps):
            bc, vol = task.run_fov(img, sf, bg, fov=i)
        torch.cuda.synchronize(dev)
        fov_ms = (time.perf_counter() - t0) / reps * 1e3
        px = z * H * W
        npad = (words + 31) // 32 * 32
        alg = 2.0 * words * bits * px
        issued = 2 * 2.0 * npad * (16 if bits <= 16 else 32) * px
        out[name] = {
            'bits': bits, 'codewords': words, 'z': z,
            'decode_kernel_ms': k_ms,
            'decode_kernel_ms_per_plane': k_ms / z,
            'decode_Mpx_per_s': px / k_ms / 1e3,
            'run_fov_ms': fov_ms, 'records': len(bc),
            'roofline': {'bound': 'mfma', 'unit': 'TFLOP/s',
                         'peak': BF16_PEAK_TFLOPS,
                         'achieved': alg / k_ms / 1e9,
                         'frac': alg / k_ms / 1e9 / BF16_PEAK_TFLOPS,
                         'issued_f16': issued / k_ms / 1e9,
                         'issued_frac': issued / k_ms / 1e9 / BF16_PEAK_TFLOPS}}
        del img, bc, vol, task
        torch.cuda.empty_cache()
    return out


def spawn_ranks(n, argv):
    """``--gpus N`` without a launcher: start the N rank processes ourselves.
    The children are FRESH interpreters created before this process has made
    any GPU call (device_count() does not initialise the GPU on this image);
    the parent only waits and returns the worst exit code."""
    import socket
    import subprocess
    backend = 'nccl'
    for i, a in enumerate(argv):
        if a == '--backend' and i + 1 < len(argv):
            backend = argv[i + 1]
        elif a.startswith('--backend='):
            backend = a.split('=', 1)[1]
    if backend == 'nccl':
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write(
                'bench.py: --gpus {} asked for but {} GPU(s) visible; RCCL needs '
                'one GPU per rank (--backend gloo rehearses the multi-rank path '
                'on fewer)\n'.format(n, have))
            return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def timed_steady(task, fovs, sf, bg, steps, warmup, barrier):
    """ONE run_fovs generator across warm-up and the timed window.  After
    `warmup` yielded FOVs the pipeline is full (one FOV decoded, two with their
    tables queued); the window then runs from a barrier + device sync to a
    device sync + barrier around exactly `steps` more yields: every stage --
    decode, labelling, table, frame rows, DataFrame -- executes `steps` times
    inside it, nothing of the fill and nothing beyond the drain of the stages
    it queued itself."""
    import itertools
    stacks = (fovs[i % len(fovs)] for i in itertools.count())
    gen = task.run_fovs(stacks, sf, bg)
    for _ in range(warmup):
        next(gen)
    barrier()
    task.kernel_events = []
    t0 = time.perf_counter()
    for _ in range(steps):
        bc, vol = next(gen)
    barrier()
    dt = time.perf_counter() - t0
    events, task.kernel_events = task.kernel_events, None
    gen.close()
    return dt, events, bc, vol


def timed_cold(task, fovs, sf, bg, steps, barrier, serial=False):
    """`steps` FOVs through a pipeline that starts empty and is drained inside
    the window (r01/r02's definition of a step)."""
    barrier()
    task.kernel_events = []
    t0 = time.perf_counter()
    bc = vol = None
    if serial:
        for i in range(steps):
            bc, vol = task.run_fov(fovs[i % len(fovs)], sf, bg, fov=i)
    else:
        for bc, vol in task.run_fovs((fovs[i % len(fovs)] for i in range(steps)),
                                     sf, bg):
            pass
    barrier()
    dt = time.perf_counter() - t0
    events, task.kernel_events = task.kernel_events, None
    return dt, events, bc, vol


def config4(codebook, dev, rank, world, group_backend, n_fovs=400, F=50,
            iterations=3, chunk=100, hw=(H, W), z=Z, spots=SPOTS_PER_PLANE):
    """BASELINE configs[3] / SURVEY 8d config 4: `n_fovs` synthetic FOVs (seed
    1234 + fov), FOV i on rank i mod world; first the scale-factor loop --
    `iterations` chained OptimizeIteration passes of F random (fov, z)
    fragments (one random stream, seed 99), with optimize_background off and on, each pass
    ending in the packed [F, 2B+N] all-reduce -- then Decode of every FOV with
    the factors the last pass produced (decode.py:107-110), through
    run_fovs_sharded.  FOVs are generated on the device outside the clocks, at
    most `chunk` resident per rank at a time."""
    import torch.distributed as dist
    from merlin_amd import optimize
    from merlin_amd.decode import Decode, shard_fovs
    from merlin_amd.synthetic import synthetic_fov
    dist_on = world > 1
    bits = codebook.get_barcodes().shape[1]
    h, w = hw

    def barrier():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()

    def max_over_ranks(x):
        if not dist_on:
            return x
        t = torch.tensor([x], dtype=torch.float64,
                         device=dev if group_backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    fovs = list(range(n_fovs))

    # ---- OptimizeIteration chain.  The draw of pass k is a function of the
    # seed alone, so the planes this rank will be asked for are known: they are
    # generated before the clock starts (the reference reads them from disk).
    out = {'fovs': n_fovs, 'fragments_per_iteration': F, 'iterations': iterations,
           'ranks': world}
    planes = {}
    draws = []
    rng = np.random.RandomState(99)       # one stream, as run_optimize_chain
    for k in range(iterations):
        draws.append([[int(rng.choice(fovs)), int(rng.choice(list(range(z))))]
                      for _ in range(F)])
    for d in draws:
        for i in optimize.shard_fragments(F, rank, world):
            f, zz = d[i]
            if (f, zz) not in planes:
                planes[(f, zz)] = synthetic_fov(
                    codebook.get_barcodes(), z, h, w, spots, SEED + f,
                    dev)[:, zz].clone()
    hist = optimize.pixel_histogram(torch.stack(list(planes.values()), 1)) \
        if planes else torch.zeros((bits, 65534), dtype=torch.float64, device=dev)
    init = optimize.initial_scale_factors(optimize.summed_pixel_histogram(hist))
    for name, bgopt in (('optimize', False), ('optimize_background', True)):
        params = {'fov_per_iteration': F, 'optimize_background': bgopt}
        # untimed pass: streams, allocator, codebook table
        optimize.run_optimize_chain(codebook, lambda f, zz: planes[(f, zz)], fovs, z,
                                    init, 1, params, dev, None, 99)
        barrier()
        t0 = time.perf_counter()
        chain = optimize.run_optimize_chain(
            codebook, lambda f, zz: planes[(f, zz)], fovs, z, init, iterations,
            params, dev, None, 99)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        assert [c.parameters['fov_index'] for c in chain] == draws
        out[name] = {'ms_per_iteration': dt / iterations * 1e3,
                     'ms_per_fragment': dt / iterations / F * 1e3,
                     'scale_factors': [round(float(v), 3)
                                       for v in chain[-1].get_scale_factors()],
                     'barcodes_per_fragment': float(np.mean(
                         chain[-1].get_barcode_count_history()[-1].sum()))}
        if bgopt:
            out[name]['backgrounds'] = [round(float(v), 3)
                                        for v in chain[-1].get_backgrounds()]
        else:
            sf, bg = chain[-1].get_scale_factors(), chain[-1].get_backgrounds()
    del planes

    # the exchange step on its own (one [F, 2B+N] all-reduce)
    N = codebook.get_barcodes().shape[0]
    mine_f = optimize.shard_fragments(F, rank, world)
    rng = np.random.default_rng(rank)
    rf = rng.uniform(0.5, 1.5, (len(mine_f), bits))
    bgr = rng.normal(size=(len(mine_f), bits))
    seen = rng.integers(0, 50, (len(mine_f), N)).astype(np.float64)
    for _ in range(3):
        got = optimize.exchange_fragment_results(rf, bgr, seen, mine_f, F, dev)
    barrier()
    t1 = time.perf_counter()
    for _ in range(20):
        got = optimize.exchange_fragment_results(rf, bgr, seen, mine_f, F, dev)
    torch.cuda.synchronize(dev)
    out['exchange_us'] = max_over_ranks((time.perf_counter() - t1) / 20 * 1e6)
    out['exchange_payload_bytes'] = F * (2 * bits + N) * 8
    out['exchange_own_rows_intact'] = bool(
        np.array_equal(got[0][mine_f], rf) and np.array_equal(got[2][mine_f], seen))

    # ---- Decode of every FOV, sharded
    task = Decode(codebook, {'crop_width': 100, 'distance_threshold': 0.5167,
                             'lowpass_sigma': 1, 'minimum_area': 0}, device=dev)
    mine = shard_fovs(fovs, rank, world)
    total = 0.0
    records = 0
    for c0 in range(0, max(len(mine), 1), chunk):
        part = mine[c0:c0 + chunk]
        resident = {f: synthetic_fov(codebook.get_barcodes(), z, h, w, spots,
                                     SEED + f, dev) for f in part}
        if c0 == 0 and part:        # untimed: first-use allocations
            for _ in task.run_fovs((resident[part[0]] for _ in range(4)), sf, bg):
                pass
        barrier()
        t0 = time.perf_counter()
        for f, bc, vol in task.run_fovs_sharded(
                lambda f: resident[f], sf, bg, part, rank=0, world=1):
            records += len(bc)
        barrier()
        total += time.perf_counter() - t0
        del resident
    total = max_over_ranks(total)
    if dist_on:
        t = torch.tensor([records], dtype=torch.float64,
                         device=dev if group_backend == 'nccl' else 'cpu')
        dist.all_reduce(t)
        records = int(t.item())
    out['decode'] = {'seconds': total, 'fovs_per_sec': n_fovs / total,
                     'Mpx_per_s': n_fovs * z * h * w / total / 1e6,
                     'ms_per_fov_per_gpu': total / max(len(mine), 1) * 1e3,
                     'records': records,
                     'scale_factors_from': 'optimize (3rd pass)'}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-host-pipeline', action='store_true')
    ap.add_argument('--no-preprocess', action='store_true')
    ap.add_argument('--no-codebooks', action='store_true')
    ap.add_argument('--no-config4', action='store_true')
    ap.add_argument('--no-image-output', action='store_true')
    ap.add_argument('--config4-fovs', type=int, default=400)
    ap.add_argument('--config4-fragments', type=int, default=50)
    ap.add_argument('--config4-size', type=int, default=H,
                    help='FOV edge in pixels for config 4 (tests use small FOVs)')
    ap.add_argument('--backend', default='nccl',
                    help='torch.distributed backend for --gpus > 1 (nccl = RCCL; '
                         'gloo only to rehearse the multi-rank path on one GPU)')
    ap.add_argument('--serial', action='store_true',
                    help='one run_fov call per step instead of the pipelined run_fovs')
    ap.add_argument('--rendezvous-only', action='store_true',
                    help='start the ranks, all-reduce once, print n_gpus and exit '
                         '(checks the launch path; touches no GPU with gloo)')
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # no launcher: be the launcher.  Nothing has touched a GPU yet.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus {} but the launcher started {} rank(s) '
                         '(WORLD_SIZE)\n'.format(args.gpus, world))
        sys.exit(2)
    dist_on = world > 1

    if args.rendezvous_only:
        import torch.distributed as dist
        n = 1
        if dist_on:
            if args.backend == 'nccl':
                torch.cuda.set_device(local_rank)
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
                t = torch.ones(1, device='cuda')
            else:
                dist.init_process_group(args.backend)
                t = torch.ones(1)
            dist.all_reduce(t)
            n = int(t.item())
            assert n == dist.get_world_size()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({'n_gpus': n, 'rendezvous_only': True,
                              'backend': args.backend}))
        return

    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        sys.stderr.write('bench.py: no GPU visible\n')
        sys.exit(2)
    if dist_on and args.backend == 'nccl' and n_dev < world:
        sys.stderr.write('bench.py: {} ranks but {} GPU(s) visible; RCCL needs one GPU '
                         'per rank\n'.format(world, n_dev))
        sys.exit(2)
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if dist_on:
        import torch.distributed as dist
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)
        world = dist.get_world_size()       # the ranks RCCL actually connected

    from merlin_amd.decode import Decode
    from merlin_amd.synthetic import synthetic_fov

    codebook = load_codebook()
    params = {'crop_width': 100, 'distance_threshold': 0.5167,
              'lowpass_sigma': 1, 'minimum_area': 0}
    task = Decode(codebook, params, device=dev)

    # two distinct FOVs per rank, alternated, so no step sees a warm cache of
    # its own previous outputs
    fovs = [synthetic_fov(codebook.get_barcodes(), Z, H, W, SPOTS_PER_PLANE,
                          SEED + 2 * rank + i, dev) for i in range(2)]
    sf = initial_scale_factors(fovs[0])
    bg = np.zeros(B)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if not dist_on:
            return x
        t = torch.tensor([x], dtype=torch.float64,
                         device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # table / staging buffers are carved from torch's caching allocator: give it
    # its working set up front (three FOVs in flight) so that no timed FOV pays a
    # hipMalloc because it is a little larger than every block cached so far;
    # and the page-locked blocks of the frame rows, read once (the first CPU
    # access to a page of a fresh hipHostMalloc block faults: without this the
    # first ~20 FOVs of a process hand their rows over 1.3 ms slower each,
    # dev/first_fovs.py)
    task.reserve_memory(fovs[0].shape, expectedBarcodes=Z * SPOTS_PER_PLANE)

    # A step = one FOV through Decode.run_fovs (stack in HBM -> barcode DataFrame
    # on the host).  Steady state: `steps` FOVs through a full pipeline; cold:
    # the same count through a pipeline that fills and drains inside the window.
    steady = not args.serial and args.warmup >= 4
    if steady:
        dt, k_events, bc, vol = timed_steady(task, fovs, sf, bg, args.steps,
                                             args.warmup, barrier)
    else:
        if args.warmup:
            timed_cold(task, fovs, sf, bg, args.warmup, barrier, args.serial)
        dt, k_events, bc, vol = timed_cold(task, fovs, sf, bg, args.steps, barrier,
                                           args.serial)
    dt = max_over_ranks(dt)
    n_records = len(bc)
    n_decoded = int((vol.decoded >= 0).sum())
    del bc, vol
    dt_cold = None
    if steady:
        dt_cold = max_over_ranks(timed_cold(task, fovs, sf, bg, args.steps,
                                            barrier)[0])

    # ---- roofline of the dominant kernel (decode_kernel): HIP events around
    # each of its launches inside the timed region, on the stream the library
    # launches on (torch's current stream); the launch is the only work between
    # the two events of a pair ----
    px_fov = Z * H * W
    k_ms = sum(a.elapsed_time(b) for a, b in k_events) / max(len(k_events), 1)
    # the pipeline's period: decode-kernel start to decode-kernel start inside
    # the window.  The window itself is steps x period + the drain of the last
    # FOV's labelling / table / rows (about 1 ms: 3 % at 20 steps, 0.3 % at 200)
    period_ms = (k_events[0][0].elapsed_time(k_events[-1][0]) / (len(k_events) - 1)
                 if len(k_events) > 1 else None)
    # the same kernel with the GPU to itself (in the timed region it shares the
    # CUs with the previous FOV's sort / region statistics on a second stream)
    for r in range(12):
        v = task.decode_fov_device(fovs[r % 2], sf, bg)
    torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(20):
        v = task.decode_fov_device(fovs[r % 2], sf, bg)
    e1.record()
    torch.cuda.synchronize(dev)
    k_alone_ms = e0.elapsed_time(e1) / 20
    del v
    # what a plain device-to-device copy reaches on this part (verify
    # the 8 TB/s figure with a memcpy probe): 1 GiB read + 1 GiB written per copy
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize(dev)
    copy_gbs = 10 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    # ALGORITHMIC bytes per launch, SURVEY 8d: 2B (u16 stack) + 2 (int16 id) + 4
    # (magnitude) + 4 (distance) = 2B + 10 = 42 B per pixel x the 29 360 128
    # pixels one launch decodes.  `roofline.achieved` / `frac` use this figure.
    survey_bytes = px_fov * (2 * B + 10)
    achieved = survey_bytes / (k_ms * 1e-3) / 1e9
    # What this implementation has to move for the reference's outputs: float64
    # distances (+4) and 4B of normalised traces per DECODED pixel (the region
    # table reads them): reported as frac_impl_bytes.
    impl_bytes = px_fov * (2 * B + 14) + n_decoded * 4 * B

    # HBM bytes and VALU instructions per launch come from PMC counters, which
    # need separate rocprofv3 --pmc passes (dev/profile.sh, dev/
    # profile_counters.sh) and cannot be read from inside this process.  The
    # files record the sha256 of the kernel source they were measured on: a
    # kernel edited since then reports null instead of a stale number.
    import hashlib
    src_hash = hashlib.sha256(open(os.path.join(
        ROOT, 'merlin_amd', 'csrc', 'decode.hip'), 'rb').read()).hexdigest()
    traffic = valu = None
    stale = []
    tfile = os.path.join(ROOT, 'profiles', 'decode_traffic.json')
    if os.path.exists(tfile):
        t = json.load(open(tfile))
        if t.get('decode_hip_sha256') == src_hash:
            traffic = t.get('hbm_bytes_per_launch')
        else:
            stale.append('profiles/decode_traffic.json')
    cfile = os.path.join(ROOT, 'profiles', 'decode_counters.json')
    if os.path.exists(cfile):
        c = json.load(open(cfile))
        if c.get('decode_hip_sha256') == src_hash:
            n_valu = c['per_launch']['SQ_INSTS_VALU']
            # issue-bound floor: 1024 SIMDs, one wave64 VALU instruction per 2
            # cycles each (157 TFLOP/s fp32 vector peak), at the 2.4 GHz peak
            floor_ms = n_valu * 2 / (256 * 4) / 2.4e9 * 1e3
            valu = {'wave_instructions_per_launch': n_valu,
                    'per_64px_row': n_valu / (px_fov / 64),
                    'issue_floor_ms': floor_ms,
                    'frac_of_issue_peak': floor_ms / k_ms,
                    'source': 'profiles/decode_counters.json'}
        else:
            stale.append('profiles/decode_counters.json')

    single = rank == 0 and world == 1

    # ---- PCIe-inclusive rate (never `value`): the same FOVs starting in
    # page-locked HOST memory, uploaded by the pinned multi-buffered pipeline
    # (Decode.run_fovs_from_host) one FOV ahead of the kernels ----
    host_line = None
    if single and not args.no_host_pipeline:
        hosts = [f.cpu().pin_memory() for f in fovs]
        n_host = max(4, min(args.steps, 20))
        for _ in task.run_fovs_from_host((hosts[i % 2] for i in range(3)), sf, bg):
            pass
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in task.run_fovs_from_host((hosts[i % 2] for i in range(n_host)), sf, bg):
            pass
        torch.cuda.synchronize(dev)
        ht = (time.perf_counter() - t1) / n_host
        host_line = {'ms_per_fov': ht * 1e3, 'Mpx_per_s': px_fov / ht / 1e6,
                     'h2d_GB_per_s': fovs[0].numel() * 2 / ht / 1e9, 'fovs': n_host,
                     'note': 'u16 stack in pinned host memory -> barcode DataFrame; '
                             'H2D of FOV k+1 overlaps the kernels of FOV k'}
        del hosts

    # ---- decode_3d (decode.py:129-161): the same FOVs with one 26-connected
    # labelling of the volume and float32 distances instead of per-plane regions
    mode3d_line = None
    if single and not args.no_host_pipeline:
        t3 = Decode(codebook, dict(params, decode_3d=True), device=dev)
        for _ in t3.run_fovs((fovs[i % 2] for i in range(6)), sf, bg):
            pass
        torch.cuda.synchronize(dev)
        n3 = max(8, min(args.steps, 40))
        t1 = time.perf_counter()
        for bc3, _ in t3.run_fovs((fovs[i % 2] for i in range(n3)), sf, bg):
            pass
        torch.cuda.synchronize(dev)
        h3 = (time.perf_counter() - t1) / n3
        mode3d_line = {'ms_per_fov': h3 * 1e3, 'Mpx_per_s': px_fov / h3 / 1e6,
                       'fovs': n3, 'records_per_fov': len(bc3),
                       'note': 'decode_3d=True: regions are 26-connected across the '
                               '7 planes, distances float32; cold window (fill and '
                               'drain inside).  The crop of decoding.py:244 compares x '
                               'with shape[0], which is Z in 3-D mode: crop_width 100 '
                               'keeps no record, in the reference as here; the table '
                               'before the crop is what the time is spent on'}
        del t3, bc3

    # ---- the reference's default: write_decoded_images (decode.py:47-48,
    # 165-168) -- one 352 MB float32 hyperstack per FOV leaves the GPU and goes
    # to disk.  The headline step has no output directory; this is the same
    # loop with one. ----
    image_line = None
    if single and not args.no_image_output:
        import shutil
        import tempfile

        def image_run(parent, n_img):
            outdir = tempfile.mkdtemp(prefix='mf_bench_img_', dir=parent)
            try:
                wtask = Decode(codebook, params, device=dev, outputDirectory=outdir)
                for _ in wtask.run_fovs((fovs[i % 2] for i in range(2)), sf, bg):
                    pass
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in wtask.run_fovs((fovs[i % 2] for i in range(n_img)), sf, bg,
                                        fovs=list(range(n_img))):
                    pass                 # run_fovs returns with every file on disk
                torch.cuda.synchronize(dev)
                it = (time.perf_counter() - t1) / n_img
                nbytes = os.path.getsize(os.path.join(outdir, 'decoded_0.tif'))
                return {'ms_per_fov': it * 1e3, 'fovs': n_img,
                        'file_bytes_per_fov': nbytes,
                        'GB_per_s_to_files': nbytes / it / 1e9}
            finally:
                shutil.rmtree(outdir, ignore_errors=True)

        image_line = image_run(None, 12)
        image_line['directory'] = 'tempfile.mkdtemp()'
        image_line['pcie_floor_ms'] = image_line['file_bytes_per_fov'] / 55e9 * 1e3
        image_line['note'] = (
            'run_fovs with outputDirectory set: float32 (decoded, magnitude, '
            'distance) hyperstack per FOV; cast + interleave + D2H on a stream of '
            'their own, file writes on 4 worker threads behind the loop; every '
            'file complete before the clock stops')
        # the same into memory-backed files: what is left when the file system
        # is out of the way (D2H + page-cache copy)
        if os.path.isdir('/dev/shm') and shutil.disk_usage('/dev/shm').free > (8 << 30):
            image_line['dev_shm'] = image_run('/dev/shm', 12)

    # ---- the stage in front of the decoder, reported beside
    # the headline number: high-pass + 20 float64 Lucy-Richardson iterations per
    # 2048 x 2048 camera frame (112 frames make one FOV) ----
    pre_line = None
    if single and not args.no_preprocess:
        from merlin_amd.preprocess import DeconvolutionPreprocess
        n_frames = B * Z                          # one FOV's camera frames
        g = torch.Generator(device=dev).manual_seed(SEED)
        frames = (2000 + fovs[0].float()
                  + 10 * torch.randn(fovs[0].shape, device=dev, generator=g)
                  ).clamp_(0, 65535).to(torch.uint16)
        pre = DeconvolutionPreprocess(device=dev)
        pre.preprocess_device(frames)
        torch.cuda.synchronize(dev)
        p0 = torch.cuda.Event(enable_timing=True)
        p1 = torch.cuda.Event(enable_timing=True)
        p0.record()
        for _ in range(3):
            pre.preprocess_device(frames)
        p1.record()
        torch.cuda.synchronize(dev)
        ms_frame = p0.elapsed_time(p1) / 3 / n_frames
        # bytes this implementation must move (630 B/px), and the
        # r01 figure (every iteration 34 B + an initialisation pass: 720 B/px)
        from merlin_amd.preprocess import algorithmic_bytes_per_pixel
        pre_bytes = algorithmic_bytes_per_pixel(20) * H * W
        pre_bytes_r01 = (34 * 20 + 40) * H * W
        pre_line = {'ms_per_frame': ms_frame, 'Mpx_per_s': H * W / ms_frame / 1e3,
                    'ms_per_fov': ms_frame * B * Z, 'dtype': 'f64',
                    'roofline': {'bound': 'hbm', 'achieved': pre_bytes / ms_frame / 1e6,
                                 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                 'frac': pre_bytes / ms_frame / 1e6 / HBM_PEAK_GBS,
                                 'bytes_per_px': pre_bytes // (H * W),
                                 'frac_r01_bytes': pre_bytes_r01 / ms_frame / 1e6 / HBM_PEAK_GBS},
                    'note': 'camera frames (offset 2000 + the synthetic FOV) -> uint16 stack + '
                            'pixel histogram for the {} frames of one FOV'.format(n_frames)}
        del frames, pre
        torch.cuda.empty_cache()

    del fovs
    torch.cuda.empty_cache()
    cfg4 = None
    if not args.no_config4:
        cfg4 = config4(codebook, dev, rank, world, args.backend,
                       n_fovs=args.config4_fovs, F=args.config4_fragments,
                       hw=(args.config4_size, args.config4_size),
                       spots=max(1, SPOTS_PER_PLANE * args.config4_size ** 2
                                 // (H * W)))

    if rank == 0:
        mpx = world * args.steps * px_fov / dt / 1e6
        out = {
            'metric': 'decoded_Mpixels_per_sec',
            'value': mpx, 'unit': 'Mpx/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'u16/f32/f64',
            'data': 'synthetic',
            'fovs_per_sec': world * args.steps / dt,
            'timing': ('steady: one run_fovs pipeline kept full across warm-up and the '
                       'timed window; {} FOVs yielded (DataFrames complete) and {} of '
                       'every stage executed between two barrier + device syncs'
                       .format(args.steps, args.steps)) if steady else
                      ('cold: the pipeline fills and drains inside the window'
                       + (' (--serial: one run_fov per step)' if args.serial else '')),
            'config': {
                'workload': 'single FOV 2048x2048x7 z-planes, 16-bit MHD4 '
                            'codebook (140 codewords), u16 stack resident in '
                            'HBM -> barcode DataFrame on host; no decoded-image '
                            'files (see with_image_output)',
                'bits': B, 'z': Z, 'h': H, 'w': W, 'codewords': 140,
                'spots_per_plane': SPOTS_PER_PLANE,
                'records_per_fov': n_records, 'decoded_px_per_fov': n_decoded,
                'parallelism': 'fov-sharded x{}'.format(world)},
            'roofline': {
                'bound': 'hbm', 'kernel': 'decode_kernel<u16,R=2,B<=16>',
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                'algorithmic_bytes': survey_bytes,
                'bytes_per_px': 2 * B + 10,
                'copy_probe_GBs': copy_gbs,
                'frac_impl_bytes': impl_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'impl_bytes': impl_bytes, 'valu': valu,
                'stale_counter_files': stale,
                'kernel_ms': k_ms, 'launches_timed': len(k_events),
                'alone': {'kernel_ms': k_alone_ms,
                          'achieved': survey_bytes / (k_alone_ms * 1e-3) / 1e9,
                          'frac': survey_bytes / (k_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          'frac_impl_bytes': impl_bytes / (k_alone_ms * 1e-3) / 1e9
                          / HBM_PEAK_GBS,
                          'note': 'same kernel back to back on an otherwise idle GPU; '
                                  'in the timed region it overlaps the previous '
                                  "FOV's region-table stage"}},
        }
        if dt_cold is not None:
            out['ms_per_step_cold'] = dt_cold / args.steps * 1e3
        out['fov_period_ms'] = period_ms
        if host_line is not None:
            out['host_pipeline'] = host_line
        if image_line is not None:
            out['with_image_output'] = image_line
        if mode3d_line is not None:
            out['decode_3d'] = mode3d_line
        if pre_line is not None:
            out['preprocess'] = pre_line
        if cfg4 is not None:
            out['config4'] = cfg4
        if dist_on:
            out['optimize_exchange'] = {
                'rccl_ranks': world, 'backend': args.backend,
                'exchange_us': cfg4['exchange_us'] if cfg4 else None,
                'payload_bytes': cfg4['exchange_payload_bytes'] if cfg4 else None,
                'note': 'one packed [F, 2B+N] float64 all-reduce per '
                        'OptimizeIteration, host numpy in -> host numpy out'}
        if world == 1 and not args.no_codebooks:
            out['codebooks'] = codebook_configs(dev)
        if not args.no_cpu_baseline and world == 1:
            fov0 = synthetic_fov(codebook.get_barcodes(), Z, H, W, SPOTS_PER_PLANE,
                                 SEED, dev)
            plane = fov0[:, 0].cpu().numpy()
            del fov0
            out['cpu_baseline'] = cpu_baseline(codebook, plane, sf, bg,
                                               task.parameters)
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
