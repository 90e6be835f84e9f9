"""bench.py -- decoded Mpixels/s for the pixel-decode hot path on MI355X.

One "step" = one full field of view (BASELINE.json configs[1]: 2048x2048,
7 z-planes, 16-bit MHD4 codebook with 140 codewords) taken from a uint16 stack
already resident in HBM to the barcode table on the host:
decode_pixels for every plane (one launch) -> connected components -> region
table -> D2H of the records -> the reference's DataFrame.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

With N ranks every rank decodes its own FOVs (FOVs are independent units,
analysis/decode.py:71-72): weak scaling, no data-path collective.
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
            'blur': r['blur'], 'source': 'profiles/cpu_reference_baseline.json'}
    return out


BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA


def codebook_configs(dev):
    """BASELINE configs 3 and 5: the large-codebook nearest-codeword
    path, timed with HIP events around back-to-back decode launches.
    flops: algorithmic = 2*N*B per pixel (the [px,B] x [B,N] contraction of
    decoding.py:112-116); issued = 3 exact bf16 splits x 2 x Npad x K."""
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
        issued = 3 * 2.0 * npad * (16 if bits <= 16 else 32) * px
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
                         'issued_bf16': issued / k_ms / 1e9,
                         'issued_frac': issued / k_ms / 1e9 / BF16_PEAK_TFLOPS}}
        del img, bc, vol, task
        torch.cuda.empty_cache()
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
    ap.add_argument('--backend', default='nccl',
                    help='torch.distributed backend for --gpus > 1 (nccl = RCCL; '
                         'gloo only to rehearse the multi-rank path on one GPU)')
    ap.add_argument('--serial', action='store_true',
                    help='one run_fov call per step instead of the pipelined run_fovs')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist_on = world > 1
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if dist_on:
        import torch.distributed as dist
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)

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

    def run(count):
        # a step = one FOV through Decode.run_fovs (stack in HBM -> barcode
        # DataFrame on the host); run_fovs overlaps FOV k's PCIe/host tail
        # with FOV k+1's kernels, and every FOV's DataFrame is complete
        # before the loop ends
        bc = vol = None
        stacks = (fovs[i % 2] for i in range(count))
        for bc, vol in task.run_fovs(stacks, sf, bg):
            pass
        return bc, vol

    def barrier():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.serial:                          # one run_fov call per step
        def run(count):                      # noqa: F811
            bc = vol = None
            for i in range(count):
                bc, vol = task.run_fov(fovs[i % 2], sf, bg, fov=i)
            return bc, vol

    if args.warmup:
        run(args.warmup)
    barrier()
    task.kernel_events = []          # HIP events around every decode launch
    t0 = time.perf_counter()
    bc, vol = run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    k_events, task.kernel_events = task.kernel_events, None
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64,
                         device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_records = len(bc)
    n_decoded = int((vol.decoded >= 0).sum())
    del bc, vol

    # ---- the one collective of the path (outside the timed decode loop, which
    # has none): OptimizeIteration's exchange of the per-fragment refactors,
    # background refactors and barcode counts (optimize.py:328-345, 358-373,
    # 406-408) as ONE packed [F, 2B+N] sum-all-reduce over RCCL ----
    exchange = None
    if dist_on:
        from merlin_amd import optimize
        F, N = 50, 140
        mine = optimize.shard_fragments(F, rank, world)
        rng = np.random.default_rng(rank)
        rf, bgr = rng.uniform(0.5, 1.5, (len(mine), B)), rng.normal(size=(len(mine), B))
        seen = rng.integers(0, 50, (len(mine), N)).astype(np.float64)
        for _ in range(3):
            got = optimize.exchange_fragment_results(rf, bgr, seen, mine, F, dev)
        barrier()
        t1 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            got = optimize.exchange_fragment_results(rf, bgr, seen, mine, F, dev)
        torch.cuda.synchronize(dev)
        ex_us = (time.perf_counter() - t1) / reps * 1e6
        ok = bool(np.array_equal(got[0][mine], rf) and np.array_equal(got[2][mine], seen))
        exchange = {'rccl_ranks': world, 'backend': args.backend,
                    'exchange_us': ex_us,
                    'payload_bytes': F * (2 * B + N) * 8, 'own_rows_intact': ok,
                    'note': 'one packed [F, 2B+N] float64 all-reduce per '
                            'OptimizeIteration, host numpy in -> host numpy out'}

    # ---- roofline of the dominant kernel (decode_kernel): HIP events around
    # each of its launches inside the timed region, on the stream the library
    # launches on (torch's current stream); the launch is the only work between
    # the two events of a pair ----
    px_fov = Z * H * W
    k_ms = sum(a.elapsed_time(b) for a, b in k_events) / max(len(k_events), 1)
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
    # algorithmic bytes per pixel: 2B in (u16 stack) + 2 (int16 id) + 4
    # (magnitude) + 8 (float64 distance) = 2B + 14, plus 4B of normalised
    # traces for every decoded pixel (the FOV path keeps no others)
    alg_bytes = px_fov * (2 * B + 14) + n_decoded * 4 * B
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    # SURVEY 8d's figure: 2B + 10 per pixel (float32 distances, no traces)
    survey_bytes = px_fov * (2 * B + 10)

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

    # ---- PCIe-inclusive rate (never `value`): the same FOVs starting in
    # page-locked HOST memory, uploaded by the pinned multi-buffered pipeline
    # (Decode.run_fovs_from_host) one FOV ahead of the kernels ----
    host_line = None
    if rank == 0 and world == 1 and not args.no_host_pipeline:
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

    # ---- the stage in front of the decoder, reported beside
    # the headline number: high-pass + 20 float64 Lucy-Richardson iterations per
    # 2048 x 2048 camera frame (112 frames make one FOV) ----
    pre_line = None
    if rank == 0 and world == 1 and not args.no_preprocess:
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
        # bytes this implementation must move (646 B/px), and the
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
            'config': {
                'workload': 'single FOV 2048x2048x7 z-planes, 16-bit MHD4 '
                            'codebook (140 codewords), u16 stack resident in '
                            'HBM -> barcode DataFrame on host',
                'bits': B, 'z': Z, 'h': H, 'w': W, 'codewords': 140,
                'spots_per_plane': SPOTS_PER_PLANE,
                'records_per_fov': n_records, 'decoded_px_per_fov': n_decoded,
                'parallelism': 'fov-sharded x{}'.format(world)},
            'roofline': {
                'bound': 'hbm', 'kernel': 'decode_kernel<u16,R=2,B<=16>',
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                'frac_survey_bytes': survey_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'survey_bytes': survey_bytes, 'valu': valu,
                'stale_counter_files': stale,
                'kernel_ms': k_ms, 'launches_timed': len(k_events),
                'alone': {'kernel_ms': k_alone_ms,
                          'achieved': alg_bytes / (k_alone_ms * 1e-3) / 1e9,
                          'frac': alg_bytes / (k_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          'note': 'same kernel back to back on an otherwise idle GPU; '
                                  'in the timed region it overlaps the previous '
                                  "FOV's region-table stage"},
                'algorithmic_bytes': alg_bytes},
        }
        if host_line is not None:
            out['host_pipeline'] = host_line
        if pre_line is not None:
            out['preprocess'] = pre_line
        if exchange is not None:
            out['optimize_exchange'] = exchange
        if world == 1 and not args.no_codebooks:
            out['codebooks'] = codebook_configs(dev)
        if not args.no_cpu_baseline and world == 1:
            plane = fovs[0][:, 0].cpu().numpy()
            out['cpu_baseline'] = cpu_baseline(codebook, plane, sf, bg,
                                               task.parameters)
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
