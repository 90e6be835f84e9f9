"""Decode throughput for BASELINE configs 3 and 5 (large codebooks, the MFMA
nearest-codeword path).  Prints one JSON line per config:
  cfg3: 22 bits, 1000 weight-4 codewords, one 2048x2048 plane
  cfg5: 32 bits, 10000 weight-4 codewords, 10 planes of 2048x2048 (the f16
        table is 640 KB: streamed through LDS in chunks)
flops: algorithmic = 2*N*B per pixel (the [px,B]x[B,N] contraction);
issued = 2 float16 terms * 2 * Npad * K (K = 16 or 32) per pixel."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlin_amd.codebook import random_weight_codebook
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov

dev = torch.device('cuda:0')
BF16_PEAK = 2500.0   # TFLOP/s dense f16 / bf16, MI355X_MICROARCH.md

def run(name, B, N, Z, seed, reps):
    cb = random_weight_codebook(B, N, 4, seed=seed)
    task = Decode(cb, {'crop_width': 100}, device=dev)
    img = synthetic_fov(cb.get_barcodes(), Z, 2048, 2048, 40000, 1234, dev)
    sf = np.full(B, 500.0); bg = np.zeros(B)
    px = Z * 2048 * 2048
    for _ in range(2):
        v = task.decode_fov_device(img, sf, bg)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        v = task.decode_fov_device(img, sf, bg)
    e1.record(); torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / reps
    ndec = int((v.decoded >= 0).sum())
    del v
    t0 = time.perf_counter()
    for i in range(reps):
        bc, vol = task.run_fov(img, sf, bg, fov=i)
    torch.cuda.synchronize()
    fov_ms = (time.perf_counter() - t0) / reps * 1e3
    npad = (N + 31) // 32 * 32
    K = 16 if B <= 16 else 32
    alg = 2.0 * N * B * px
    issued = 2 * 2.0 * npad * K * px        # two float16 terms of the trace
    print(json.dumps({
        'config': name, 'bits': B, 'codewords': N, 'z': Z, 'pixels': px,
        'decode_kernel_ms': k_ms, 'decode_Mpx_per_s': px / k_ms / 1e3,
        'run_fov_ms': fov_ms, 'run_fov_Mpx_per_s': px / fov_ms / 1e3,
        'decoded_px': ndec, 'records': len(bc),
        'roofline': {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': BF16_PEAK,
                     'achieved': alg / (k_ms * 1e-3) / 1e12,
                     'frac': alg / (k_ms * 1e-3) / 1e12 / BF16_PEAK,
                     'issued_f16': issued / (k_ms * 1e-3) / 1e12,
                     'issued_frac': issued / (k_ms * 1e-3) / 1e12 / BF16_PEAK}}), flush=True)
    del img, bc, vol
    torch.cuda.empty_cache()

run('cfg3_22bit_1000cw_1z', 22, 1000, 1, 7, 10)
run('cfg5_32bit_10000cw_10z', 32, 10000, 10, 11, 3)
