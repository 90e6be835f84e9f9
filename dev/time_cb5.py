"""decode kernel time of config 5 (32 bits x 10000 codewords, 10 z) and config
3, for the library MERFISH_HIP_LIB names; checks the decoded image against the
default library's if dev/variants/ref_cb5.npy exists (else writes it)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlin_amd.codebook import random_weight_codebook
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
for name, B, N, Z, seed, reps in (('cfg3', 22, 1000, 1, 7, 20), ('cfg5', 32, 10000, 10, 11, 5)):
    cb = random_weight_codebook(B, N, 4, seed=seed)
    task = Decode(cb, {'crop_width': 100}, device=dev)
    img = synthetic_fov(cb.get_barcodes(), Z, 2048, 2048, 40000, 1234, dev)
    sf = np.full(B, 500.0); bg = np.zeros(B)
    for _ in range(2):
        v = task.decode_fov_device(img, sf, bg)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        v = task.decode_fov_device(img, sf, bg)
    e1.record(); torch.cuda.synchronize()
    h = int(v.decoded.to(torch.int64).add(2).mul(torch.arange(v.decoded.numel(), device=dev).view(v.decoded.shape) % 1000003).sum())
    hd = float(v.distances.sum())
    print(os.path.basename(os.environ.get('MERFISH_HIP_LIB', 'default')), name,
          '%.3f ms' % (e0.elapsed_time(e1) / reps), 'checksum', h, repr(hd), flush=True)
    del v, img
