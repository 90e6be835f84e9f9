"""Per-stage timing of one FOV (development aid)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
from merlin_amd import _lib

dev = torch.device('cuda:0')
cb = bench.load_codebook()
task = Decode(cb, {'crop_width': 100}, device=dev)
t = time.time(); img = synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234, dev); torch.cuda.synchronize(); print('gen', time.time()-t)
sf = bench.initial_scale_factors(img); print('sf', sf)
bg = np.zeros(16)
def T(f, n=3):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3, r
ms, vol = T(lambda: task.decode_fov_device(img, sf, bg)); print('decode_kernel ms', ms)
print('decoded px', int((vol.decoded>=0).sum()), 'of', vol.decoded.numel())
L = _lib.lib()
Z,H,W = vol.shape
labels = torch.empty((Z,H,W), dtype=torch.int32, device=dev); counts = torch.zeros(2, dtype=torch.int64, device=dev)
sp = torch.cuda.current_stream().cuda_stream
ms,_ = T(lambda: L.mf_label_components(vol.decoded.data_ptr(), Z,H,W,0, labels.data_ptr(), counts.data_ptr(), sp)); print('label ms', ms, counts.cpu())
def tab():
    vol._tables.clear(); return task.decoder.region_table(vol, False)
ms, table = T(tab); print('region_table (label+table+D2H) ms', ms, len(table))
def frames():
    return task.decoder._frame_from_rows(table, np.arange(len(table)), False, 16, 0, 100, 0, None, 0, (H,W))
t=time.perf_counter(); df = frames(); print('one DataFrame all rows ms', (time.perf_counter()-t)*1e3, len(df))
ms, r = T(lambda: task.run_fov(img, sf, bg, 0)); print('run_fov ms', ms, len(r[0]))
