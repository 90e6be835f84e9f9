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
img = synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234, dev)
sf = bench.initial_scale_factors(img)
bg = np.zeros(16)
def T(f, n=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3, r
ms, vol = T(lambda: task.decode_fov_device(img, sf, bg)); print('decode_kernel ms %.2f' % ms)
L = _lib.lib()
Z,H,W = vol.shape
labels = torch.empty((Z,H,W), dtype=torch.int32, device=dev); counts = torch.zeros(130, dtype=torch.int64, device=dev)
sp = torch.cuda.current_stream().cuda_stream
ms,_ = T(lambda: L.mf_label_components(vol.decoded.data_ptr(), Z,H,W,0, labels.data_ptr(), counts.data_ptr(), sp)); print('label ms %.2f' % ms, counts.cpu().tolist())
n_pix, n_reg = counts.cpu().tolist()[:2]
B = 16
rec = torch.empty((n_reg, 12), dtype=torch.float64, device=dev); inten = torch.empty((n_reg, B), dtype=torch.float32, device=dev)
wsb = L.mf_region_workspace_bytes(Z,H,W,n_pix,n_reg); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
tr = vol.traces
def tab():
    return L.mf_region_table(vol.decoded.data_ptr(), vol.magnitudes.data_ptr(), vol.distances.data_ptr(), 2, tr.data_ptr(), tr.stride(0), tr.stride(1), labels.data_ptr(), B,Z,H,W,0,n_pix,n_reg, rec.data_ptr(), inten.data_ptr(), None, None, ws.data_ptr(), wsb, sp)
ms,_ = T(tab); print('region_table kernels ms %.2f' % ms, 'ws MB', wsb/1e6)
ms,_ = T(lambda: (rec.cpu(), inten.cpu())); print('D2H rec+inten ms %.2f' % ms)
ms,_ = T(lambda: torch.empty(wsb, dtype=torch.uint8, device=dev)); print('alloc ws ms %.3f' % ms)
def full_table():
    vol._tables.clear(); return task.decoder.region_table(vol, False)
ms, table = T(full_table); print('region_table() python total ms %.2f' % ms)
ms, df = T(lambda: task.decoder.extract_barcodes(vol, 0, 100)); print('extract_barcodes (filter+D2H+DataFrame) ms %.2f' % ms, len(df))
ms, r = T(lambda: task.run_fov(img, sf, bg, 0)); print('run_fov ms %.2f' % ms, len(r[0]))
