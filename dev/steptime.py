"""Where does a bench step spend its time (development aid)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.decode import Decode
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = bench.load_codebook()
task = Decode(cb, {'crop_width': 100}, device=dev)
fovs = [synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234 + i, dev) for i in range(2)]
sf = bench.initial_scale_factors(fovs[0]); bg = np.zeros(16)
def sync(): torch.cuda.synchronize()
for i in range(8):
    sync(); t0 = time.perf_counter()
    vol = task.decode_fov_device(fovs[i % 2], sf, bg); sync(); t1 = time.perf_counter()
    table = task.decoder.region_table(vol, False); sync(); t2 = time.perf_counter()
    bc = task.decoder.extract_barcodes(vol, i, 100); sync(); t3 = time.perf_counter()
    print('step %d decode %.2f table %.2f frame %.2f total %.2f ms  mem %.1f GB' % (i, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t3-t0)*1e3, torch.cuda.memory_reserved()/1e9))
print('--- run_fov, result kept in tuple')
for i in range(6):
    sync(); t0 = time.perf_counter()
    bc, vol = task.run_fov(fovs[i % 2], sf, bg, fov=i); sync()
    print('run_fov %.2f ms' % ((time.perf_counter()-t0)*1e3))
print('--- run_fov, freed first')
for i in range(6):
    bc = vol = None
    sync(); t0 = time.perf_counter()
    bc, vol = task.run_fov(fovs[i % 2], sf, bg, fov=i); sync()
    print('run_fov %.2f ms' % ((time.perf_counter()-t0)*1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(3):
    bc, vol = task.run_fov(fovs[i % 2], sf, bg, fov=i); sync()
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
