"""Cost of the pixel histogram inside the preprocess stage on noisy frames
(many non-zero output pixels) and on the bench frames (few)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'dev'))
import bench
from merlin_amd.preprocess import DeconvolutionPreprocess
from merlin_amd.synthetic import synthetic_fov
from bench_preprocess import camera_frames
dev = torch.device('cuda:0')
cb = bench.load_codebook()
fov = synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234, dev)
noisy = (2000 + fov.float() + 10 * torch.randn(fov.shape, device=dev)).clamp_(0, 65535).to(torch.uint16)
noisy = noisy.reshape(-1, 2048, 2048)
clean = camera_frames(112, 2048, 2048)
pre = DeconvolutionPreprocess(device=dev)
for name, frames in (('noisy', noisy), ('bench', clean)):
    for hist in (True, False):
        out, _ = pre.preprocess_device(frames, histogram=hist); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2): pre.preprocess_device(frames, histogram=hist)
        e1.record(); torch.cuda.synchronize()
        print(name, 'histogram', hist, 'ms/frame %.4f' % (e0.elapsed_time(e1) / 2 / 112),
              'nonzero %.3f' % float((out.view(torch.int16) != 0).float().mean()))
