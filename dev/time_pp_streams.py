"""Preprocess of one FOV's 112 frames: one stream vs two concurrent streams."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from merlin_amd.preprocess import DeconvolutionPreprocess
from merlin_amd.synthetic import synthetic_fov
dev = torch.device('cuda:0')
cb = bench.load_codebook()
fov = synthetic_fov(cb.get_barcodes(), 7, 2048, 2048, 40000, 1234, dev)
frames = (2000 + fov.float() + 10 * torch.randn(fov.shape, device=dev)).clamp_(0, 65535).to(torch.uint16)
frames = frames.reshape(-1, 2048, 2048)
pre = DeconvolutionPreprocess(device=dev)
streams = [torch.cuda.Stream(dev) for _ in range(4)]

def one():
    return [pre.preprocess_device(frames)[0]]

def split(k):
    parts = frames.chunk(k)
    cur = torch.cuda.current_stream(dev)
    outs = []
    for s, p in zip(streams, parts):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(pre.preprocess_device(p)[0])
    for s in streams[:k]:
        cur.wait_stream(s)
    return outs

ref = torch.cat(one())
for name, f in (('one stream', one), ('two streams', lambda: split(2)), ('four streams', lambda: split(4)),
                ('one stream', one)):
    out = torch.cat(f()); torch.cuda.synchronize()
    ok = torch.equal(out.view(torch.int16), ref.view(torch.int16))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): f()
    e1.record(); torch.cuda.synchronize()
    print(name, 'ms/FOV %.2f' % (e0.elapsed_time(e1) / 3), 'same output', ok)
