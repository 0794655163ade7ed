import sys, importlib, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_parity import _scene
from oracle import raster_ref
ops = importlib.import_module('3danimals_amd.ops')
dev = torch.device('cuda:0')
for B,H,W in [(1,64,64),(2,50,70),(4,256,256)]:
    _, faces, clip, _ = _scene(B)
    ref = raster_ref.rasterize(clip, faces.int(), (H, W))
    out = ops.rasterize(clip.to(dev), faces.to(dev), (H, W)).cpu()
    bad = (out[...,3] != ref[...,3])
    print(B,H,W,'id mismatches', int(bad.sum()), 'of', bad.numel(), 'covered ref', int((ref[...,3]>0).sum()), 'out', int((out[...,3]>0).sum()))
    idx = bad.nonzero()[:8]
    for b,y,x in idx.tolist():
        print('  at', b,y,x,'out', out[b,y,x].tolist(), 'ref', ref[b,y,x].tolist())
