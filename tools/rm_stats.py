import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import fennec_amd
from fennec_amd import synth
ctx = fennec_amd.Context(0)
for (W,H,DW,DH) in [(3840,2160,1920,1080),(1920,1080,3840,2160),(640,480,320,240)]:
    img = ctx.GaussianBlur(ctx.GaussianBlur(torch.from_numpy(synth.noise_image(W, H, 5)).cuda(), 2.0), 1.2)
    ctx.lanczosResize(img, DW, DH); ctx.sync()
    n = synth.noise_image(W,H,3); n[...,3]=255
    ctx.lanczosResize(torch.from_numpy(n).cuda(), DW, DH); ctx.sync()
