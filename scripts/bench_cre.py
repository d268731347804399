"""CREStereo (BASELINE.json config 5 shape: 1080x1920, 20 iterations, 1 pair per GPU) end-to-end forward time.
    python scripts/bench_cre.py [H W iters reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen
from nndepth_amd.cre_stereo import CREStereoBase, two_stage_forward

H, W, iters, reps = (int(a) for a in (sys.argv[1:5] + ["1080", "1920", "20", "3"][len(sys.argv) - 1:]))
dev = "cuda:0"
m = CREStereoBase(iters=iters)
weightgen.fill_module_(m)
m = m.to(dev).eval()
f1, f2 = weightgen.synthetic_frames(3, 1, H, W)
f1, f2 = f1.to(dev), f2.to(dev)
for _ in range(2):
    out = m(f1, f2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = m(f1, f2)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"CREStereo {H}x{W} iters={iters}: {dt * 1e3:.1f} ms / pair  ({1 / dt:.2f} pairs/s), outputs {len(out)}, |flow|max {out[-1]['up_disp'].abs().max().item():.1f}")
for _ in range(2):
    out = two_stage_forward(m, f1, f2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = two_stage_forward(m, f1, f2)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"CREStereo 2-stage (config 5 harness) {H}x{W} iters={iters}: {dt * 1e3:.1f} ms / pair  ({1 / dt:.2f} pairs/s)")
