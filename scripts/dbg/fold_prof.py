import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time
import torch
from nndepth_amd import ops, weightgen
from nndepth_amd.blocks import BasicUpdateBlock
from nndepth_amd.cost_volume import CorrBlock1D
torch.manual_seed(0)
B, H, W = 1, 68, 120
ub = BasicUpdateBlock(hidden_dim=128, cor_planes=36, context_dim=64, flow_channel=1, spatial_scale=8, arithmetic="fp16x2")
weightgen.fill_module_(ub, "update_block.")
eng = ub.to("cuda:0").eval().sync_engine("cuda:0")
net, inp = torch.tanh(torch.randn(B, 128, H, W)).cuda(), torch.relu(torch.randn(B, 64, H, W)).cuda()
f1, f2 = torch.randn(B, 256, H, W, device="cuda:0"), torch.randn(B, 256, H, W, device="cuda:0")
pyr = CorrBlock1D(f1, f2, 4, 4)._pyr
with ops.calibration():
    eng.refine(pyr, 4, 4, net, inp, 8, 32)
for _ in range(3):
    eng.refine(pyr, 4, 4, net, inp, 8, 32)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.refine(pyr, 4, 4, net, inp, 8, 32)
torch.cuda.synchronize()
print(f"loop of 32 iterations: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms = {(time.perf_counter() - t0) / 20 / 32 * 1e6:.1f} us per iteration")
