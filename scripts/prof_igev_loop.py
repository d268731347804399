"""IGEV refinement loop (config 3 shape, batch 1, 4 iterations) for rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import ops, weightgen
from nndepth_amd.blocks import BasicUpdateBlock
dev = "cuda:0"
B, G, H, W = 1, 8, 136, 240
fm1, fm2 = torch.randn(B, 128, H, W, device=dev), torch.randn(B, 128, H, W, device=dev)
ub = BasicUpdateBlock(hidden_dim=64, cor_planes=576, context_dim=64, flow_channel=1, spatial_scale=4)
weightgen.fill_module_(ub, "igev.update_block.")
eng = ub.to(dev).sync_engine(dev)
net, inp = torch.tanh(torch.randn(B, 64, H, W, device=dev)), torch.relu(torch.randn(B, 64, H, W, device=dev))
init = -20 * torch.rand(B, 1, H, W, device=dev)
feat = ops.group_corr_build(fm1, fm2, G, G, 4)
geo = ops.pyramid_from_level0(feat[:B * G * H * W * W].clone(), B * G, H, W, 4)
il = ops.igev_interleave_pyramids(feat, geo, B, G, H, W, 4)
for _ in range(2):
    eng.refine_igev(feat, geo, G, 4, 4, net, inp, 4, 4, disp_init=init, keep_all=True)
    eng.refine_igev(feat, geo, G, 4, 4, net, inp, 4, 4, disp_init=init, keep_all=True, interleaved=il)
torch.cuda.synchronize()
