"""Stand-alone hipEvent timing of every conv launch of the IGEV update block (hidden 64, 576 lookup planes) at 136x240."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen
from nndepth_amd.blocks import BasicUpdateBlock
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ub = BasicUpdateBlock(hidden_dim=64, cor_planes=576, context_dim=64, flow_channel=1, spatial_scale=4)
weightgen.fill_module_(ub, "igev.update_block.")
eng = ub.to("cuda:0").sync_engine("cuda:0")
ws = eng.workspace(B, 136, 240, "cuda:0")
ws.normal_()
tot_ms = tot_fl = 0.0
for i, nm in enumerate(eng.conv_names()):
    ms, fl = eng.profile_conv(i, B, 136, 240, 10, "cuda:0")
    tot_ms += ms; tot_fl += fl
    print(f"{i:2d} {nm:30s} {ms*1e3:8.1f} us {fl/1e9:7.2f} GF {fl/ms/1e9:6.1f} TF")
print(f"all: {tot_ms*1e3:.0f} us, {tot_fl/1e9:.1f} GF, {tot_fl/tot_ms/1e9:.1f} TF")
