"""Workgroup-shape sweep of conv_split: launch time of every loop conv in a split arithmetic for forced (ny, ks, P) = (output-
channel groups across workgroups, intra-workgroup split-K, sub-tiles per wave) via NND_SPLIT_CFG, next to the picker's own
choice and to the generic (pre-round-3) kernel (NND_SPLIT_NO_FAST).
    python scripts/sweep_split.py H W [B] [arithmetic] [raft|igev]       (on the GPU box; one subprocess per configuration)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from nndepth_amd import weightgen
from nndepth_amd.blocks import BasicUpdateBlock
H, W, B = int(os.environ["AB_H"]), int(os.environ["AB_W"]), int(os.environ["AB_B"])
igev = os.environ.get("AB_MODEL", "raft") == "igev"  # IGEV's update block: hidden 64, 576 correlation planes, 1/4 resolution
ub = BasicUpdateBlock(hidden_dim=64 if igev else 128, cor_planes=576 if igev else 36, context_dim=64, flow_channel=1,
                      spatial_scale=4 if igev else 8, arithmetic=os.environ.get("AB_ARITH", "fp16x2"))
weightgen.fill_module_(ub, "update_block.")
ub = ub.to("cuda:0"); eng = ub.sync_engine("cuda:0")
ws = eng.workspace(B, H, W, "cuda:0"); ws.normal_()
out = []
for i, nm in enumerate(eng.conv_names()):
    if nm in ("encoder.convc1", "mask.2"):
        continue
    try:
        ms, fl = eng.profile_conv(i, B, H, W, 20, "cuda:0")
        out.append(f"{ms*1e3:7.1f}")
    except Exception as e:
        out.append("    n/a")
print(" ".join(out), flush=True)
''' % ROOT

if __name__ == "__main__":
    H, W = int(sys.argv[1]), int(sys.argv[2])
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    arith = sys.argv[4] if len(sys.argv) > 4 else "fp16x2"
    model = sys.argv[5] if len(sys.argv) > 5 else "raft"
    print(f"{H}x{W} batch {B} {arith} {model} update block; columns: convc2 convf2 conv zr1 q1 zr2 q2 fhm   (us per launch); rows: ny,ks,P")
    cfgs = [None, "generic"] + [f"{ny},{ks},2" for ny in (1, 2, 3, 4, 6, 8, 16) for ks in (1, 2, 4)]  # P = 3 / 4 are not instantiated
    for cfg in cfgs:
        env = dict(os.environ, AB_H=str(H), AB_W=str(W), AB_B=str(B), AB_ARITH=arith, AB_MODEL=model)
        if cfg == "generic":
            env["NND_SPLIT_NO_FAST"] = "1"
        elif cfg:
            env["NND_SPLIT_CFG"] = cfg
        r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.strip() and "amdgpu" not in l]
        print(f"{cfg or 'picker':8s} {line[-1] if line else 'failed: ' + r.stderr[-200:]}", flush=True)
