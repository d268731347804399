"""Workgroup-shape sweep of conv_split.hip at the encoder's 3x3 shapes (544x960, 2 frames; NCHW tensors through
nnd_conv2d_forward_ex): forced (ny, ks, P) via NND_SPLIT_CFG next to the picker's choice and to the exact fp32 kernel.
    python scripts/sweep_split_encoder.py        (on the GPU box; one subprocess per configuration)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from nndepth_amd import ops
ar = os.environ["AB_ARITH"]
out = []
for (C, H, W) in ((64, 272, 480), (96, 136, 240), (128, 68, 120)):
    torch.manual_seed(0)
    conv = ops.Conv2d(torch.randn(C, C, 3, 3) * 0.05, torch.randn(C) * 0.1, "cuda:0", arithmetic=ar)
    x = torch.randn(2, C, H, W, device="cuda:0")
    try:
        y = conv(x, relu=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = conv(x, relu=True)
        e1.record()
        torch.cuda.synchronize()
        out.append(f"{e0.elapsed_time(e1) * 100:7.1f}")
    except Exception as e:
        out.append("    n/a")
print(" ".join(out), flush=True)
''' % ROOT

if __name__ == "__main__":
    print("columns: 64->64 @272x480x2   96->96 @136x240x2   128->128 @68x120x2   (us per launch, 3x3 + ReLU)")
    ar2 = sys.argv[1] if len(sys.argv) > 1 else "fp16x2"
    cfgs = [("fp32", None), (ar2, None), (ar2, "generic")] + [(ar2, f"{ny},{ks},{p}") for p in (2, 4) for ny in (1, 2, 3, 4) for ks in (1, 2, 4)]
    for ar, cfg in cfgs:
        env = dict(os.environ, AB_ARITH=ar)
        if cfg == "generic":
            env["NND_SPLIT_NO_FAST"] = "1"
        elif cfg:
            env["NND_SPLIT_CFG"] = cfg
        r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.strip() and "amdgpu" not in l]
        print(f"{ar:7s} {cfg or 'picker':8s} {line[-1] if line else 'failed: ' + r.stderr[-300:]}", flush=True)
