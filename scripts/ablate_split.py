"""Ablation of conv_split.hip (timing-only builds under scripts/ablate/, made by scripts/build_ablate.sh): per-conv launch time
of the loop convs (AB_ARITH = fp16x2 | bf16x3) at 68x120 for each build, plus the phase stamps of the STAMPS build.
    python scripts/ablate_split.py            (on the GPU box; spawns one subprocess per library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys, ctypes as C
sys.path.insert(0, %r)
import numpy as np, torch
from nndepth_amd import weightgen
from nndepth_amd.blocks import BasicUpdateBlock
from nndepth_amd._lib import LIB_PATH
H, W = int(os.environ.get("AB_H", 68)), int(os.environ.get("AB_W", 120))
ub = BasicUpdateBlock(hidden_dim=128, cor_planes=36, context_dim=64, flow_channel=1, spatial_scale=8, arithmetic=os.environ.get("AB_ARITH", "fp16x2"))
weightgen.fill_module_(ub, "update_block.")
ub = ub.to("cuda:0"); eng = ub.sync_engine("cuda:0")
ws = eng.workspace(1, H, W, "cuda:0"); ws.normal_()
names = eng.conv_names()
out = []
stamps = os.environ.get("AB_STAMPS") == "1"
raw = C.CDLL(LIB_PATH)
buf = (C.c_ulonglong * (4096 * 8))()
for i, nm in enumerate(names):
    if nm in ("encoder.convc1", "mask.2"):
        continue
    ms, fl = eng.profile_conv(i, 1, H, W, 30, "cuda:0")
    line = f"{nm:32s} {ms*1e3:7.1f} us"
    if stamps:
        eng.profile_conv(i, 1, H, W, 1, "cuda:0")
        torch.cuda.synchronize()
        ns = {"fp16x2": 2, "bf16x3": 3}[os.environ.get("AB_ARITH", "fp16x2")]
        assert getattr(raw, f"nnd_debug_read_split_stamps_ns{ns}")(buf, 4096 * 8) == 0
        full = np.array(buf[:], dtype=np.int64).reshape(4096, 8)
        keep = (full[:, 0] > 0) & (full[:, 4] >= full[:, 0])
        a = full[keep][:, :5]
        clk = full[keep]
        ghz = np.median((clk[:, 6] - clk[:, 5]) / np.maximum(clk[:, 2] - clk[:, 1], 1) * 0.1)  # cycles per 10 ns -> GHz
        us = (a - a[:, 0].min()) / 100.0
        ph = np.diff(us, axis=1)
        line += (f" | WGs {len(a):4d} start spread {us[:,0].max():5.1f} | prologue {ph[:,0].mean():5.1f} K-loop {ph[:,1].mean():6.1f} "
                 f"reduce {ph[:,2].mean():4.1f} epilogue {ph[:,3].mean():5.1f} | last end {us[:,4].max():6.1f} | K-loop clock {ghz:4.2f} GHz")
    print(line, flush=True)
''' % ROOT

if __name__ == "__main__":
    libs = [("product", None)] + [(n[4:-3], os.path.join(ROOT, "scripts/ablate", n)) for n in sorted(os.listdir(os.path.join(ROOT, "scripts/ablate"))) if n.startswith("lib_") and n.endswith(".so")]
    for name, path in libs:
        env = dict(os.environ)
        if path:
            env["NND_LIB"] = path
        env["AB_STAMPS"] = "1" if name.startswith("STAMPS") else "0"
        print(f"==== {name}", flush=True)
        subprocess.run([sys.executable, "-c", WORKER], env=env, check=False)
