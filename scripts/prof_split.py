"""Stand-alone launch time of every conv of the RAFT-Stereo loop at 68x120 in the three arithmetics (exact fp32 MFMA, bf16x3 and
fp16x2 split MFMA), hipEvents on the launch stream (nnd_profile_conv), then the whole 544x960 / 32-iteration forward.
    python scripts/prof_split.py [H W]          (on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nndepth_amd import weightgen  # noqa: E402
from nndepth_amd.raft_stereo import BaseRAFTStereo  # noqa: E402

dev = "cuda:0"
Hf, Wf = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (68, 120)
ARITHS = ("fp32", "bf16x3", "fp16x2")
models = {}
for ar in ARITHS:
    m = BaseRAFTStereo(iters=32, context_dim=64, arithmetic=ar)
    weightgen.fill_module_(m)
    models[ar] = m.to(dev).eval()
rows = {}
for ar, m in models.items():
    eng = m.update_block.sync_engine(dev)
    for i, nm in enumerate(eng.conv_names()):
        ms, fl = eng.profile_conv(i, 1, Hf, Wf, 30, dev)
        rows.setdefault(nm, {})[ar] = (ms * 1e3, fl / 1e9)
print(f"{'conv':34s} {'GFLOP':>7s} " + " ".join(f"{ar + ' us':>10s} {'TF(alg)':>8s}" for ar in ARITHS))
tot = {ar: 0.0 for ar in ARITHS}
for nm, r in rows.items():
    for ar in ARITHS:
        tot[ar] += r[ar][0]
    print(f"{nm:34s} {r['fp32'][1]:7.2f} " + " ".join(f"{r[ar][0]:10.1f} {r[ar][1] / r[ar][0] * 1e3:8.1f}" for ar in ARITHS))
print(f"{'sum':34s} {'':7s} " + " ".join(f"{tot[ar]:10.1f} {'':8s}" for ar in ARITHS))
f1, f2 = (x.to(dev) for x in weightgen.synthetic_frames(100, 1, Hf * 8, Wf * 8))
outs = {}
for ar, m in models.items():
    for _ in range(3):
        m(f1, f2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = m(f1, f2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    outs[ar] = out[-1]["up_disp"]
    print(f"forward {Hf * 8}x{Wf * 8} / 32 iters, {ar}: {dt * 1e3:.2f} ms = {1 / dt:.1f} pairs/s")
for ar in ARITHS[1:]:
    print(f"max-abs up_disp {ar} vs fp32 path: {(outs['fp32'] - outs[ar]).abs().max().item():.2e}")
