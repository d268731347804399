"""Per-operator time of the IGEV cost-volume regulariser's HIP path at the 544x960 shape (1 x 8 x 240 x 136 x 240):
every Conv3d launch, trilinear upsample, gate and layout conversion, with its algorithmic GFLOP / MB.
    python scripts/prof_regulariser_layers.py        (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nndepth_amd import ops, weightgen  # noqa: E402
from nndepth_amd.igev_stereo import CostVolumeFilterNetwork  # noqa: E402

dev = "cuda:0"
reg = CostVolumeFilterNetwork(8, [40, 80, 160]).to(dev).eval()
reg.arithmetic = sys.argv[1] if len(sys.argv) > 1 else "fp32"
weightgen.fill_module_(reg, "igev.cv_regularizer.")
e = reg._engines(dev)
rows = torch.randn(1, 8, 136, 240, 240, device=dev)
feats = [torch.rand(1, 40, 68, 120, device=dev), torch.rand(1, 80, 34, 60, device=dev), torch.rand(1, 160, 17, 30, device=dev)]
table = []


def timed(name, fn, gflop=0.0, mb=0.0, reps=5):
    out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    table.append((name, us, gflop, mb))
    return out


def conv(name, eng, x0, x1=None):
    N, Dp, c0, H, W = x0.shape
    st, co = eng.desc.stride, eng.desc.Cout
    ci = eng.desc.Cin0 + eng.desc.Cin1
    D = Dp - 2
    Do, Ho, Wo = (D + st - 1) // st, (H + st - 1) // st, (W + st - 1) // st
    gf = 2.0 * Do * Ho * Wo * co * ci * 27 / 1e9
    mb = (x0.numel() + (x1.numel() if x1 is not None else 0) + (Do + 2) * co * Ho * Wo) * 4 / 1e6
    return timed(f"{name}: Conv3d {ci}->{co} s{st} @{Do}x{Ho}x{Wo}", lambda: eng(x0, x1), gf, mb)


def gate(name, vol, g, feat):
    lg = timed(name + " logits (2 x 1x1 conv)", lambda: g[1](g[0](feat, relu=True)))
    return timed(name + " gate", lambda: ops.volume_gate_(vol, lg), 0.0, 2 * vol.numel() * 4 / 1e6)


with torch.no_grad():
    x0 = timed("rows -> depth-major", lambda: ops.volume_rows_to_depth_major(rows), 0, 2 * rows.numel() * 4 / 1e6)
    c1 = conv("conv1.1", e["conv1"][1], conv("conv1.0", e["conv1"][0], x0))
    c1 = gate("conv1", c1, e["g1"], feats[0])
    c2 = conv("conv2.1", e["conv2"][1], conv("conv2.0", e["conv2"][0], c1))
    c2 = gate("conv2", c2, e["g2"], feats[1])
    c3 = conv("conv3.1", e["conv3"][1], conv("conv3.0", e["conv3"][0], c2))
    c3 = gate("conv3", c3, e["g3"], feats[2])
    u3 = timed("upsample c3", lambda: ops.volume_upsample2x(c3), 0, 9 * c3.numel() * 4 / 1e6)
    c2 = conv("proj_3", e["proj_3"], conv("conv3_up", e["conv3_up"], u3), c2)
    c2 = gate("conv3_up", c2, e["g3u"], feats[1])
    u2 = timed("upsample c2", lambda: ops.volume_upsample2x(c2), 0, 9 * c2.numel() * 4 / 1e6)
    c1 = conv("proj_2", e["proj_2"], conv("conv2_up", e["conv2_up"], u2), c1)
    c1 = gate("conv2_up", c1, e["g2u"], feats[0])
    u1 = timed("upsample c1", lambda: ops.volume_upsample2x(c1), 0, 9 * c1.numel() * 4 / 1e6)
    y = conv("final_conv", e["final"], conv("conv1_up", e["conv1_up"], u1))
    out = timed("depth-major -> rows", lambda: ops.depth_major_to_volume_rows(y), 0, 2 * rows.numel() * 4 / 1e6)
tot = sum(t[1] for t in table)
print(f"{'operator':52s} {'us':>8s} {'GFLOP':>7s} {'TF/s':>6s} {'MB':>8s} {'TB/s':>6s}  share")
for name, us, gf, mb in table:
    print(f"{name:52s} {us:8.1f} {gf:7.2f} {gf / us * 1e3 if gf else 0:6.1f} {mb:8.1f} {mb / us:6.2f}  {100 * us / tot:4.1f} %")
print(f"{'sum':52s} {tot:8.1f} {sum(t[2] for t in table):7.2f}")
