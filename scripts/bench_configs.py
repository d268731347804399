"""Single-GPU timings of the other BASELINE.json configurations' per-GPU work (configs 3 and 4):
   config 4: RAFT-Stereo base, KITTI 375x1242 padded to 384x1248, 8 pairs per GPU, 32 iterations (whole forward)
   config 3: IGEV hot path, 544x960, batch 8: group-wise volume build + pyramids + 32-iteration loop (hidden 64);
             the Conv3d regulariser / backbone are outside the replaced path and not timed here."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import ops, weightgen
from nndepth_amd.blocks import BasicUpdateBlock
from nndepth_amd.raft_stereo import BaseRAFTStereo

dev = "cuda:0"


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


m = BaseRAFTStereo(iters=32, context_dim=64)
weightgen.fill_module_(m)
m = m.to(dev).eval()
f1, f2 = weightgen.synthetic_frames(2, 8, 384, 1248)
f1, f2 = f1.to(dev), f2.to(dev)
dt = timeit(lambda: m(f1, f2))
print(f"config 4 per-GPU work: RAFT-Stereo 8 x 384x1248, 32 iters: {dt * 1e3:.1f} ms / batch = {8 / dt:.1f} pairs/s")
del m, f1, f2

for B in (1, 8):
    G, H, W = 8, 136, 240
    fm1, fm2 = torch.randn(B, 128, H, W, device=dev), torch.randn(B, 128, H, W, device=dev)
    ub = BasicUpdateBlock(hidden_dim=64, cor_planes=576, context_dim=64, flow_channel=1, spatial_scale=4)
    weightgen.fill_module_(ub, "igev.update_block.")
    ub = ub.to(dev)
    eng = ub.sync_engine(dev)
    net, inp = torch.tanh(torch.randn(B, 64, H, W, device=dev)), torch.relu(torch.randn(B, 64, H, W, device=dev))
    init = -20 * torch.rand(B, 1, H, W, device=dev)
    t_build = timeit(lambda: ops.group_corr_build(fm1, fm2, G, G, 4))
    feat = ops.group_corr_build(fm1, fm2, G, G, 4)
    lvl0 = feat[:B * G * H * W * W].clone()
    t_pyr = timeit(lambda: ops.pyramid_from_level0(lvl0, B * G, H, W, 4))
    geo = ops.pyramid_from_level0(lvl0, B * G, H, W, 4)
    t_il = timeit(lambda: ops.igev_interleave_pyramids(feat, geo, B, G, H, W, 4))
    il = ops.igev_interleave_pyramids(feat, geo, B, G, H, W, 4)
    t_loop0 = timeit(lambda: eng.refine_igev(feat, geo, G, 4, 4, net, inp, 4, 32, disp_init=init, keep_all=True), reps=2)
    t_loop = timeit(lambda: eng.refine_igev(feat, geo, G, 4, 4, net, inp, 4, 32, disp_init=init, keep_all=True, interleaved=il), reps=2)
    print(f"config 3 hot path, batch {B}: volume build {t_build * 1e3:.2f} ms, geo pyramid {t_pyr * 1e3:.2f} ms, interleave {t_il * 1e3:.2f} ms, "
          f"32-iteration loop {t_loop * 1e3:.1f} ms ({t_loop / 32 / B * 1e6:.0f} us per iteration and sample; "
          f"{t_loop0 / 32 / B * 1e6:.0f} us gathering from the reference-layout pyramids)")
    del feat, geo, lvl0, il
