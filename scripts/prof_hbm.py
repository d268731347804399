"""HBM-bound kernels of the hot path at their BASELINE.json config shapes: time per call (torch events on the launch
stream = torch's current stream) and achieved GB/s over the ALGORITHMIC bytes (compulsory reads + writes).
    python scripts/prof_hbm.py            (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import ops

DEV = "cuda:0"
PEAK = 8000.0  # GB/s


def timeit(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def line(name, us, mbytes):
    gbs = mbytes / 1e3 / (us * 1e-6)
    print(f"{name:58s} {us:8.1f} us  {mbytes:8.1f} MB  {gbs:7.0f} GB/s  {100 * gbs / PEAK:5.1f} % of 8 TB/s")


def main():
    torch.manual_seed(0)
    f = lambda *s: torch.randn(*s, device=DEV)
    # RAFT-Stereo @544x960 (config 2): 68x120, C=256
    B, C, H, W = 1, 256, 68, 120
    f1, f2 = f(B, C, H, W), f(B, C, H, W)
    pyr = ops.corr1d_build(f1, f2, 4)
    line("corr1d_build 68x120 C=256", timeit(lambda: ops.corr1d_build(f1, f2, 4)), (2 * f1.numel() + pyr.numel()) * 4 / 1e6)
    coords = torch.arange(W, device=DEV).float().view(1, 1, 1, W).repeat(B, 1, H, 1) - 10 * torch.rand(B, 1, H, W, device=DEV)
    line("corr1d_lookup 68x120 (36 ch)", timeit(lambda: ops.corr1d_lookup(pyr, coords, 4, 4)), (2 * 36 + 36 + 1) * H * W * 4 / 1e6)
    flow, mask = f(B, 1, H, W), f(B, 576, H, W)
    line("convex_upsample r8 68x120 (mask read)", timeit(lambda: ops.convex_upsample(flow, mask, 8)), (576 + 1 + 64) * H * W * 4 / 1e6)
    # KITTI batch 8 per GPU (config 4): 48x156
    B, H, W = 8, 48, 156
    f1, f2 = f(B, C, H, W), f(B, C, H, W)
    pyr = ops.corr1d_build(f1, f2, 4)
    line("corr1d_build 8x48x156 C=256", timeit(lambda: ops.corr1d_build(f1, f2, 4)), (2 * f1.numel() + pyr.numel()) * 4 / 1e6)
    coords = torch.arange(W, device=DEV).float().view(1, 1, 1, W).repeat(B, 1, H, 1) - 10 * torch.rand(B, 1, H, W, device=DEV)
    line("corr1d_lookup 8x48x156", timeit(lambda: ops.corr1d_lookup(pyr, coords, 4, 4)), (2 * 36 + 36 + 1) * B * H * W * 4 / 1e6)
    # IGEV @544x960 (config 3, one sample): 136x240, 8 groups
    B, G, H, W = 1, 8, 136, 240
    f1, f2 = f(B, 128, H, W), f(B, 128, H, W)
    fp = ops.group_corr_build(f1, f2, G, G, 4)
    line("group_corr_build 136x240 G=8", timeit(lambda: ops.group_corr_build(f1, f2, G, G, 4), 10), (2 * B * 64 * H * W + fp.numel()) * 4 / 1e6)
    lvl0 = fp[:B * G * H * W * W].clone()
    gp = ops.pyramid_from_level0(lvl0, B * G, H, W, 4)
    line("pyramid_from_level0 136x240 G=8 (incl. level-0 copy)", timeit(lambda: ops.pyramid_from_level0(lvl0, B * G, H, W, 4), 10), (lvl0.numel() + fp.numel()) * 4 / 1e6)
    coords = torch.arange(W, device=DEV).float().view(1, 1, 1, W).repeat(B, 1, H, 1) - 20 * torch.rand(B, 1, H, W, device=DEV)
    line("igev_lookup 136x240 (576 ch)", timeit(lambda: ops.igev_lookup(fp, gp, coords, G, 4, 4), 20), (3 * 576 + 1) * H * W * 4 / 1e6)
    il = ops.igev_interleave_pyramids(fp, gp, B, G, H, W, 4)
    line("igev_interleave_pyramids 136x240 G=8 (levels 0-3)", timeit(lambda: ops.igev_interleave_pyramids(fp, gp, B, G, H, W, 4), 10), 2 * il.numel() * 4 / 1e6)
    conv = torch.nn.Conv3d(G, 1, 3, 1, 1)
    geo0 = gp[:B * G * H * W * W]
    line("igev_init_disparity 136x240x240 (squeezer + soft-argmin)", timeit(lambda: ops.igev_init_disparity(geo0, conv.weight, conv.bias, B, G, H, W, W), 10),
         (geo0.numel() + B * H * W) * 4 / 1e6)
    rows = geo0.view(B, G, H, W, W)
    dm = ops.volume_rows_to_depth_major(rows)
    line("volume_rows_to_depth_major 8x136x240x240", timeit(lambda: ops.volume_rows_to_depth_major(rows), 10), (rows.numel() + dm.numel()) * 4 / 1e6)
    line("depth_major_to_volume_rows 8x136x240x240", timeit(lambda: ops.depth_major_to_volume_rows(dm), 10), 2 * rows.numel() * 4 / 1e6)
    half = torch.randn(B, 122, 16, H // 2, W // 2, device=DEV)
    up = ops.volume_upsample2x(half)
    line("volume_upsample2x 16ch 120x68x120 -> 240x136x240", timeit(lambda: ops.volume_upsample2x(half), 10), (half.numel() + up.numel()) * 4 / 1e6)
    del fp, gp, il, dm, half, up
    # CREStereo @1080x1920 (config 5): 1/8 135x240, 1/16 67x120, 1/32 33x60, C=256
    for (H, W) in ((135, 240), (67, 120), (33, 60)):
        B, C = 1, 256
        f1, f2 = f(B, C, H, W), f(B, C, H, W)
        # a smooth flow field (what the network produces); lanes of a wave then gather from 1-2 cache lines
        yy, xx = torch.meshgrid(torch.arange(H, device=DEV).float(), torch.arange(W, device=DEV).float(), indexing="ij")
        flow = torch.stack([-(6 + 4 * torch.sin(xx / 23) * torch.cos(yy / 17)), 0.7 * torch.sin(xx / 31 + yy / 13)], 0)[None].contiguous()
        off = torch.rand(B, 18, H, W, device=DEV) * 2 - 1
        scratch = torch.empty_like(f2)
        alg = (2 * C + 2 + 36) * H * W * 4 / 1e6
        for sp in (False, True):
            line(f"agcl_corr_iter {H}x{W} small_patch={int(sp)}", timeit(lambda: ops.agcl_corr_iter(f1, f2, flow, sp, scratch)), alg)
            line(f"agcl_corr_offset {H}x{W} small_patch={int(sp)}", timeit(lambda: ops.agcl_corr_offset(f1, f2, flow, off, sp)), alg + 18 * H * W * 4 / 1e6)
        fl2, mask = f(B, 2, H, W), f(B, 576, H, W)
        line(f"convex_upsample r8 2ch {H}x{W}", timeit(lambda: ops.convex_upsample(fl2, mask, 8)), (576 + 2 + 128) * H * W * 4 / 1e6)


if __name__ == "__main__":
    main()
