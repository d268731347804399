"""Live timing of the HBM-bound kernels of the hot path against the 8 TB/s roofline.

Used by bench.py (`roofline.hbm_group`) and scripts/prof_hbm.py.  Every kernel is launched through the C-ABI; `achieved` =
ALGORITHMIC bytes (compulsory reads + writes of the call, stated per row) / average launch duration.  The duration is that of
the KERNEL: `reps` launches are captured once into a HIP graph on a side stream and the graph is replayed between two events,
so the 4-8 us kernels are not reported as the ~10 us of ctypes / Python launch overhead per call that timing the calls
themselves measured.  The output allocations of the Python wrappers are part of the capture (graph-private
pool) and cost nothing at replay.
"""
from typing import Callable, Dict, List

import torch

from . import ops

NOT_CAPTURED = []  # calls whose launches could not be captured into a graph (timed as Python calls instead): reported by bench.py
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak (6.3 TB/s is what a float4 copy achieves)


def time_us(fn: Callable[[], object], reps: int = 50, warm: int = 3, replays: int = 3) -> float:
    """Average duration (us) of one launch of `fn`'s kernel(s): `reps` launches captured into a HIP graph, replayed `replays`
    times between two events; the best replay counts (the first one after the capture pays for the graph's upload)."""
    for _ in range(warm):  # also raises the dynamic-LDS limits etc. outside the capture
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for _ in range(reps):
                    fn()
    except Exception as e:  # a wrapper that cannot be captured: time the calls themselves and say so (never silently)
        torch.cuda.synchronize()
        NOT_CAPTURED.append(f"{getattr(fn, '__qualname__', fn)}: {type(e).__name__}: {e}"[:200])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    del graph
    return best


def _row(name: str, us: float, mbytes: float, note: str) -> Dict[str, object]:
    gbs = mbytes / 1e3 / (us * 1e-6)
    return {"kernel": name, "us": round(us, 2), "algorithmic_mb": round(mbytes, 2), "gb_per_s": round(gbs, 1),
            "frac_of_8tbs": round(gbs / HBM_PEAK_GBS, 4), "bytes": note}


def raft_rows(dev, B: int = 1, C: int = 256, H: int = 68, W: int = 120) -> List[Dict[str, object]]:
    """RAFT-Stereo (configs[1]: 544x960 -> 68x120, C = 256): pyramid build, lookup, convex upsample."""
    f = lambda *s: torch.randn(*s, device=dev)
    rows = []
    f1, f2 = f(B, C, H, W), f(B, C, H, W)
    pyr = ops.corr1d_build(f1, f2, 4)
    rows.append(_row(f"corr1d_build {B}x{H}x{W} C={C}", time_us(lambda: ops.corr1d_build(f1, f2, 4)),
                     (2 * f1.numel() + pyr.numel()) * 4 / 1e6, "2 fmaps read + 5 pyramid levels written"))
    coords = torch.arange(W, device=dev).float().view(1, 1, 1, W).repeat(B, 1, H, 1) - 10 * torch.rand(B, 1, H, W, device=dev)
    rows.append(_row(f"corr1d_lookup {B}x{H}x{W} (36 ch)", time_us(lambda: ops.corr1d_lookup(pyr, coords, 4, 4)),
                     (2 * 36 + 36 + 1) * B * H * W * 4 / 1e6, "2 taps x 36 samples read + coords + 36 ch written"))
    flow, mask = f(B, 1, H, W), f(B, 576, H, W)
    rows.append(_row(f"convex_upsample r8 {B}x{H}x{W}", time_us(lambda: ops.convex_upsample(flow, mask, 8)),
                     (576 + 1 + 64) * B * H * W * 4 / 1e6, "576-ch mask + flow read, 64 px/px written"))
    return rows


def igev_rows(dev, B: int = 1, G: int = 8, H: int = 136, W: int = 240) -> List[Dict[str, object]]:
    """IGEV (configs[2], one sample: 544x960 -> 136x240, 8 groups): volume build, pyramids, lookup, init."""
    f = lambda *s: torch.randn(*s, device=dev)
    rows = []
    f1, f2 = f(B, 128, H, W), f(B, 128, H, W)
    fp = ops.group_corr_build(f1, f2, G, G, 4)
    n0 = B * G * H * W * W
    rows.append(_row(f"group_corr_build {H}x{W} G={G}, level 0 (the model's call)", time_us(lambda: ops.group_corr_build(f1, f2, G, G, 4, pooled=False), 12),
                     (2 * B * 64 * H * W + n0) * 4 / 1e6, "64 ch of 2 fmaps read + level 0 written"))
    rows.append(_row(f"group_corr_build {H}x{W} G={G}, 5 levels", time_us(lambda: ops.group_corr_build(f1, f2, G, G, 4), 12),
                     (2 * B * 64 * H * W + fp.numel()) * 4 / 1e6, "64 ch of 2 fmaps read + 5 levels written"))
    gp = fp.clone()
    rows.append(_row(f"pyramid_pool_levels {H}x{W} G={G} (on demand only)", time_us(lambda: ops.pyramid_pool_levels_(gp, B * G, H, W, 4), 12),
                     (2 * fp.numel() - B * G * H * W * W) * 4 / 1e6, "levels 0-3 read, levels 1-4 written"))
    coords = torch.arange(W, device=dev).float().view(1, 1, 1, W).repeat(B, 1, H, 1) - 20 * torch.rand(B, 1, H, W, device=dev)
    rows.append(_row(f"igev_lookup {H}x{W} (576 ch)", time_us(lambda: ops.igev_lookup(fp, gp, coords, G, 4, 4), 20),
                     (3 * 576 + 1) * B * H * W * 4 / 1e6, "2 taps x 576 samples read + 576 ch written"))
    il = ops.igev_interleave_pyramids(fp, gp, B, G, H, W, 4)
    rows.append(_row(f"igev_interleave_level0 {H}x{W} G={G} (the model's call)", time_us(lambda: ops.igev_interleave_level0(fp, gp, B, G, H, W, 4), 12),
                     (2 * n0 + il.numel()) * 4 / 1e6, "level 0 of both volumes read, 4 interleaved levels written"))
    rows.append(_row(f"igev_interleave_pyramids {H}x{W} G={G} (from pooled pyramids)", time_us(lambda: ops.igev_interleave_pyramids(fp, gp, B, G, H, W, 4), 12),
                     2 * il.numel() * 4 / 1e6, "levels 0-3 of both pyramids read + written"))
    conv = torch.nn.Conv3d(G, 1, 3, 1, 1)
    geo0 = gp[:B * G * H * W * W]
    rows.append(_row(f"igev_init_disparity {H}x{W}x{W}", time_us(lambda: ops.igev_init_disparity(geo0, conv.weight, conv.bias, B, G, H, W, W), 12),
                     (geo0.numel() + B * H * W) * 4 / 1e6, "volume read once + disparity written"))
    rows_v = geo0.view(B, G, H, W, W)
    dm = ops.volume_rows_to_depth_major(rows_v)
    rows.append(_row(f"volume_rows_to_depth_major {G}x{H}x{W}x{W}", time_us(lambda: ops.volume_rows_to_depth_major(rows_v), 12),
                     (rows_v.numel() + dm.numel()) * 4 / 1e6, "volume read + written"))
    rows.append(_row(f"depth_major_to_volume_rows {G}x{H}x{W}x{W}", time_us(lambda: ops.depth_major_to_volume_rows(dm), 12),
                     2 * rows_v.numel() * 4 / 1e6, "volume read + written"))
    half = torch.randn(B, W // 2 + 2, 16, H // 2, W // 2, device=dev)
    up = ops.volume_upsample2x(half)
    rows.append(_row(f"volume_upsample2x 16ch {W // 2}x{H // 2}x{W // 2}", time_us(lambda: ops.volume_upsample2x(half), 12),
                     (half.numel() + up.numel()) * 4 / 1e6, "low-res volume read + x8 volume written"))
    return rows


def cre_rows(dev, sizes=((135, 240), (67, 120), (33, 60)), C: int = 256) -> List[Dict[str, object]]:
    """CREStereo (configs[4]: 1080x1920 -> 1/8, 1/16, 1/32): AGCL both modes, 2-channel convex upsample."""
    f = lambda *s: torch.randn(*s, device=dev)
    rows = []
    for (H, W) in sizes:
        B = 1
        f1, f2 = f(B, C, H, W), f(B, C, H, W)
        # a smooth flow field (what the network produces): the lanes of a wave then gather from 1-2 cache lines
        yy, xx = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
        flow = torch.stack([-(6 + 4 * torch.sin(xx / 23) * torch.cos(yy / 17)), 0.7 * torch.sin(xx / 31 + yy / 13)], 0)[None].contiguous()
        off = torch.rand(B, 18, H, W, device=dev) * 2 - 1
        scratch = torch.empty_like(f2)
        f1c, f2c = ops.nchw_to_nhwc(f1), ops.nchw_to_nhwc(f2)  # made once per cascade stage (10 iterations)
        alg = (2 * C + 2 + 36) * H * W * 4 / 1e6
        for sp in (False, True):
            rows.append(_row(f"agcl_corr_iter {H}x{W} small_patch={int(sp)}", time_us(lambda: ops.agcl_corr_iter(f1, f2, flow, sp, scratch)),
                             alg, "2 fmaps + flow read, 36 ch written"))
            rows.append(_row(f"agcl_corr_offset {H}x{W} small_patch={int(sp)}",
                             time_us(lambda: ops.agcl_corr_offset(f1c, f2c, flow, off, sp, channels_last=True)),
                             alg + 18 * H * W * 4 / 1e6, "2 channels-last fmaps + flow + 18 offsets read, 36 ch written"))
        rows.append(_row(f"nchw_to_nhwc {C}ch {H}x{W}", time_us(lambda: ops.nchw_to_nhwc(f1)), 2 * C * H * W * 4 / 1e6, "read + written once"))
        fl2, mask = f(B, 2, H, W), f(B, 576, H, W)
        rows.append(_row(f"convex_upsample r8 2ch {H}x{W}", time_us(lambda: ops.convex_upsample(fl2, mask, 8)),
                         (576 + 2 + 128) * H * W * 4 / 1e6, "576-ch mask + flow read, 2 x 64 px/px written"))
    return rows


def format_rows(rows: List[Dict[str, object]]) -> str:
    return "\n".join(f"{r['kernel']:58s} {r['us']:8.1f} us  {r['algorithmic_mb']:8.1f} MB  {r['gb_per_s']:7.0f} GB/s  "
                     f"{100 * r['frac_of_8tbs']:5.1f} % of 8 TB/s" for r in rows)
