"""Per-GPU work of BASELINE.json configs 3, 4 and 5 on one MI355X, per arithmetic (exact fp32 MFMA / bf16x3 / fp16x2 split):
    config 3  IGEV hot path at 544x960 (136x240 at 1/4): volume, regulariser, init, 32-iteration loop, batch 1 and 8
    config 4  RAFT-Stereo, 8 x 384x1248 (KITTI padded), 32 iterations
    config 5  CREStereo 1080x1920, 20 iterations: single cascade and the 2-stage harness
    widening  Coarse2Fine RAFT-Stereo 512x960, 3 stages x 12 iterations (the class defaults), test double's encoder side
    python scripts/bench_configs.py [fp32,bf16x3,fp16x2]           (on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
from nndepth_amd import weightgen  # noqa: E402

dev = "cuda:0"


def timeit(fn, reps, warm=3):
    """Median wall time of `reps` synchronised calls, ms (a one-off allocator hiccup in one rep does not move it)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    timeit.best = ts[0]
    return ts[len(ts) // 2]


def main():
    from igev_double import make_igev
    from nndepth_amd.cre_stereo import CREStereoBase, two_stage_forward
    from nndepth_amd.igev_stereo import CostVolumeFilterNetwork, IGEVStereoBase
    from nndepth_amd.raft_stereo import BaseRAFTStereo
    ar, which = sys.argv[1], sys.argv[2]
    if which == "kitti":
        m = BaseRAFTStereo(iters=32, context_dim=64, arithmetic=ar)
        weightgen.fill_module_(m)
        m = m.to(dev).eval()
        f1, f2 = (x.to(dev) for x in weightgen.synthetic_frames(2, 8, 384, 1248))
        ms = timeit(lambda: m(f1, f2), 7)
        print(f"config 4 per-GPU work [{ar}]: RAFT-Stereo 8 x 384x1248, 32 iters: {ms:.1f} ms / batch = {8e3 / ms:.1f} pairs/s", flush=True)
    elif which == "cre":
        m = CREStereoBase(iters=20, arithmetic=ar)
        weightgen.fill_module_(m)
        m = m.to(dev).eval()
        f1, f2 = (x.to(dev) for x in weightgen.synthetic_frames(3, 1, 1080, 1920))
        ms = timeit(lambda: m(f1, f2), 9)
        best = timeit.best
        ms2 = timeit(lambda: two_stage_forward(m, f1, f2), 9)
        print(f"config 5 per-GPU work [{ar}]: CREStereo 1080x1920, 20 iters: cascade {ms:.1f} ms / pair (median of 9; best {best:.1f}), 2-stage harness {ms2:.1f} ms / pair", flush=True)
    elif which == "c2f":
        from c2f_double import make_c2f
        from nndepth_amd.raft_stereo import Coarse2FineRAFTStereoBase
        m = make_c2f(Coarse2FineRAFTStereoBase, corr_levels=1, arithmetic=ar)
        weightgen.fill_module_(m, "c2f.")
        m = m.to(dev).eval()
        f1, f2 = (x.to(dev) for x in weightgen.synthetic_frames(5, 1, 512, 960))
        ms = timeit(lambda: m(f1, f2), 9)
        print(f"widening [{ar}]: Coarse2Fine RAFT-Stereo 512x960, 3 x 12 iters (tiny encoder side): {ms:.1f} ms / pair = {1e3 / ms:.1f} pairs/s", flush=True)
    else:
        B = int(which[4:])
        m = make_igev(IGEVStereoBase, CostVolumeFilterNetwork, iters=32, hidden_dim=64, context_dim=64, arithmetic=ar)
        weightgen.fill_module_(m, "igev.")
        m = m.to(dev).eval()
        f1, f2 = (x.to(dev) for x in weightgen.synthetic_frames(4, B, 544, 960))
        ms = timeit(lambda: m(f1, f2), 5, warm=2)
        print(f"config 3 [{ar}]: IGEV 544x960 batch {B}, 32 iters (tiny backbone): {ms:.1f} ms / batch = {B * 1e3 / ms:.2f} pairs/s; "
              f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2:
        main()
    else:  # one process per (arithmetic, configuration): every measurement starts from a fresh allocator and library state
        import subprocess
        for ar in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("fp32", "bf16x3", "fp16x2")):
            for which in ("kitti", "cre", "igev1", "igev8", "c2f"):
                subprocess.run([sys.executable, os.path.abspath(__file__), ar, which], check=False)
