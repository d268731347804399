"""CREStereo (BASELINE.json config 5 shape: 1080x1920, 20 iterations, 1 pair per GPU) end-to-end forward time: direct launches
and HIP-graph replay (nndepth_amd/graph.py), optionally with the process restricted to a few host cores BEFORE anything touches
the GPU (the path must not depend on a fast, idle host).
    python scripts/bench_cre.py [H W iters reps cores]"""
import os, sys, time
args = sys.argv[1:6] + ["1080", "1920", "20", "9", "0"][len(sys.argv) - 1:]
H, W, iters, reps, cores = (int(a) for a in args)
if cores > 0:
    os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:cores]))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen
from nndepth_amd.cre_stereo import CREStereoBase, two_stage_forward
from nndepth_amd.graph import GraphedForward

dev = "cuda:0"
m = CREStereoBase(iters=iters)
weightgen.fill_module_(m)
m = m.to(dev).eval()
f1, f2 = weightgen.synthetic_frames(3, 1, H, W)
f1, f2 = f1.to(dev), f2.to(dev)


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1], out


print(f"host cores available to this process: {len(os.sched_getaffinity(0))}")
med, best, worst, out = timeit(lambda: m(f1, f2), reps)
print(f"CREStereo {H}x{W} iters={iters}, direct launches: median {med:.1f} ms / pair (best {best:.1f}, worst {worst:.1f}; {1e3 / med:.2f} pairs/s), "
      f"outputs {len(out)}, |flow|max {out[-1]['up_disp'].abs().max().item():.1f}")
# host time alone: enqueue without waiting for the GPU
torch.cuda.synchronize()
t0 = time.perf_counter()
out = m(f1, f2)
t_host = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
print(f"  host-side enqueue time of one pair (no wait): {t_host:.1f} ms")
fwd = GraphedForward(m)
med, best, worst, out_g = timeit(lambda: fwd(f1, f2), reps)
print(f"CREStereo {H}x{W} iters={iters}, HIP-graph replay: median {med:.1f} ms / pair (best {best:.1f}, worst {worst:.1f}; {1e3 / med:.2f} pairs/s), "
      f"equal to direct: {all(torch.equal(a['up_disp'], b['up_disp']) for a, b in zip(out, out_g))}")
med, best, worst, _ = timeit(lambda: two_stage_forward(m, f1, f2), reps)
print(f"CREStereo 2-stage (config 5 harness) {H}x{W} iters={iters}, direct launches: median {med:.1f} ms / pair (best {best:.1f})")
