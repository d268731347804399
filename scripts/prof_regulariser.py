import sys, time
sys.path.insert(0, "/root/repo")
import torch
from nndepth_amd import weightgen
from nndepth_amd.igev_stereo import CostVolumeFilterNetwork
dev = "cuda:0"
reg = CostVolumeFilterNetwork(8, [40, 80, 160]).to(dev).eval()
weightgen.fill_module_(reg, "igev.cv_regularizer.")
x = torch.randn(1, 8, 240, 136, 240, device=dev)
feats = [torch.rand(1, 40, 68, 120, device=dev), torch.rand(1, 80, 34, 60, device=dev), torch.rand(1, 160, 17, 30, device=dev)]
with torch.no_grad():
    for _ in range(2):
        y = reg(x, feats)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        y = reg(x, feats)
    torch.cuda.synchronize()
    print("regulariser (PyTorch-ROCm) per sample: %.1f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
    # per-layer timing via hooks
    times = {}
    def mk(name):
        def pre(m, i):
            torch.cuda.synchronize(); times[name] = -time.perf_counter()
        def post(m, i, o):
            torch.cuda.synchronize(); times[name] += time.perf_counter()
        return pre, post
    for name, mod in reg.named_modules():
        if isinstance(mod, torch.nn.Conv3d):
            a, b = mk(name)
            mod.register_forward_pre_hook(a); mod.register_forward_hook(b)
    y = reg(x, feats)
    for k, v in times.items():
        print(f"  {k:28s} {v*1e3:7.2f} ms")
    print("  sum of Conv3d: %.1f ms" % (sum(times.values()) * 1e3))
