"""Hot path of Coarse2FineGroupRepViTRAFTStereo (cascade behind the encoder side) at 512x960: stages 8x15, 32x60, 128x240, 12 iterations
each (the class default), batch 1, synthetic stage tensors.     python scripts/bench_c2f.py [arithmetic] [iters]    (on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
from nndepth_amd import ops, weightgen  # noqa: E402
from nndepth_amd.raft_stereo import Coarse2FineRAFTStereoBase  # noqa: E402
from c2f_double import make_c2f  # noqa: E402

if __name__ == "__main__":
    arith = sys.argv[1] if len(sys.argv) > 1 else "fp16x2"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dev = "cuda:0"
    torch.manual_seed(0)
    m = make_c2f(Coarse2FineRAFTStereoBase, iters=iters, corr_levels=1, arithmetic=arith)
    weightgen.fill_module_(m, "c2f.")
    m = m.to(dev).eval()
    B, hw = 1, (512, 960)
    feats = [torch.randn(2 * B, c, h, w, device=dev) for c, h, w in ((256, 8, 15), (64, 32, 60), (64, 128, 240))]
    cnets = [torch.randn(B, 256, h, w, device=dev) for h, w in ((8, 15), (32, 60), (128, 240))]
    with torch.no_grad():
        if arith == "fp16x2":
            with ops.calibration():
                m.refine_stages(feats, cnets, hw)
        for fused in (True, False):
            m.fused_loop = fused
            for _ in range(3):
                m.refine_stages(feats, cnets, hw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                out = m.refine_stages(feats, cnets, hw)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            print(f"{arith} iters {iters} x 3 stages, {'fused loop' if fused else 'seam-by-seam'}: {ms:.2f} ms per pair "
                  f"({len(out)} outputs at {tuple(out[-1]['up_disp'].shape)})", flush=True)
        m.fused_loop = True
        for idx in range(3):  # per stage: the loop alone
            f1, f2 = feats[idx][:B].contiguous(), feats[idx][B:].contiguous()
            net, inp = ops.split_tanh_relu(cnets[idx], 128)
            eng = m.update_block.sync_engine(dev)
            pyr = ops.raft_group_corr_build(f1, f2, 4, 1)
            for _ in range(3):
                eng.refine_group(pyr, 4, 1, 4, net, inp, 4, iters)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                eng.refine_group(pyr, 4, 1, 4, net, inp, 4, iters)
            torch.cuda.synchronize()
            print(f"  stage {idx} {tuple(f1.shape[-2:])}: loop {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms = "
                  f"{(time.perf_counter() - t0) / 10 / iters * 1e6:.1f} us per iteration", flush=True)
