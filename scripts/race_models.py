"""Reproducibility of the IGEV and CREStereo forwards (and of one depth-marching Conv3d layer) beside another stream's fp16x2
RAFT-Stereo encoder / full forward (aggressor thread): every victim forward is compared bit for bit with its undisturbed result.
    python scripts/race_models.py reps"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from igev_double import make_igev
from nndepth_amd import weightgen, ops
from nndepth_amd.raft_stereo import BaseRAFTStereo
from nndepth_amd.cre_stereo import CREStereoBase
from nndepth_amd.igev_stereo import IGEVStereoBase, CostVolumeFilterNetwork
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
am = BaseRAFTStereo(iters=4, context_dim=64, arithmetic="fp16x2")
weightgen.fill_module_(am)
am = am.to(DEV).eval()
afr = tuple(f.to(DEV) for f in weightgen.synthetic_frames(21, 1, 128, 160))
am(*afr)


def victims():
    m = make_igev(IGEVStereoBase, CostVolumeFilterNetwork, iters=4, hidden_dim=64, context_dim=64, arithmetic="fp16x2")
    weightgen.fill_module_(m, "igev.")
    m = m.to(DEV).eval()
    f = tuple(x.to(DEV) for x in weightgen.synthetic_frames(6, 1, 128, 192))
    yield "IGEV 128x192, 4 iterations", lambda: [o["up_disp"] for o in m(*f)]
    c = CREStereoBase(iters=2, arithmetic="fp16x2")
    weightgen.fill_module_(c)
    c = c.to(DEV).eval()
    g = tuple(x.to(DEV) for x in weightgen.synthetic_frames(3, 1, 256, 320))
    yield "CREStereo 256x320, iters=2", lambda: [o["up_disp"] for o in c(*g)]
    torch.manual_seed(0)
    w = torch.randn(8, 16, 3, 3, 3) * 0.05
    conv = ops.Conv3dNorm(w, None, 1, None, 1e-5, 0.01, 0, DEV, arithmetic="fp16x2")
    x = torch.randn(1, 42, 16, 36, 80, device=DEV)
    x[:, 0] = 0
    x[:, -1] = 0
    yield "Conv3d 16->8 (depth-marching kernel) 40x36x80", lambda: [conv(x)]
    w2 = torch.randn(16, 8, 3, 3, 3) * 0.05
    conv2 = ops.Conv3dNorm(w2, None, 2, None, 1e-5, 0.01, 0, DEV, arithmetic="fp16x2")
    x2 = torch.randn(1, 42, 8, 36, 80, device=DEV)
    x2[:, 0] = 0
    x2[:, -1] = 0
    yield "Conv3d 8->16 stride 2 (depth-marching kernel) 40x36x80", lambda: [conv2(x2)]


for name, fn in victims():
    with torch.no_grad():
        base = [o.clone() for o in fn()]
    torch.cuda.synchronize()
    for aggr in ("encoder", "forward"):
        stop = [False]

        def work():
            st = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(st):
                while not stop[0]:
                    am.forward_fnet(*afr) if aggr == "encoder" else am(*afr)
                    st.synchronize()

        th = threading.Thread(target=work, daemon=True)
        th.start()
        bad = 0
        try:
            st = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(st), torch.no_grad():
                for _ in range(reps):
                    out = fn()
                    st.synchronize()
                    bad += any(not torch.equal(a, b) for a, b in zip(out, base))
        finally:
            stop[0] = True
            th.join(timeout=60)
        print(f"[{name}; aggressor: fp16x2 RAFT-Stereo {aggr}] {bad} of {reps} forwards differ from the undisturbed result", flush=True)
