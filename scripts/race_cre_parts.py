"""Which part of the CREStereo forward is not reproducible beside another stream's fp16x2 encoder: each part runs as the victim
with fixed inputs, compared bit for bit with its undisturbed result.   python scripts/race_cre_parts.py reps [arithmetic]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nndepth_amd import weightgen, ops
from nndepth_amd.raft_stereo import BaseRAFTStereo
from nndepth_amd.cre_stereo import CREStereoBase
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ar = sys.argv[2] if len(sys.argv) > 2 else "fp32"
am = BaseRAFTStereo(iters=4, context_dim=64, arithmetic="fp16x2")
weightgen.fill_module_(am)
am = am.to(DEV).eval()
afr = tuple(f.to(DEV) for f in weightgen.synthetic_frames(21, 1, 128, 160))
am(*afr)
c = CREStereoBase(iters=2, arithmetic=ar)
weightgen.fill_module_(c)
c = c.to(DEV).eval()
g = tuple(x.to(DEV) for x in weightgen.synthetic_frames(3, 1, 256, 320))
with torch.no_grad():
    fmap1, fmap2 = (t.float() for t in c.forward_fnet(*g))
    net, inp = ops.split_tanh_relu(fmap1, c.hidden_dim)
    f1_8, f1_16 = ops.avg_pool_2x_4x(fmap1)
    f2_8, f2_16 = ops.avg_pool_2x_4x(fmap2)
    net8, net16 = ops.avg_pool_2x_4x(net)
    inp8, inp16 = ops.avg_pool_2x_4x(inp)
    conv16, conv8 = c._offset_convs(DEV)
    off16 = ops.conv2d_offset(conv16, f1_16, c.range_16)
    off8 = ops.conv2d_offset(conv8, f1_8, c.range_8)
    a1, a2 = c.self_att_fn.forward_maps(f1_16, f2_16)
    flow16 = torch.zeros(1, 2, *f1_16.shape[2:], device=DEV)
    flow8 = torch.randn(1, 2, *f1_8.shape[2:], device=DEV)
    flow4 = torch.randn(1, 2, *fmap1.shape[2:], device=DEV)


def stage(f1, f2, att, n_, i_, fl, off, iters, iter_mode):
    outs = []
    c._stage(c.corr_cls(f1, f2, att=att) if att is not None else c.corr_cls(f1, f2), n_, i_, fl, off, iters, iter_mode, outs)
    return [o["up_disp"] for o in outs]


parts = {
    "encoder (instance norm)": lambda: list(c.forward_fnet(*g)),
    "split_tanh_relu + avg pools + offset convs": lambda: list(ops.split_tanh_relu(fmap1, c.hidden_dim)) + list(ops.avg_pool_2x_4x(fmap1)) + [ops.conv2d_offset(conv16, f1_16, c.range_16), ops.conv2d_offset(conv8, f1_8, c.range_8)],
    "LoFTR self-attention (forward_maps)": lambda: list(c.self_att_fn.forward_maps(f1_16, f2_16)),
    "stage 1/16 (cross attention + offset AGCL + update block)": lambda: stage(a1, a2, c.cross_att_fn, net16, inp16, flow16, off16, 1, False),
    "stage 1/8 (offset AGCL + update block)": lambda: stage(f1_8, f2_8, None, net8, inp8, flow8, off8, 1, False),
    "stage 1/4 (window AGCL + update block), 2 iterations": lambda: stage(fmap1, fmap2, None, net, inp, flow4, None, 2, True),
    "resize_bilinear_ac": lambda: [ops.resize_bilinear_ac(flow8, fmap1.shape[2:], 2.0)],
}
for name, fn in parts.items():
    with torch.no_grad():
        base = [o.clone() for o in fn()]
    torch.cuda.synchronize()
    stop = [False]

    def work():
        st = torch.cuda.Stream(device=DEV)
        with torch.cuda.stream(st):
            while not stop[0]:
                am.forward_fnet(*afr)
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
    print(f"[CREStereo {ar}: {name}] {bad} of {reps} differ", flush=True)
