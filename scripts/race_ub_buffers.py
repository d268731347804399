"""Which workspace tensor of one fp32 update-block step (CREStereo shape: hidden 128, context 128, 2-channel flow) first differs
from the undisturbed step beside another stream's fp16x2 encoder.   python scripts/race_ub_buffers.py reps [fc] [ctx]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen
from nndepth_amd.raft_stereo import BaseRAFTStereo
from nndepth_amd.blocks import BasicUpdateBlock
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fc = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 128
ar = sys.argv[4] if len(sys.argv) > 4 else "fp32"
H, W = 64, 80
am = BaseRAFTStereo(iters=4, context_dim=64, arithmetic="fp16x2"); weightgen.fill_module_(am); am = am.to(DEV).eval()
afr = tuple(f.to(DEV) for f in weightgen.synthetic_frames(21, 1, 128, 160)); am(*afr)
torch.manual_seed(0)
ub = BasicUpdateBlock(hidden_dim=128, cor_planes=36, context_dim=ctx, flow_channel=fc, spatial_scale=4, arithmetic=ar)
weightgen.fill_module_(ub, "update_block."); ub = ub.to(DEV).eval()
net, inp = torch.tanh(torch.randn(1, 128, H, W, device=DEV)), torch.relu(torch.randn(1, ctx, H, W, device=DEV))
corr, flow = torch.randn(1, 36, H, W, device=DEV), torch.randn(1, fc, H, W, device=DEV) * 3
eng = ub.sync_engine(DEV)
with torch.no_grad():
    base = [o.clone() for o in ub(net, inp, corr, flow)]
ws = eng._ws
n = ((H + 3) // 4) * ((W + 7) // 8) * 32
names, sizes = ["c1", "cf", "f1", "hx", "z", "rh", "fm", "corr", "mask", "delta", "coords", "flow"], [256, 256, 128, 256 + ctx, 128, 128, 384, 36, 16 * 9, fc, 1, fc]
offs, o = [], 0
for s in sizes:
    offs.append(o)
    o += (s * n + 63) // 64 * 64
base_ws = ws.clone()
torch.cuda.synchronize()
stop = [False]


def work():
    st = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(st):
        while not stop[0]:
            am.forward_fnet(*afr); st.synchronize()


th = threading.Thread(target=work, daemon=True); th.start()
bad, hist = 0, {}
try:
    st = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(st), torch.no_grad():
        for rep in range(reps):
            out = ub(net, inp, corr, flow); st.synchronize()
            if any(not torch.equal(a, b) for a, b in zip(out, base)):
                bad += 1
                d = ws != base_ws
                which = tuple(nm for nm, of, sz in zip(names, offs, sizes) if bool(d[of:of + sz * n].any()))
                hist[which] = hist.get(which, 0) + 1
                if bad <= 6 and "delta" in which:
                    of = offs[names.index("delta")]
                    dd = (ws[of:of + fc * n] - base_ws[of:of + fc * n]).view(fc, n // 32, 32)
                    nzt = sorted(set(dd.abs().sum(2).nonzero()[:, 1].tolist()))
                    t0 = nzt[0]
                    print(f"   rep {rep}: delta differs in sub-tiles {nzt[:12]} ({len(nzt)}); sub-tile {t0}: channel 0 diffs {[round(float(v), 5) for v in dd[0, t0]]}")
                    print(f"            channel 1 diffs {[round(float(v), 5) for v in dd[1, t0]] if fc > 1 else None}; values ch0 {[round(float(v), 3) for v in base_ws[of:of + n].view(-1, 32)[t0][:8]]}")
finally:
    stop[0] = True; th.join(timeout=60)
print(f"[update block step {ar} fc={fc} ctx={ctx} {H}x{W}] {bad} of {reps} differ; workspace tensors differing: {hist}")
