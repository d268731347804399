"""CREStereo forward (victim) beside another stream's fp16x2 RAFT-Stereo encoder: mismatching forwards per diagnostic switch /
arithmetic (localisation).   python scripts/race_cre.py reps [arithmetic]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nndepth_amd import weightgen
from nndepth_amd.raft_stereo import BaseRAFTStereo
from nndepth_amd.cre_stereo import CREStereoBase
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ar = sys.argv[2] if len(sys.argv) > 2 else "fp16x2"
am = BaseRAFTStereo(iters=4, context_dim=64, arithmetic="fp16x2")
weightgen.fill_module_(am)
am = am.to(DEV).eval()
afr = tuple(f.to(DEV) for f in weightgen.synthetic_frames(21, 1, 128, 160))
am(*afr)
c = CREStereoBase(iters=2, arithmetic=ar)
weightgen.fill_module_(c)
c = c.to(DEV).eval()
if os.environ.get("VICTIM_ENC"):
    c.arithmetic_encoder = os.environ["VICTIM_ENC"]
g = tuple(x.to(DEV) for x in weightgen.synthetic_frames(3, 1, 256, 320))
with torch.no_grad():
    base = [o["up_disp"].clone() for o in c(*g)]
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
bad, first = 0, {}
try:
    st = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(st), torch.no_grad():
        for _ in range(reps):
            out = [o["up_disp"] for o in c(*g)]
            st.synchronize()
            for k, (a, b) in enumerate(zip(out, base)):
                if not torch.equal(a, b):
                    bad += 1
                    first[k] = first.get(k, 0) + 1
                    break
finally:
    stop[0] = True
    th.join(timeout=60)
print(f"[CREStereo {ar} {' '.join(k + '=' + os.environ[k] for k in os.environ if k.startswith('NND_') or k.startswith('VICTIM_'))}] {bad} of {reps} forwards differ; first differing output index {first}", flush=True)
