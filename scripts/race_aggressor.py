"""Victim / aggressor localisation of the two-stream mismatch: thread A runs the RAFT-Stereo forward (victim) and is compared
with its serial result; thread B runs ONE kind of work in a loop on its own stream (aggressor): the HIP encoder in an arithmetic,
the refinement loop only, a torch matmul, or nothing.   python scripts/race_aggressor.py victim_arith aggressor reps"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen
from nndepth_amd.raft_stereo import BaseRAFTStereo
DEV = "cuda:0"
var, aggr, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])


def model(ar):
    m = BaseRAFTStereo(iters=6, context_dim=64, arithmetic=ar)
    weightgen.fill_module_(m)
    return m.to(DEV).eval()


victim = model(var)
if os.environ.get("VICTIM_ENC"):  # the victim's encoder in another arithmetic than its loop (localisation)
    victim.arithmetic = os.environ["VICTIM_ENC"]
fr = tuple(f.to(DEV) for f in weightgen.synthetic_frames(20, 1, 96, 160))
serial = [o["up_disp"].clone() for o in victim(*fr)]
am = model(aggr.split(":")[1]) if ":" in aggr else None
afr = tuple(f.to(DEV) for f in weightgen.synthetic_frames(21, 1, 128, 160))
if am is not None:
    am(*afr)
mmx = torch.randn(2048, 2048, device=DEV)
torch.cuda.synchronize()
stop = [False]
bad = 0
first = {}


def aggressor():
    st = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(st):
        while not stop[0]:
            if aggr.startswith("enc"):
                am.forward_fnet(*afr)
            elif aggr.startswith("full"):
                am(*afr)
            elif aggr == "mm":
                (mmx @ mmx)
            st.synchronize()


th = threading.Thread(target=aggressor)
if aggr != "none":
    th.start()
st = torch.cuda.Stream(device=DEV)
with torch.cuda.stream(st):
    for rep in range(reps):
        out = victim(*fr)
        st.synchronize()
        for k in range(6):
            if not torch.equal(out[k]["up_disp"], serial[k]):
                bad += 1
                first[k] = first.get(k, 0) + 1
                break
stop[0] = True
if aggr != "none":
    th.join()
print(f"[victim {var} {' '.join(k + '=' + os.environ[k] for k in os.environ if k.startswith('NND_') or k.startswith('VICTIM_'))}, aggressor {aggr}] {bad} mismatching forwards of {reps}; first differing iteration histogram {first}")
