"""Repeats the two-host-threads / two-streams test of tests/test_gpu_parity.py and counts mismatches against the serial results
per (thread, iteration), for the default build and with diagnostic switches (to localise a kernel whose result depends on how its
waves are scheduled):   python scripts/race_threads.py [reps] [arithmetic]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen
from nndepth_amd.raft_stereo import BaseRAFTStereo
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ar = sys.argv[2] if len(sys.argv) > 2 else "fp16x2"


def model():
    m = BaseRAFTStereo(iters=6, context_dim=64, arithmetic=ar)
    weightgen.fill_module_(m)
    if os.environ.get("RACE_ENC"):  # the encoder in another arithmetic than the loop (localisation)
        m.arithmetic = os.environ["RACE_ENC"]
    return m.to(DEV).eval()


models = [model(), model()]
frames = [tuple(f.to(DEV) for f in weightgen.synthetic_frames(20 + i, 1, 96 + 32 * i, 160)) for i in range(2)]
serial = [[o["up_disp"].clone() for o in models[i](*frames[i])] for i in range(2)]
torch.cuda.synchronize()
bad = {}
for rep in range(reps):
    results = [None, None]
    gate = threading.Barrier(2)

    def work(i):
        st = torch.cuda.Stream(device=DEV)
        with torch.cuda.stream(st):
            gate.wait()
            for _ in range(6):
                out = models[i](*frames[i])
            st.synchronize()
        results[i] = out

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(2):
        for k in range(6):
            if not torch.equal(results[i][k]["up_disp"], serial[i][k]):
                d = (results[i][k]["up_disp"] - serial[i][k]).abs()
                bad.setdefault((i, k), []).append((rep, d.max().item(), int((d > 0).sum())))
                break
print(f"[{ar}] switches: {' '.join(k + '=' + os.environ[k] for k in os.environ if k.startswith('NND_') or k.startswith('RACE_')) or '-'}: {sum(len(v) for v in bad.values())} mismatching runs of {2 * reps}")
for k, v in sorted(bad.items()):
    print("   thread, first differing iteration", k, "(rep, max-abs, elements):", v[:4])
