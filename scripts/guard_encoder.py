"""Out-of-bounds write detector for the HIP encoder: frames, fmap, cnet and the workspace are carved out of ONE arena with
sentinel-filled guard bands between them; after nnd_encoder_forward every guard must still hold the sentinel.
    python scripts/guard_encoder.py [arithmetic] [H] [W]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen, ops
from nndepth_amd.ops import lib, _p, _stream, check
from nndepth_amd.raft_stereo import BaseRAFTStereo
DEV = torch.device("cuda:0")
ar = sys.argv[1] if len(sys.argv) > 1 else "fp16x2"
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (128, 160)
m = BaseRAFTStereo(iters=2, context_dim=64, arithmetic=ar)
weightgen.fill_module_(m)
m = m.to(DEV).eval()
eng = m._encoder_engine(DEV)
N, ncnet = 2, 1
h8, w8 = H, W
for _ in range(3):
    h8, w8 = (h8 + 1) // 2, (w8 + 1) // 2
need = int(lib.nnd_encoder_workspace_floats(C.byref(eng.desc), N, H, W))
G = 1 << 20  # guard floats
sizes = [N * 3 * H * W, N * eng.desc.output_dim * h8 * w8, ncnet * eng.desc.cnet_dim * h8 * w8, need]
SENT = 12345.0
arena = torch.full((sum(sizes) + G * (len(sizes) + 1) + 1024,), SENT, dtype=torch.float32, device=DEV)
views, guards, off = [], [], 0
for s in sizes:
    guards.append(arena[off:off + G]); off += G
    off = (off + 63) // 64 * 64
    views.append(arena[off:off + s]); off += s
guards.append(arena[off:off + G])
fr = torch.cat(weightgen.synthetic_frames(20, 1, H, W), 0).to(DEV)
views[0].copy_(fr.reshape(-1))
for rep in range(3):
    check(lib.nnd_encoder_forward(C.byref(eng.desc), _p(eng.packed), _p(views[0]), _p(views[1]), _p(views[2]), ncnet, _p(views[3]), N, H, W,
                                  _stream(DEV)), "encoder_forward")
torch.cuda.synchronize()
names = ["before frames", "frames | fmap", "fmap | cnet", "cnet | workspace", "after workspace"]
for nm, g in zip(names, guards):
    badm = g != SENT
    nb = int(badm.sum())
    first = int(badm.nonzero()[0]) if nb else -1
    last = int(badm.nonzero()[-1]) if nb else -1
    print(f"[{ar} {H}x{W}] guard {nm:18s}: {nb} floats overwritten" + (f" (offsets {first}..{last} of the band)" if nb else ""))
print(f"[{ar} {H}x{W}] workspace floats {need}, frames intact: {bool(torch.equal(views[0], fr.reshape(-1)))}")
