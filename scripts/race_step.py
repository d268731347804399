"""One update-block step with fixed inputs (victim) under an aggressor thread running the fp16x2 HIP encoder: which outputs differ
from the undisturbed step, where (pixels / channels), and by how much.   python scripts/race_step.py reps"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndepth_amd import weightgen
from nndepth_amd.blocks import BasicUpdateBlock
from nndepth_amd.raft_stereo import BaseRAFTStereo
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
H, W = int(os.environ.get("RS_H", 12)), int(os.environ.get("RS_W", 20))
ub = BasicUpdateBlock(hidden_dim=128, cor_planes=36, context_dim=64, flow_channel=1, spatial_scale=8, arithmetic="fp16x2")
weightgen.fill_module_(ub, "update_block.")
ub = ub.to(DEV).eval()
torch.manual_seed(21)
net, inp = torch.tanh(torch.randn(1, 128, H, W)).to(DEV), torch.relu(torch.randn(1, 64, H, W)).to(DEV)
corr, flow = torch.randn(1, 36, H, W).to(DEV), (torch.randn(1, 1, H, W) * 3).to(DEV)
eng = ub.sync_engine(DEV)
with torch.no_grad():
    base = [o.clone() for o in ub(net, inp, corr, flow)]
PL = ((H + 3) // 4) * ((W + 7) // 8) * 32  # tile-major plane of one channel
ws = eng._ws


def cf_snapshot():  # workspace tensor cf (256 channels: 0..191 convc2, 192..255 the flow branch), 4 channels interleaved
    return ws[256 * PL:2 * 256 * PL].clone().view(64, PL, 4).permute(0, 2, 1).reshape(256, PL)


base_cf = cf_snapshot()
import ctypes as C, numpy as np
from nndepth_amd._lib import LIB_PATH
raw = C.CDLL(LIB_PATH)
DUMP = hasattr(raw, "nnd_debug_read_fb_dump")  # only in an experimental build that dumped the kernel's LDS (not kept)


def fb_dump():
    w, p = np.zeros((64, 256), np.float32), np.zeros((64, 2048), np.uint32)
    assert raw.nnd_debug_read_fb_dump(w.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p)) == 0
    return w, p


if DUMP:
    torch.cuda.synchronize()
    base_win, base_patch = fb_dump()
am = BaseRAFTStereo(iters=2, context_dim=64, arithmetic="fp16x2")
weightgen.fill_module_(am)
am = am.to(DEV).eval()
afr = tuple(f.to(DEV) for f in weightgen.synthetic_frames(21, 1, 128, 160))
am(*afr)
torch.cuda.synchronize()
stop = [False]


def aggressor():
    st = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(st):
        while not stop[0]:
            am.forward_fnet(*afr)
            st.synchronize()


th = threading.Thread(target=aggressor, daemon=True)
th.start()
import atexit
atexit.register(lambda: stop.__setitem__(0, True))
bad, shown = 0, 0
st = torch.cuda.Stream(device=DEV)
try:
  with torch.cuda.stream(st), torch.no_grad():
    for rep in range(reps):
        out = ub(net, inp, corr, flow)
        st.synchronize()
        d = [(o - b).abs() for o, b in zip(out, base)]
        if any(float(x.max()) > 0 for x in d):
            bad += 1
            if shown < 6 and DUMP:
                w, p = fb_dump()
                for t in range(9):
                    dw = np.nonzero(w[t, :192] != base_win[t, :192])[0]
                    dp = np.nonzero(p[t] != base_patch[t])[0]
                    # data words only: row stride 896 B (10 positions x 80 B, padded), position = 64 B of pieces + 16 B pad
                    dp = np.array([x for x in dp.tolist() if (x * 4) % 896 < 800 and ((x * 4) % 896) % 80 < 64], dtype=np.int64)
                    if len(dw) or len(dp):
                        print(f"      dump sub-tile {t}: window elements differing {dw.tolist()[:40]} (row = e // 16) values {w[t, dw[:4]].tolist()} vs {base_win[t, dw[:4]].tolist()}; patch DATA words differing {len(dp)}: (row, position, word) {[((x * 4) // 896, ((x * 4) % 896) // 80, (((x * 4) % 896) % 80) // 4) for x in dp.tolist()[:12]]} got {[hex(int(v)) for v in p[t, dp[:4]]]} base {[hex(int(v)) for v in base_patch[t, dp[:4]]]}")
            if shown < 6:
                dc = (cf_snapshot() - base_cf).abs()
                nz = (dc > 0).nonzero()
                chs = sorted(set(nz[:, 0].tolist()))
                tiles = sorted(set((nz[:, 1] // 32).tolist()))
                print(f"   rep {rep} cf buffer: {len(nz)} elements differ, max {float(dc.max()):.3e}; channels {chs[:8]}..{chs[-4:] if chs else []} ({len(chs)}), sub-tiles {tiles}")
                for t in tiles[:2]:
                    blk = dc[192:256, t * 32:(t + 1) * 32]
                    cnt = (blk > 0).sum(0).view(4, 8).tolist()
                    mx = blk.max(0).values.view(4, 8)
                    print(f"      sub-tile {t}: channels differing per pixel (4 rows x 8 cols): {cnt}")
                    print(f"      sub-tile {t}: max |diff| per pixel: {[[round(float(v), 4) for v in r] for r in mx]}")
                    cc = (blk > 0).sum(1).tolist()
                    print(f"      sub-tile {t}: pixels differing per channel: {cc}")
                    print(f"      sub-tile {t}: differing (channel-192, pixel) count {int((blk > 0).sum())} of 2048; rel. size vs value {float(blk.max()):.2e} / {float(base_cf[192:256, t*32:(t+1)*32].abs().max()):.2e}; channels {sorted(set((blk > 0).nonzero()[:, 0].tolist()))[:12]} pixels {sorted(set((blk > 0).nonzero()[:, 1].tolist()))[:12]}")
            if shown < 6:
                shown += 1
                for name, x in zip(("net", "mask", "delta"), d):
                    nz = (x > 0).nonzero()
                    if len(nz):
                        ch = sorted(set(nz[:, 1].tolist()))
                        ys = sorted(set(nz[:, 2].tolist()))
                        xs = sorted(set(nz[:, 3].tolist()))
                        print(f"   rep {rep} {name}: {len(nz)} elements differ, max {float(x.max()):.2e}; channels {len(ch)} ({ch[:6]}...), rows {ys}, cols {xs}")
finally:
    stop[0] = True
th.join(timeout=30)
print(f"[step {H}x{W} {' '.join(k + '=' + os.environ[k] for k in os.environ if k.startswith('NND_'))}] {bad} mismatching steps of {reps}")
