#!/usr/bin/env python3
"""Headline benchmark: stereo pairs/s, RAFT-Stereo base, 544x960, 32 GRU iterations, batch 1 per GPU
(BASELINE.json configs[1]).  One "step" = one full forward() of one stereo pair per rank: HIP encoder
(csrc/encoder.hip) + HIP correlation pyramid + the fused HIP refinement loop producing all 32 up_disp maps;
at N > 1 every rank works on its own pair (weak scaling, no collective inside forward) and the step ends
with the RCCL all-gather of the final disparity.  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description), including
  "roofline":     dominant kernel (the implicit-GEMM conv of encoder.convc2 in the selected arithmetic) measured live
                  with hipEvents on the launch stream: `achieved` = algorithmic TFLOP/s from its duration INSIDE the
                  fused loop (nnd_profile_loop_conv), `peak` = dense 16-bit MFMA peak (2500 TFLOP/s) / MFMA products per fp32
                  product of the arithmetic (157.3 for the exact fp32 MFMA), `standalone` the same launch alone on the
                  chip (nnd_profile_conv), `event_pair_ms` what the two events alone measure at that place of the loop
                  (nnd_profile_loop_event_pair; `achieved` / `frac` stay on the raw in-loop figure); `hbm_group` = the HBM-bound kernels of the path (pyramid build, lookup,
                  upsample, IGEV / CREStereo volume kernels) against the 8 TB/s roofline, same run;
  "cpu_baseline": the oracle's PyTorch-eager CPU restatement of the reference forward timed on the
                  host cores (rank 0, N=1 only), and the GPU-vs-oracle max-abs of that same pair.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H_IMG, W_IMG, ITERS = 544, 960, 32
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
DTYPE = {"fp32": "f32", "bf16x3": "f32 carried as 3 bf16 pieces per operand (6 MFMA products, fp32 accumulate)",
         "fp16x2": "f32 carried as 2 range-scaled fp16 pieces per operand (3 MFMA products, fp32 accumulate)"}
ARITH_NOTE = {
    "fp32": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32) for every convolution",
    "bf16x3": "fp32 tensors everywhere; inside the MFMA GEMMs listed here each fp32 operand is carried as 3 bf16 pieces, "
              "x = x0+x1+x2, w = w0+w1+w2, and the 6 products x_i*w_j (i+j<=2) run on v_mfma_f32_32x32x16_bf16 with fp32 "
              "accumulation (dropped terms <= 2^-24|x||w|; per-op error vs float64 BELOW the exact fp32-MFMA kernel's, "
              "tests/test_gpu_split.py): every update-block conv except convc1 (convc2, convf2, conv, the GRU z/r/q convs and their "
              "context terms, flow_head.conv1 + mask.0, mask.2 inside the fused mask + upsample kernel) and the feature encoder's "
              "stride-1 3x3 convs + cnet_proj.  Exact fp32 (fp32 MFMA or fp32 VALU): encoder stem, stride-2 convs and 1x1 shortcuts, "
              "correlation build + lookup + convc1, convf1, flow_head.conv2, softmax + convex upsample, all epilogues.  "
              "Selectable: --arithmetic fp32 (exact path, also timed in this line as exact_fp32_path)"}
ARITH_NOTE["fp16x2"] = (
    "fp32 tensors everywhere; inside the MFMA GEMMs listed for bf16x3 each fp32 operand is carried as 2 fp16 pieces (22 significand "
    "bits), x = x0+x1, w = w0+w1, both range-scaled by exact powers of two (weights per layer at pack time, activations per layer "
    "by a scale calibrated on the first forward from the largest |activation| the layer stages: 32 x headroom above it, 13 octaves "
    "below it at full precision — csrc/calib.hip, tests/test_gpu_split.py input-scale sweep 2^-12..2^12; undone after the K loop), and the 3 products x0*w0, x0*w1, x1*w0 run on v_mfma_f32_32x32x16_f16 with fp32 accumulation: "
    "half the matrix work of bf16x3, per-op error vs float64 at the exact fp32-MFMA kernel's level (tests/test_gpu_split.py).  Same "
    "layer coverage as bf16x3; everything else exact fp32.  Selectable: --arithmetic fp32 | bf16x3")
SPLIT_PRODUCTS = {"fp32": 1, "bf16x3": 6, "fp16x2": 3}
# algorithmic work of one pair: 32 x 43.15 GFLOP loop + 0.5 GFLOP pyramid + 155.2 GFLOP encoder + cnet_proj
E2E_TFLOP = (32 * 43.15 + 0.5 + 155.2) / 1e3


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-hbm-group", action="store_true")
    ap.add_argument("--arithmetic", default="fp16x2", choices=["fp16x2", "bf16x3", "fp32"],
                    help="MFMA arithmetic of the update-block / encoder convolutions (the model classes' default is the same): fp16x2 = "
                         "fp32 operands carried as 2 range-scaled fp16 pieces, 3 products on v_mfma_f32_32x32x16_f16; bf16x3 = 3 bf16 "
                         "pieces, 6 products on v_mfma_f32_32x32x16_bf16; fp32 accumulate in both (csrc/conv_split.hip, parity-gated by "
                         "tests/test_gpu_split.py + tests/test_gpu_realdata.py); fp32 = exact fp32 MFMA (csrc/conv_mfma.hip)")
    ap.add_argument("--launch", default="direct", choices=["direct", "graph"],
                    help="direct = every kernel of a step enqueued by the C-ABI calls (default); graph = the whole forward captured once "
                         "into a HIP graph and replayed (nndepth_amd/graph.py: one host-side launch per pair, same kernels, bit-identical)")
    ap.add_argument("--config", default="raft544", choices=["raft544", "kitti64", "cre8"],
                    help="raft544 = BASELINE.json configs[1] (the headline, default); kitti64 = configs[3]: 64 KITTI-size pairs "
                         "sharded over the ranks; cre8 = configs[4]: 8 CREStereo 1080x1920 pairs, 2-stage cascade, sharded")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="control-flow check without a GPU (tests/test_dist_cpu.py): ranks meet over gloo, barrier, max over ranks, "
                         "rank 0 prints a JSON line with the world it saw; no kernel runs and no throughput is reported")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.rendezvous_only:
        return rendezvous_only(args)
    if args.config != "raft544":
        return sharded_config(args)
    import torch
    from nndepth_amd import parallel, weightgen
    from nndepth_amd.raft_stereo import BaseRAFTStereo

    rank, world, local = parallel.init_distributed("nccl")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    # spec of BaseRAFTStereo(context_dim=64) without touching oracle/: keys+shapes from the module itself
    model = BaseRAFTStereo(iters=ITERS, context_dim=64, arithmetic=args.arithmetic)
    weightgen.fill_module_(model)
    model = model.to(dev).eval()
    f1, f2 = weightgen.synthetic_frames(100 + rank, 1, H_IMG, W_IMG)  # a different pair per rank
    f1, f2 = f1.to(dev), f2.to(dev)

    fwd = model
    if args.launch == "graph":
        from nndepth_amd.graph import GraphedForward
        fwd = GraphedForward(model)

    def step():
        out = fwd(f1, f2)
        final = out[-1]["up_disp"]
        if world > 1:
            final = parallel.gather_disparity(final)
        return out, final

    if args.arithmetic == "fp16x2":  # the per-layer activation scales (csrc/calib.hip): never inside the timed region, whatever --warmup is
        model.calibrate(f1, f2)
        torch.cuda.synchronize(dev)
    log(f"rank {rank}/{world}: model + inputs resident on {dev}; warm-up x{args.warmup}")
    for _ in range(args.warmup):
        step()
        torch.cuda.synchronize(dev)
        log("warm-up step done")
    torch.cuda.synchronize(dev)
    parallel.barrier()
    torch.cuda.synchronize(dev)
    # per-step events on the launch stream (no host sync inside the timed region): median next to the mean
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        out, final = step()
        marks[i + 1].record()
    torch.cuda.synchronize(dev)
    parallel.barrier()
    torch.cuda.synchronize(dev)
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2]
    log(f"timed region: {args.steps} steps in {elapsed:.3f} s (median step {median_ms:.3f} ms)")

    result = {
        "metric": "stereo pairs/sec at 544x960, 32 iters (RAFT-Stereo)",
        "value": world * args.steps / elapsed,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "ms_per_step_median": median_ms,
        "ms_per_step_min_max": [step_ms[0], step_ms[-1]],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE[args.arithmetic],
        "data": "synthetic",
        "config": {"workload": "RAFT-Stereo base (ctx 64), 544x960, 32 iters, batch 1 per GPU, all 32 up_disp emitted",
                   "pairs_per_step": world, "parallelism": f"batch-parallel x{world}" if world > 1 else "single", "launch": args.launch,
                   "arithmetic": ARITH_NOTE[args.arithmetic]},
    }
    if args.arithmetic != "fp32" and rank == 0 and world == 1:
        # the same pair through the exact fp32-MFMA path, same run: what the split arithmetic buys and what it changes
        exact = BaseRAFTStereo(iters=ITERS, context_dim=64, arithmetic="fp32")
        exact.load_state_dict(model.state_dict())
        exact = exact.to(dev).eval()
        for _ in range(2):
            ex_out = exact(f1, f2)
        torch.cuda.synchronize(dev)
        n_ex = max(5, min(20, args.steps))
        t1 = time.perf_counter()
        for _ in range(n_ex):
            ex_out = exact(f1, f2)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t1) / n_ex
        result["exact_fp32_pairs_per_s"] = 1.0 / dt  # the same pair through the exact fp32-MFMA path, same run
        result["config"]["exact_fp32_pairs_per_s"] = 1.0 / dt  # (also inside `config`: the driver's record keeps `config` and `roofline` whole)
        result["exact_fp32_path"] = {
            "value": 1.0 / dt, "unit": "pairs/s", "ms_per_step": 1e3 * dt, "steps": n_ex, "dtype": DTYPE["fp32"],
            "max_abs_up_disp_vs_this_run": float((ex_out[-1]["up_disp"] - out[-1]["up_disp"]).abs().max())}
        del exact, ex_out

    # ------------------------------------------------------------------ roofline (dominant kernel)
    if not args.no_roofline and rank == 0:
        eng = model.update_block.sync_engine(dev)
        Hf, Wf = H_IMG // 8, W_IMG // 8
        names = eng.conv_names()
        rows, tot_ms, tot_fl = [], 0.0, 0.0
        log("roofline: per-conv hipEvent timing")
        for i, nm in enumerate(names):
            ms, fl = eng.profile_conv(i, 1, Hf, Wf, 20, dev)
            rows.append({"conv": nm, "ms": ms, "gflop": fl / 1e9, "tflops": fl / ms / 1e9})
            tot_ms += ms
            tot_fl += fl
        # dominant launch = most algorithmic FLOPs (ties -> the first, encoder.convc2: also the longest one inside the loop);
        # picking by measured time would flip between convc2 and flow_head.conv1+mask.0 (same FLOPs, 62 vs 64 us) on noise
        dom = max(rows, key=lambda r: round(r["gflop"], 3))
        # ... and the same launches where they run in production: inside the fused loop
        from nndepth_amd.cost_volume import CorrBlock1D
        fmap1, fmap2, cnet = model.forward_fnet(f1, f2)
        net, inp = torch.split(cnet, [model.hidden_dim, model.context_dim], dim=1)
        net, inp = torch.tanh(net).contiguous(), torch.relu(inp).contiguous()
        pyr = CorrBlock1D(fmap1, fmap2, 4, 4)._pyr
        loop_ms, loop_fl = 0.0, 0.0
        for i, r in enumerate(rows):
            if r["conv"] in ("encoder.convc1", "mask.2"):
                continue  # fused into lookup+convc1 / into mask+upsample: no launch of their own in the loop
            r["ms_in_loop"] = eng.profile_loop_conv(i, pyr, 4, 4, net, inp, 8, ITERS)
            r["tflops_in_loop"] = r["gflop"] / r["ms_in_loop"]
            if r["conv"] == "encoder.convf2":  # the probe brackets the whole flow branch: convf1 + convf2 (one launch with bf16x3)
                r["in_loop_covers"] = "encoder.convf1 + encoder.convf2"
            loop_ms += r["ms_in_loop"]
            loop_fl += r["gflop"]
        # what an event pair alone measures at the dominant conv's place in the loop (both events in front of it): reported beside
        # the raw figure, which stays the one `achieved` / `frac` are computed from
        ev_pair_ms = eng.profile_loop_conv(rows.index(dom), pyr, 4, 4, net, inp, 8, ITERS, event_pair_only=True)
        nprod = SPLIT_PRODUCTS[args.arithmetic]  # MFMA FLOPs executed per algorithmic FLOP
        # what the matrix pipes SUSTAIN on this box under the power limit with nothing else in flight (diagnostic, ~60 ms): context for
        # `executed_mfma_tflops`, not the roofline's `peak` (that stays the guide's nominal figure)
        sustained = None
        if nprod > 1:
            import ctypes as _C
            from nndepth_amd._lib import lib as _nlib, check as _ncheck
            scr = torch.zeros(16, device=dev)
            clk = torch.zeros(512, dtype=torch.int64, device=dev)
            tf, ghz = _C.c_float(), _C.c_float()
            with torch.cuda.device(dev):
                _ncheck(_nlib.nnd_profile_mfma16_peak(1 if nprod == 6 else 0, 3, 20.0, _C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                                                      _C.c_void_p(scr.data_ptr()), _C.c_void_p(clk.data_ptr()), _C.byref(tf), _C.byref(ghz)),
                        "profile_mfma16_peak")
            sustained = {"tflops": tf.value, "shader_clock_ghz": ghz.value,
                         "measured": "4 independent v_mfma_f32_32x32x16 chains per wave, 3 waves per SIMD on every CU, pseudo-random operands "
                                     "in registers, 20 ms (nnd_profile_mfma16_peak); nominal 2500 TFLOP/s at 2.4 GHz"}
        # `achieved` / `peak` are ALGORITHMIC TFLOP/s: peak = what the matrix pipe could deliver of this arithmetic's fp32-equivalent
        # products = dense 16-bit MFMA peak / products per fp32 product (fp16x2: 2500 / 3 = 833; bf16x3: 2500 / 6 = 417; fp32: 157.3).
        # frac is therefore also the executed-MFMA utilisation (executed = nprod x algorithmic, against 2500).
        peak = PEAK_FP32_MFMA_TFLOPS if nprod == 1 else PEAK_BF16_MFMA_TFLOPS / nprod
        kern = {1: "conv_mfma_kernel (fp32 v_mfma_f32_32x32x2_f32)",
                6: "conv_split_kernel (v_mfma_f32_32x32x16_bf16, 6 products per fp32 product)",
                3: "conv_split_kernel (v_mfma_f32_32x32x16_f16, 3 products per fp32 product)"}[nprod]
        result["roofline"] = {
            "bound": "mfma", "kernel": kern + " — " + dom["conv"],
            "achieved": dom["tflops_in_loop"], "peak": peak, "unit": "TFLOP/s",
            "frac": dom["tflops_in_loop"] / peak, "traffic": _traffic(args.arithmetic),
            "flops_counted": "algorithmic (2*B*H*W*Cout*Cin*KH*KW); peak = dense 16-bit MFMA peak 2500 TFLOP/s / %d MFMA products per "
                             "fp32 product" % nprod if nprod > 1 else "algorithmic (2*B*H*W*Cout*Cin*KH*KW) against the fp32 MFMA peak",
            "executed_mfma_tflops": nprod * dom["tflops_in_loop"], "executed_mfma_peak": PEAK_BF16_MFMA_TFLOPS if nprod > 1 else PEAK_FP32_MFMA_TFLOPS,
            "sustained_mfma_on_this_box": sustained,
            "executed_frac_of_sustained_mfma": (nprod * dom["tflops_in_loop"] / sustained["tflops"]) if sustained else None,
            "algorithmic_frac_of_fp32_mfma_peak": dom["tflops_in_loop"] / PEAK_FP32_MFMA_TFLOPS,
            "measured": "inside the fused 32-iteration loop (hipEvents around the launch on its stream, average of iterations 2..32)",
            "launch_ms": dom["ms_in_loop"], "launch_gflop": dom["gflop"],
            "event_pair_ms": ev_pair_ms,
            "launch_ms_minus_event_pair": dom["ms_in_loop"] - ev_pair_ms,
            "frac_minus_event_pair": dom["gflop"] / max(dom["ms_in_loop"] - ev_pair_ms, 1e-6) / peak,
            "standalone": {"launch_ms": dom["ms"], "achieved": dom["tflops"], "frac": dom["tflops"] / peak},
            # aggregates in ALGORITHMIC TFLOP/s against the same `peak` (the arithmetic's ceiling); *_vs_fp32_mfma_peak: against the
            # 157.3 TFLOP/s of the exact fp32 MFMA the path sat under until round 2
            "all_convs": {"ms_per_iter": tot_ms, "gflop_per_iter": tot_fl / 1e9,
                          "tflops": tot_fl / tot_ms / 1e9, "frac": tot_fl / tot_ms / 1e9 / peak,
                          "frac_vs_fp32_mfma_peak": tot_fl / tot_ms / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                          "measured": "stand-alone launches; algorithmic FLOPs"},
            "loop_convs": {"ms_per_iter": loop_ms, "gflop_per_iter": loop_fl,
                           "tflops": loop_fl / loop_ms, "frac": loop_fl / loop_ms / peak,
                           "frac_vs_fp32_mfma_peak": loop_fl / loop_ms / PEAK_FP32_MFMA_TFLOPS,
                           "measured": "the 8 stand-alone conv launches of an iteration, timed in the loop; algorithmic FLOPs"},
            "end_to_end": {"tflop_per_pair": E2E_TFLOP, "tflops": E2E_TFLOP / (elapsed / args.steps),
                           "frac": E2E_TFLOP / (elapsed / args.steps) / peak,
                           "frac_vs_fp32_mfma_peak": E2E_TFLOP / (elapsed / args.steps) / PEAK_FP32_MFMA_TFLOPS,
                           "measured": "algorithmic FLOPs of the whole pair (encoder, pyramid and every VALU kernel included) / step time"},
            "per_conv": rows,
        }
        if not args.no_hbm_group:
            from nndepth_amd import profiling
            log("roofline: HBM-bound kernels vs 8 TB/s")
            torch.manual_seed(0)
            result["roofline"]["hbm_group"] = {
                "peak_gb_per_s": profiling.HBM_PEAK_GBS,
                "raft_544x960": profiling.raft_rows(dev),
                "igev_544x960_per_sample": profiling.igev_rows(dev),
                "cre_1080x1920": profiling.cre_rows(dev),
                "timed": "kernel durations: 50 (12-20 for the >100 MB kernels) launches captured into a HIP graph, graph replayed between "
                         "two events (no Python / ctypes launch overhead in the figure)",
                "not_captured": profiling.NOT_CAPTURED,
            }
            torch.cuda.empty_cache()

    # ------------------------------------------------------------------ CPU baseline (oracle port)
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from oracle import torch_ref as R  # checker / baseline only — never on the product path
        ncpu = usable_cores()
        torch.set_num_threads(ncpu)
        log(f"cpu baseline on {ncpu} cores (oracle port, PyTorch CPU eager)")
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        c1, c2 = f1.cpu(), f2.cpu()
        with torch.no_grad():
            ref = R.raft_stereo_forward(sd, c1, c2, ITERS)  # warm-up + parity reference
            n, t_cpu = 0, 0.0
            while n < 12 and t_cpu < 12.0:  # a bounded sample: about 12 s of CPU work
                t1 = time.perf_counter()
                R.raft_stereo_forward(sd, c1, c2, ITERS)
                t_cpu += time.perf_counter() - t1
                n += 1
                log(f"cpu forward {n}: {t_cpu:.1f} s total")
        result["cpu_baseline"] = {
            "value": n / t_cpu, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} full forwards of the same 544x960 / 32-iter pair (oracle/torch_ref.py, PyTorch CPU eager fp32)",
        }
        result["parity_max_abs_vs_oracle"] = float((out[-1]["up_disp"].cpu() - ref[-1]).abs().max())
        result["config"]["parity_max_abs_vs_oracle"] = result["parity_max_abs_vs_oracle"]

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        parallel.barrier()  # rank 0's roofline leg runs after the timed region: leave together
        dist.destroy_process_group()


def self_launch(n, argv, backend_env=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD `python -m torch.distributed.run`
    (one process per GPU, rendezvous on 127.0.0.1) and relay its output — rank 0's JSON line is the last stdout line.
    Called before anything has touched the GPU or imported torch: this process never initialises HIP and is never
    replaced (no exec), it only waits for the child and returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:  # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required by RCCL on this pool
    env.update(backend_env or {})
    log(f"--gpus {n} without WORLD_SIZE: launching {' '.join(cmd)}")
    return subprocess.run(cmd, env=env).returncode


def rendezvous_only(args):
    from nndepth_amd import parallel
    rank, world, local = parallel.init_distributed("gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    parallel.barrier()
    slowest = parallel.max_over_ranks(1.0 + rank, "cpu")
    parallel.barrier()
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "config": args.config, "max_over_ranks": slowest}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def sharded_config(args):
    """BASELINE.json configs[3] / configs[4]: a fixed batch of pairs sharded over the ranks (strong scaling; one step = one
    pass over the whole batch; the only collective is the all-gather of the final disparities, nndepth_amd/parallel.py).

      kitti64: 64 pairs of 375x1242 -> Padder(divis_by=32) -> RAFT-Stereo base, 32 iterations, micro-batches of 8 pairs ->
               unpad -> all-gather of (64,1,375,1242)                          (8 pairs per GPU on a full node)
      cre8:    8 pairs of 1080x1920 -> CREStereo, 20 iterations, 2-stage cascade (half resolution, then full resolution
               seeded with it) -> all-gather of (8,2,1080,1920)                (1 pair per GPU on a full node)
    """
    import torch
    from nndepth_amd import parallel, weightgen

    rank, world, local = parallel.init_distributed("nccl")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if args.config == "kitti64":
        from nndepth_amd.prepost import Padder
        from nndepth_amd.raft_stereo import BaseRAFTStereo
        n_pairs, micro, hw = 64, 8, (375, 1242)
        model = BaseRAFTStereo(iters=ITERS, context_dim=64, arithmetic=args.arithmetic)
        weightgen.fill_module_(model)
        model = model.to(dev).eval()
        padder = Padder(hw, divis_by=32)

        def forward(f1, f2):
            p1, p2 = padder.pad(f1, f2)
            return padder.unpad(model(p1, p2)[-1]["up_disp"])
        name = "RAFT-Stereo base (ctx 64), 64 x 375x1242 (Padder 32 -> 384x1248), 32 iters, micro-batches of 8"
        metric = "stereo pairs/sec at KITTI 1242x375, 32 iters (RAFT-Stereo), batch 64 sharded"
    else:
        from nndepth_amd.cre_stereo import CREStereoBase, two_stage_forward
        n_pairs, micro, hw = 8, 1, (1080, 1920)
        model = CREStereoBase(iters=20, arithmetic=args.arithmetic)
        weightgen.fill_module_(model)
        model = model.to(dev).eval()

        def forward(f1, f2):
            return two_stage_forward(model, f1, f2)[-1]["up_disp"]
        name = "CREStereo, 8 x 1080x1920, 20 iters, 2-stage cascade (half resolution, then full resolution seeded with it)"
        metric = "stereo pairs/sec at 1080x1920, 20 iters (CREStereo 2-stage), batch 8 sharded"
    assert world <= n_pairs
    # the rank's pairs resident in HBM before the timed region (synthetic, a different pair per id)
    mine = list(parallel.shard_range(n_pairs, rank, world))
    cache = {i: tuple(t.to(dev) for t in weightgen.synthetic_frames(200 + i, 1, *hw)) for i in mine}

    def load(ids):
        return torch.cat([cache[i][0] for i in ids]), torch.cat([cache[i][1] for i in ids])

    def step():
        return parallel.sharded_inference(n_pairs, load, forward, micro, rank, world)

    step()  # untimed: packs the weights and calibrates the fp16x2 activation scales (csrc/calib.hip), whatever --warmup is
    torch.cuda.synchronize(dev)
    log(f"rank {rank}/{world}: {len(mine)} of {n_pairs} pairs resident on {dev}; warm-up x{args.warmup}")
    for _ in range(args.warmup):
        step()
        torch.cuda.synchronize(dev)
    parallel.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    parallel.barrier()
    torch.cuda.synchronize(dev)
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    assert out.shape[0] == n_pairs and tuple(out.shape[2:]) == hw
    if rank == 0:
        print(json.dumps({
            "metric": metric, "value": n_pairs * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": DTYPE[args.arithmetic], "data": "synthetic",
            "config": {"workload": name, "arithmetic": args.arithmetic, "pairs_per_step": n_pairs, "pairs_per_rank": len(mine),
                       "parallelism": f"batch shards x{world} + RCCL all-gather of the disparities"}}))
    if world > 1:
        import torch.distributed as dist
        parallel.barrier()
        dist.destroy_process_group()


def _traffic(arithmetic="fp32"):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (profiles/), else None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            d = json.load(open(p))
            return d.get("dominant_kernel_hbm_bytes_per_launch_" + arithmetic, d.get("dominant_kernel_hbm_bytes_per_launch") if arithmetic == "fp32" else None)
        except Exception:
            return None
    return None


if __name__ == "__main__":
    main()
