#!/usr/bin/env python3
"""Benchmark of the QFA hot path on MI355X: spectra/s per training ("EM") step.

A step = QFA.forward (NLL + six gradient sums + counts over every resident spectrum) + the
data-parallel all-reduce of the packed sum/count buffer (N > 1) + Adam.update + clip -- the
reference's model.py:212-214 / :316 for one batch -- with the spectra already resident in HBM.

Workload (BASELINE.json metric "spectra/sec per EM step (N_pix=4000, N_h=16)", configs[2]):
100 000 synthetic spectra PER GPU (weak scaling), N_pix=4000 (N_b=1506), N_h=16, random pixel
masks, per-z absorption noise; synthetic data drawn from the model itself (qfa_amd/synthetic.py,
SURVEY.md 8(d)), seeds 20220702 + rank.  `--config c2` selects configs[1] (10k x 2000, N_h=8,
no masks).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (pass 2: k_grads_x at N_h = 9..16, k_grads at N_h <= 8,
the three launches k_s12_x + 2 k_grads_s3 at N_h = 17..32): its
algorithmic flops over its mean duration measured with HIP events recorded by
the library on the launch stream inside the timed region.  Two roofs are reported for it: `frac` against the float32
MFMA / VALU peak (157.3 TFLOP/s, the roof SURVEY.md 8(d) names: the arithmetic is float32), and `frac_xdl` against
the bf16 XDL pipe the contractions are actually issued on -- every float32 product is six bf16 MFMAs over operands
split into three bf16 pieces (three over the two leading pieces in stage 3 of pass 2), so the kernel's
contraction flops x 6 (x 3) are priced against the dense bf16 peak (2.5 PFLOP/s).  The kernels are BUILT against the XDL roof; `frac` can therefore exceed what the
f32 pipe could give.  After the timed region the same step runs for >= 3 s more (`sustained_*`: the clock the chip
holds under seconds of this load, not a burst), then `QFA.predict` is timed (`predict`: spectra/s and the HBM roofline
of its output writer), then `cpu_baseline` times the dense O(N_pix^3) CPU port of the reference's per-spectrum step
(oracle/dense_port.py) on a bounded sample of the same batch (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

CONFIGS = {
    #        B/GPU   Npix  Nh  masks cpu_sample
    "c2": (10000, 2000, 8, False, 64),
    "c3": (100000, 4000, 16, True, 32),
    "c1": (128, 1913, 8, True, 64),
    "c5": (20000, 8000, 32, True, 4),
}
PEAK_FP32_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
PEAK_BF16_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (XDL) peak
PEAK_HBM_GBS = 8000.0
# the true reference (imported in the survey container, SURVEY.md section 6 / BASELINE.md section 2): forward, 8 cores
REFERENCE_8CORE = {"c1": 13.4, "c2": 12.2, "c3": 1.54, "c5": 0.21}


def alg_flops(npix, k):
    """7 n k^2 + 14 n k + 65 n + 3 k^3 per spectrum, split per kernel:
    pass 1 (C, T, b, b2): 4nk^2 + 7nk + 30n + 3k^3; pass 2 (diag Sigma^-1, M Z, u): 3nk^2 + 7nk + 35n."""
    p1 = 4 * npix * k * k + 7 * npix * k + 30 * npix + 3 * k ** 3
    p2 = 3 * npix * k * k + 7 * npix * k + 35 * npix
    return p1, p2


def alg_bytes(npix, nb):
    """delta, sigma (4 B), mask (1 B) per pixel, zabs (4 B) per blue pixel, NLL out."""
    return npix * 9 + nb * 4 + 4


def usable_cores():
    """CPUs this process may actually run on: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a
    one-GPU job a 16-CPU share of a 256-CPU host; 128 torch threads on 16 CPUs ran the baseline 2x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                   # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    env = os.environ.get("QFA_CPU_THREADS")
    return int(env) if env else n


def cpu_baseline(params, batch, n_sample, npix, config):
    """Dense CPU port of the reference step on the first spectra of the batch: torch threads pinned to the usable
    cores, sample grown until ~10 s of CPU work (at most n_sample spectra, bounded at ~30 s)."""
    import numpy as np
    import torch
    from oracle import dense_port as DP
    from oracle import qfa_oracle as O
    cores = max(1, min(usable_cores(), 64))
    torch.set_num_threads(cores)
    P = DP.to_torch_params(params)
    host = [x[:n_sample].cpu() for x in batch]
    d, e, z, m = host
    DP.dense_forward(P, d[:1], e[:1], z[:1], m[:1])                 # warm the thread pool
    t0 = time.perf_counter()
    DP.dense_forward(P, d[:2], e[:2], z[:2], m[:2])
    per = (time.perf_counter() - t0) / 2
    n = int(max(2, min(n_sample, 12.0 / max(per, 1e-6))))
    t0 = time.perf_counter()
    loss, _ = DP.dense_forward(P, d[:n], e[:n], z[:n], m[:n])
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    n_lr = min(n, 8)
    O.forward(params, d[:n_lr].numpy(), e[:n_lr].numpy(), z[:n_lr].numpy(), m[:n_lr].numpy())
    dt_lr = (time.perf_counter() - t1) / n_lr
    return {
        "value": n / dt, "unit": "spectra/s", "cores": cores, "kind": "port",
        "sample": f"first {n} spectra of the rank-0 batch, dense O(Npix^3) torch-CPU float32 port of "
                  f"model.py:107-158 (oracle/dense_port.py), {dt:.1f} s wall, {cores} torch threads "
                  f"(host reports {os.cpu_count()} cpus, usable {usable_cores()})",
        "lowrank_oracle_spectra_per_s": 1.0 / dt_lr,
        "loss_finite": bool(np.isfinite(float(loss))),
        "loss_note": "the float32 port (like the reference, SURVEY App. B Q7) overflows det() to inf at N_h >= 16; the HIP "
                     "path takes log det from the pivots and stays finite; the timing is unaffected",
        "true_reference_8core_survey": REFERENCE_8CORE.get(config),
        "true_reference_note": "the imported reference's forward on the survey container's 8 cores at this shape "
                               "(SURVEY.md section 6); it cannot travel to the GPU box",
    }


def predict_leg(model, batch, mu, npix, nb, nh, seconds=1.0):
    """Throughput of QFA.predict (reference model.py:160-180, loop main.py:94-98) on the resident batch and the HBM
    roofline of its writer (k_predict_x at N_h <= 16, k_predict_x32 above): algorithmic bytes per spectrum = 4 (2 N_pix + k^2 + k + 1) out
    + 9 N_pix + 4 N_b in; the writer itself moves 8 N_pix bytes per spectrum (cont + unc)."""
    import numpy as np
    import torch
    d, e, z, m = batch
    B = d.shape[0]
    dev = d.device
    flux = d                                         # any float32 (B, Npix) works as raw flux for timing
    out = (torch.empty((B,), dtype=torch.float32, device=dev), torch.empty((B, nh), dtype=torch.float32, device=dev),
           torch.empty((B, nh, nh), dtype=torch.float32, device=dev),
           torch.empty((B, npix), dtype=torch.float32, device=dev), torch.empty((B, npix), dtype=torch.float32, device=dev))
    for _ in range(2):
        model.predict(flux, e, z, m, out=out)
    torch.cuda.synchronize()
    n, recs = 0, []
    t0 = time.perf_counter()
    while True:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for ev in evs:
            ev.record()
        model.predict(flux, e, z, m, events=evs, out=out)
        recs.append(evs)
        n += 1
        if n >= 3 and (n % 4 == 0):
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= seconds:
                break
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    st = np.array([[r[j].elapsed_time(r[j + 1]) for j in range(3)] for r in recs]).mean(axis=0)
    by_total = 4 * (2 * npix + nh * nh + nh + 1) + 9 * npix + 4 * nb
    by_writer = 8 * npix
    w_ms = float(st[2])
    return {"value": B / dt, "unit": "spectra/s", "ms_per_call": dt * 1e3, "spectra": B, "calls": n,
            "stage_ms": {"images_and_pass1": float(st[0]), "solve": float(st[1]), "writer": w_ms},
            "roofline": {"bound": "hbm", "kernel": "k_predict_x" if nh <= 16 else "k_predict_x32", "achieved": by_writer * B / (w_ms * 1e-3) / 1e9,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": by_writer * B / (w_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                         "alg_bytes_per_spectrum_writer": by_writer},
            "call_hbm_frac": (B / dt) * by_total / (PEAK_HBM_GBS * 1e9), "alg_bytes_per_spectrum": by_total}


def main():
    # The contract is ONE line on stdout.  RCCL prints a version banner to stdout when the first communicator is made
    # (and Gloo its connection report): everything but the result line goes to stderr, the result to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="spectra per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain", type=float, default=3.0, help="seconds of extra steps after the timed region (0 = skip)")
    ap.add_argument("--no-predict", action="store_true")
    ap.add_argument("--deterministic", action="store_true", help="fixed-order accumulation (model.deterministic)")
    ap.add_argument("--flags", type=lambda x: int(x, 0), default=0, help="QFA_F_* kernel-form flags (include/qfa_hip.h); 0 = defaults")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from qfa_amd import QFA, Adam, step_scheduler, synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    local %= max(1, torch.cuda.device_count())          # (fewer devices than ranks: a rehearsal on one GPU shares it)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("QFA_BENCH_FORCE_DIST") == "1"     # (the flag rehearses RCCL at N = 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL (backend "nccl" on ROCm); QFA_BENCH_BACKEND=gloo rehearses N > 1 with ranks that share one GPU (RCCL refuses
        # two ranks on a device): the packed buffer then goes through host memory, timings are not a scaling measurement
        backend = os.environ.get("QFA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    B, npix, nh, masks, n_cpu = CONFIGS[args.config]
    if args.batch:
        B = args.batch
    wav, nb, nr = synthetic.wavelength_grid(None if args.config == "c1" else npix)
    npix = len(wav)
    params, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
    cfg_index = {"c1": 1, "c2": 2, "c3": 3, "c5": 5}[args.config]
    # generate in slabs to bound temporary memory
    parts = []
    slab = 25000
    for i, s0 in enumerate(range(0, B, slab)):
        n = min(slab, B - s0)
        parts.append(synthetic.make_batch_torch(params, mu, wav, nb, n, 20220700 + cfg_index + 1000 * rank + 17 * i,
                                                dev, masks=masks))
    batch = tuple(torch.cat([p[j] for p in parts]) for j in range(4))
    del parts
    torch.cuda.empty_cache()

    model = QFA(nb, nr, nh, dev, model_params=params)
    model.deterministic = bool(args.deterministic)
    model.flags = args.flags
    model.mu = torch.tensor(mu, device=dev)
    if use_dist:
        model.enable_data_parallel()
    opt = Adam(model.parameters, dev, scheduler=step_scheduler(0.9, 10), learning_rate=1e-3, weight_decay=1e-1)

    for _ in range(args.warmup):
        model.step(opt, *batch)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
    for es in evs:
        for e in es:
            e.record()          # creates the underlying hipEvent_t; re-recorded by the library
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = []
    for i in range(args.steps):
        losses.append(model.step(opt, *batch, events=evs[i]))
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    stage = np.array([[es[j].elapsed_time(es[j + 1]) for j in range(4)] for es in evs]).mean(axis=0)   # ms
    ms_prep, ms_p1, ms_solve, ms_p2 = [float(x) for x in stage]

    # ---- sustained leg: the same step for >= --sustain seconds (held clock; lets rocm-smi sampling see the run)
    sustained = None
    if args.sustain > 0:
        n_s, t_s = 0, time.perf_counter()
        chunk = max(10, int(0.25 / max(dt / args.steps, 1e-4)))
        while time.perf_counter() - t_s < args.sustain:
            for _ in range(chunk):
                model.step(opt, *batch)
            torch.cuda.synchronize()
            n_s += chunk
        # last chunk again with stage events
        sev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(chunk)]
        for es in sev:
            for e in es:
                e.record()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t1 = time.perf_counter()
        for i in range(chunk):
            model.step(opt, *batch, events=sev[i])
        torch.cuda.synchronize()
        dts = time.perf_counter() - t1
        if use_dist:
            t = torch.tensor([dts], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        sst = np.array([[es[j].elapsed_time(es[j + 1]) for j in range(4)] for es in sev]).mean(axis=0)
        sustained = {"ms_per_step": dts / chunk * 1e3, "value": world * B * chunk / dts, "steps": chunk,
                     "after_seconds_of_load": time.perf_counter() - t_s,
                     "stage_ms": {"pf_image": float(sst[0]), "pass1_moments": float(sst[1]), "solve": float(sst[2]),
                                  "pass2_grads": float(sst[3])}}
    f1, f2 = alg_flops(npix, nh)
    by = alg_bytes(npix, nb)
    rate = world * B * args.steps / dt
    rate_gpu = B * args.steps / dt
    if nh > 16:
        p2_name = "k_s12_x+2*k_grads_s3"      # pass 2 at N_h = 17..32: three launches, timed together by the stage events
    else:
        p2_name = "k_grads_x" if 9 <= nh <= 16 and os.environ.get("QFA_PASS2_XDL", "1") != "0" else "k_grads"
    dominant = p2_name if ms_p2 >= ms_p1 else "k_moments"
    dom_ms, dom_flops = (ms_p2, f2) if dominant == p2_name else (ms_p1, f1)
    # the n k^2 (matrix-pipe) part of dom_flops and the bf16 products issued for it: pass 1 4 n k^2 x 6; pass 2
    # n k^2 (stage 1, diag Sigma^-1) x 6 + 2 n k^2 (stage 3, M Z) x 3 (QFA_S3_TERMS, qfa_common.h)
    nk2 = npix * nh * nh
    if dominant in ("k_grads_x", "k_s12_x+2*k_grads_s3"):
        xdl_flops = (6 * 1 + 3 * 2) * nk2
    elif dominant == "k_moments":
        xdl_flops = 6 * 4 * nk2
    else:
        xdl_flops = None          # k_grads (N_h <= 8): stage 1 on the float32 MFMA, no XDL roof to price against
    ach = dom_flops * B / (dom_ms * 1e-3) / 1e12
    traffic = None
    tfile = os.path.join(REPO, "profiles", f"traffic_{args.config}.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            if tj.get("B") == B:
                traffic = tj.get(dominant + "_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "spectra/sec per EM step", "value": rate, "unit": "spectra/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {B} spectra/GPU x N_pix={npix} (N_b={nb}), N_h={nh}, "
                               f"{'random pixel masks' if masks else 'no masks'}, becker tau, "
                               f"forward + {'RCCL all-reduce + ' if world > 1 else ''}Adam + clip",
                   "spectra_per_gpu": B, "n_pix": npix, "n_b": nb, "n_h": nh, "parallelism": f"dp{world}",
                   "arithmetic": "float32 throughout; pass 1 (N_h <= 16) and pass 2 (N_h = 9..16) issue their contractions as "
                                 "bf16 XDL MFMAs over operands split into three bf16 pieces (float32-exact split, float32 "
                                 "accumulate): six piece products per float32 product (error vs float64 at or below the f32 "
                                 "MFMA's, tools/ubench/bf16x3_numerics.hip), three over the two leading pieces in stage 3 "
                                 "of pass 2 (<= 3 x 2^-18 per product; the F gradient's error against the float64 oracle does not "
                                 "move, profiles/r2_accuracy.txt); k x k solve in float64"},
        "roofline": {"bound": "mfma", "kernel": dominant, "achieved": ach, "peak": PEAK_FP32_TFLOPS,
                     "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS, "traffic": traffic,
                     "kernel_ms": dom_ms, "alg_flops_per_spectrum": dom_flops,
                     # the roof the kernel is built against: contraction flops issued six-fold on the bf16 XDL pipe
                     "peak_xdl": PEAK_BF16_TFLOPS,
                     "achieved_xdl": xdl_flops * B / (dom_ms * 1e-3) / 1e12 if xdl_flops else None,
                     "frac_xdl": xdl_flops * B / (dom_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS if xdl_flops else None,
                     "xdl_flops_per_spectrum": xdl_flops,
                     "built_against": "xdl (bf16 MFMA: six piece products per float32 product, three in stage 3 of "
                                      "pass 2); frac = the survey's float32 roof"},
        "stage_ms": {"pf_image": ms_prep, "pass1_moments": ms_p1, "solve": ms_solve, "pass2_grads": ms_p2,
                     "rest_of_step": dt / args.steps * 1e3 - float(stage.sum())},
        "step_roofline": {"achieved_fp32_frac": rate_gpu * (f1 + f2) / (PEAK_FP32_TFLOPS * 1e12),
                          "achieved_hbm_frac": rate_gpu * by / (PEAK_HBM_GBS * 1e9),
                          "alg_flops_per_spectrum": f1 + f2, "alg_bytes_per_spectrum": by,
                          "binding_roof": "fp32 (VALU + MFMA); HBM ceiling at this config is "
                                          f"{100.0 * (PEAK_FP32_TFLOPS * 1e12 / (f1 + f2)) * by / (PEAK_HBM_GBS * 1e9):.1f}%"},
        "loss": float(losses[-1].item()),
    }
    if sustained is not None:
        out["sustained_ms_per_step"] = sustained["ms_per_step"]
        out["sustained"] = sustained
    if world == 1 and not args.no_predict:
        out["predict"] = predict_leg(model, batch, mu, npix, nb, nh)
        if args.config == "c3":
            # the shape the reference ships and predicts with (N_pix = 1913, N_h = 8: data/model_parameters.npz)
            wav1, nb1, nr1 = synthetic.wavelength_grid(None)
            p1, mu1 = synthetic.mock_parameters(len(wav1), nb1, 8, seed=20220700)
            b1 = synthetic.make_batch_torch(p1, mu1, wav1, nb1, 50000, 20220701, dev, masks=True)
            m1 = QFA(nb1, nr1, 8, dev, model_params=p1)
            m1.mu = torch.tensor(mu1, device=dev)
            out["predict_sdss_shape"] = dict(predict_leg(m1, b1, mu1, len(wav1), nb1, 8),
                                             workload="50000 spectra x N_pix=1913 (N_b=720), N_h=8, masks")
            del b1, m1
            torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(params, batch, n_cpu, npix, args.config)
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
