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

`python bench.py --gpus N` without a launcher starts the N ranks itself (fresh child processes, before this process
touches the GPU) and relays rank 0's line.  Rank 0 prints ONE JSON line.

`roofline` describes the dominant kernel (pass 2: at N_h <= 16 k_grads_t from 128 (N_h <= 8: 512) spectra on and k_grads_x below,
k_grads / k_grads_x at N_h <= 8, the three launches k_s12_x + 2 k_grads_s3 at N_h = 17..32; its mean duration is measured with HIP events recorded by the library on the
launch stream inside the timed region).  The contractions are float32 products ISSUED as 16-bit piece products on the XDL
pipe (pass 1 and stage 3 of pass 2: operands split into three bf16 pieces, six products per float32 product; stage 1 of pass 2
and of the writer, round 5: two float16 pieces, three products: DESIGN.md section 4), so `achieved` =
the issued 16-bit flops per launch / duration and `peak` = the dense bf16 / f16 MFMA peak (2.5 PFLOP/s); the algorithmic float32
flops against the float32 roof the survey names are kept as `achieved_alg_fp32` / `frac_vs_fp32_roof` (that fraction can
exceed 1: the kernel does not run on that pipe).  `step_roofline` holds the step's HBM side: algorithmic bytes, the
measured bytes (profiles/traffic_<config>.json, rocprofv3 PMC) and their ratio.  After the timed region the same step
runs for >= 3 s more (`sustained_*`: the clock the chip holds under seconds of this load, not a burst), then `QFA.predict`
is timed (`predict`: spectra/s and the HBM roofline of its output writer), then `cpu_baseline` times the dense O(N_pix^3)
CPU port of the reference's per-spectrum step (oracle/dense_port.py) on a bounded sample of the same batch (rank 0, N = 1).
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
    # BASELINE configs[3]: 1M spectra data-parallel over 8 GPUs = 125 000 per GPU of the c3 shape (the default of --gpus 8)
    "c4": (125000, 4000, 16, True, 32),
    "c1": (128, 1913, 8, True, 64),
    "c5": (20000, 8000, 32, True, 4),
    # the reference's own shapes: its default batch (QFA/config.py:32) on the SDSS grid, and its second shipped model
    # (data/model_parameters_desi.npz: N_pix = 9243, N_b = 2238, N_h = 8)
    "c1b": (500, 1913, 8, True, 64),
    "desi": (40000, 9243, 8, True, 4),
}
PEAK_FP32_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
PEAK_BF16_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (XDL) peak
PEAK_HBM_GBS = 8000.0
# What an MFMA + ds_read_b128 + two v_fma stream with no stall delivers once the chip has lowered its clock under the load
# (tools/ubench/xdl_power.hip mode 4 on one MI355X: 1.49 GHz; profiles/r5_ubench_xdl_power.txt).  Informational, not `peak`.
CAPPED_STREAM_TFLOPS = 1430.0
# the true reference (imported in the survey container, SURVEY.md section 6 / BASELINE.md section 2): forward, 8 cores
REFERENCE_8CORE = {"c1": 13.4, "c1b": 13.4, "c2": 12.2, "c3": 1.54, "c4": 1.54, "c5": 0.21}


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
    kp = 8 if nh <= 8 else (16 if nh <= 16 else 32)
    nks = 1 + (kp * (kp + 1) // 2 + 31) // 32               # K-steps of the writer's stage 1 ([hmean | hcov'] against [f | f_a f_b])
    w_tf = npix * nks * 3 * 16384 / 256 * B / (w_ms * 1e-3) / 1e12     # issued float16 piece-product flops (three MFMAs per K-step, 16 x 16 tile)
    return {"value": B / dt, "unit": "spectra/s", "ms_per_call": dt * 1e3, "spectra": B, "calls": n,
            "stage_ms": {"images_and_pass1": float(st[0]), "solve": float(st[1]), "writer": w_ms},
            "roofline": {"bound": "hbm", "kernel": "k_predict_x" if nh <= 16 else "k_predict_x32", "achieved": by_writer * B / (w_ms * 1e-3) / 1e9,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": by_writer * B / (w_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                         "alg_bytes_per_spectrum_writer": by_writer,
                         # the writer also runs stage 1 on the XDL pipe: at N_h > 16 that, not the stores, is what it waits for
                         "xdl": {"tflops": w_tf, "frac_of_peak": w_tf / PEAK_BF16_TFLOPS,
                                 "frac_of_clock_capped_stream": w_tf / CAPPED_STREAM_TFLOPS}},
            "call_hbm_frac": (B / dt) * by_total / (PEAK_HBM_GBS * 1e9), "alg_bytes_per_spectrum": by_total}


def epoch_leg(params, mu, wav, nb, nr, nh, B, masks, dev, seed, n_batches, epochs, use_graph, rank=0, world=1):
    """Throughput of whole training EPOCHS through the product's own loop: ``QFA.train`` over a ``DeviceDataloader`` that
    holds ``n_batches`` x B spectra resident (delta / mask built once, rows padded to 128 bytes), reshuffled every epoch
    (reference QFA/dataloader.py:154-167, QFA/model.py:204-215), every batch handed to the kernels as row numbers (ABI v3,
    qfa_amd/resident.py).  Includes everything an epoch costs: the shuffle and its upload, Adam, the scheduler step, the one
    host synchronisation per epoch.  The data set differs from the step legs' batch (same generator, other seeds)."""
    import tempfile
    import torch
    from qfa_amd import QFA, Adam, step_scheduler, synthetic
    from qfa_amd.dataloader import DeviceDataloader
    npix, N = len(wav), n_batches * B
    flux = torch.empty((N, npix), dtype=torch.float32, device=dev)
    error = torch.empty((N, npix), dtype=torch.float32, device=dev)
    zq = torch.empty((N,), dtype=torch.float32, device=dev)
    slab = 25000
    for i, s0 in enumerate(range(0, N, slab)):
        n = min(slab, N - s0)
        f, e, z = synthetic.make_batch_torch(params, mu, wav, nb, n, seed + 31 * i, dev, masks=masks, return_flux=True)
        flux[s0:s0 + n], error[s0:s0 + n], zq[s0:s0 + n] = f, e, z
        del f, e, z
    t_b = time.perf_counter()
    dl = DeviceDataloader(flux, error, zq, wav, B, dev, tau="becker", shuffle=True)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_b
    del flux, error
    torch.cuda.empty_cache()
    model = QFA(nb, nr, nh, dev, model_params=params)
    opt = Adam(model.parameters, dev, scheduler=step_scheduler(0.9, 10), learning_rate=1e-3, weight_decay=1e-1)
    big = 10 ** 9
    with tempfile.TemporaryDirectory() as td:
        model.train(opt, dl, 1, output_dir=td, save_interval=big, smooth_interval=big, quiet=True, use_graph=use_graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train(opt, dl, epochs, output_dir=td, save_interval=big, smooth_interval=big, quiet=True, use_graph=use_graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    steps = epochs * n_batches
    # one indexed step alone, with the library's stage events (a full batch of the current shuffled order)
    import numpy as np
    def one_step(rb):
        for _ in range(3):
            model.step(opt, batch=rb)
        sev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(10)]
        for es in sev:
            for e in es:
                e.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for es in sev:
            model.step(opt, batch=rb, events=es)
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / len(sev)
        sst = np.array([[es[j].elapsed_time(es[j + 1]) for j in range(4)] for es in sev]).mean(axis=0)
        return {"ms_per_step": dt1 * 1e3, "stage_ms": {"pf_image": float(sst[0]), "pass1_moments": float(sst[1]),
                                                        "solve": float(sst[2]), "pass2_grads": float(sst[3])}}
    dl.rewind()
    indexed = one_step(dl.next_batch_rows())
    # the same kernels on B consecutive rows in storage order: what the random order of a shuffled epoch costs by itself
    indexed["rows_in_storage_order"] = one_step(dl.rows_batch(0, B)[0])
    return {"value": world * N * epochs / dt, "indexed_step": indexed, "unit": "spectra/s", "ms_per_step": dt / steps * 1e3, "epochs": epochs,
            "batches_per_epoch": n_batches, "resident_spectra": N, "batch": B, "shuffled": True, "use_graph": bool(use_graph),
            "graph_steps_per_replay": 8 if use_graph else None,
            "row_stride": dl._stride, "resident_bytes": int(13 * dl._stride) * N, "loader_build_s": t_build,
            "path": "QFA.train -> DeviceDataloader.rewind / next_batch_rows -> QFA.step(batch=ResidentBatch): the kernels read "
                    "rows[s] x row_stride of the resident delta / error / mask (no per-batch kernel, copy or upload)"}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks as FRESH child processes
    (torch.distributed.run) before this process has touched the GPU, relay rank 0's single JSON line, exit with the
    launcher's status.  (Never an exec of a process that has initialised HIP.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{") and '"metric"' in ln]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    else:
        sys.stderr.write(r.stdout.decode(errors="replace")[-4000:])
    raise SystemExit(r.returncode if (r.returncode or lines) else 1)


def dry_run(args, real_stdout):
    """Launcher rehearsal without a GPU (tests/test_bench_launch.py): rendezvous, barrier, MAX all-reduce of the step time and
    the one-line protocol of the real run; no spectra are processed and the line says so."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        os.write(real_stdout, (json.dumps({"metric": "spectra/sec per EM step", "value": None, "dry_run": True,
                                           "n_gpus": world, "steps": args.steps if args.steps >= 0 else 20, "warmup": args.warmup if args.warmup >= 0 else 40,
                                           "max_over_ranks": float(t.item())}) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


def main():
    # The contract is ONE line on stdout.  RCCL prints a version banner to stdout when the first communicator is made
    # (and Gloo its connection report): everything but the result line goes to stderr, the result to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: >= 40 untimed steps / 0.25 s let the clocks settle; the first steps after the data generation ran 1-5 % slower than the
    # >= 3 s sustained leg of the same kernels in round 5's runs with 5)
    ap.add_argument("--steps", type=int, default=-1, help="timed steps; default: 20, or as many as 0.1 s of steps take (small batches)")
    ap.add_argument("--warmup", type=int, default=-1,
                    help="untimed steps in front of the timed ones; default: 40, or as many as 0.25 s of steps take (small batches)")
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: c3 (100 000 spectra per GPU); with --gpus 8: c4 (BASELINE configs[3]: 1M spectra over 8 GPUs)")
    ap.add_argument("--batch", type=int, default=0, help="spectra per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain", type=float, default=3.0, help="seconds of extra steps after the timed region (0 = skip)")
    ap.add_argument("--no-predict", action="store_true")
    ap.add_argument("--no-epoch", action="store_true", help="skip the epoch leg (QFA.train over a resident, reshuffled data set)")
    ap.add_argument("--epoch-batches", type=int, default=0, help="batches resident in the epoch leg (default: 4, more for small batches)")
    ap.add_argument("--epoch-graph", type=int, default=-1, help="epoch leg: replay the step as a hipGraph (default: batches <= 2048 spectra)")
    ap.add_argument("--deterministic", action="store_true", help="fixed-order accumulation (model.deterministic)")
    ap.add_argument("--flags", type=lambda x: int(x, 0), default=0, help="QFA_F_* kernel-form flags (include/qfa_hip.h); 0 = defaults")
    ap.add_argument("--dry-run", action="store_true", help="launcher rehearsal without a GPU: no spectra are processed")
    args = ap.parse_args()
    if args.config is None:
        args.config = "c4" if args.gpus >= 8 else "c3"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.dup2(real_stdout, 1)
        self_launch(args)
    if args.dry_run:
        return dry_run(args, real_stdout)

    import numpy as np
    import torch
    import torch.distributed as dist
    from qfa_amd import QFA, Adam, step_scheduler, synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running with {world} rank(s)", file=sys.stderr)
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
    if args.config == "desi":
        wav, nb, nr = synthetic.desi_grid()
    else:
        wav, nb, nr = synthetic.wavelength_grid(None if args.config in ("c1", "c1b") else npix)
    npix = len(wav)
    params, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
    cfg_index = {"c1": 1, "c2": 2, "c3": 3, "c4": 4, "c5": 5, "c1b": 11, "desi": 13}[args.config]
    # generate in slabs to bound temporary memory
    parts = []
    slab = 25000
    for i, s0 in enumerate(range(0, B, slab)):
        n = min(slab, B - s0)
        parts.append(synthetic.make_batch_torch(params, mu, wav, nb, n, 20220700 + cfg_index + 1000 * rank + 17 * i,
                                                dev, masks=masks, return_zq=True))
    batch = tuple(torch.cat([p[j] for p in parts]) for j in range(4))
    # the factored-z input form of the same batch (include/qfa_hip.h: 1 + zabs = zq1 x pix_ratio, QFA/dataloader.py:102)
    zfac = ((1.0 + torch.cat([p[4] for p in parts])).contiguous(),
            torch.tensor((wav[:nb] / synthetic.LYA).astype(np.float32), device=dev))
    del parts
    torch.cuda.empty_cache()

    model = QFA(nb, nr, nh, dev, model_params=params)
    model.deterministic = bool(args.deterministic)
    model.flags = args.flags
    model.mu = torch.tensor(mu, device=dev)
    if use_dist:
        model.enable_data_parallel()
    opt = Adam(model.parameters, dev, scheduler=step_scheduler(0.9, 10), learning_rate=1e-3, weight_decay=1e-1)

    if args.warmup < 0:
        # default: 40 steps, or 0.25 s worth of them when a step is short (c2, c1b: 40 steps are 4 - 7 ms, the clocks have not settled)
        for _ in range(3):
            model.step(opt, *batch)
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        for _ in range(5):
            model.step(opt, *batch)
        torch.cuda.synchronize()
        per = max((time.perf_counter() - t_w) / 5, 1e-6)
        args.warmup = int(max(40, min(5000, 0.25 / per)))
        if args.steps < 0:                                  # (20 steps of a 0.1-ms step are mostly the queue filling up)
            args.steps = int(max(20, min(2000, 0.1 / per)))
        if use_dist:                                        # (every rank the same counts)
            tw = torch.tensor([args.warmup, args.steps], dtype=torch.int64, device=dev)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            args.warmup, args.steps = int(tw[0].item()), int(tw[1].item())
    if args.steps < 0:
        args.steps = 20
    # (at least two untimed steps: the second sighting of the zabs tensor runs the one-time structure test of QFA.auto_factor_zabs)
    for _ in range(max(args.warmup, 2 if (model.auto_factor_zabs and nb > 0) else 0)):
        model.step(opt, *batch)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
    for es in evs:
        for e in es:
            e.record()          # creates the underlying hipEvent_t; re-recorded by the library
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ar_evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if use_dist else None
    t0 = time.perf_counter()
    losses = []
    for i in range(args.steps):
        if ar_evs is not None:
            model._ar_events = ar_evs[i]
        losses.append(model.step(opt, *batch, events=evs[i]))
    model._ar_events = None
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_rank = dt
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    stage = np.array([[es[j].elapsed_time(es[j + 1]) for j in range(4)] for es in evs]).mean(axis=0)   # ms
    ms_prep, ms_p1, ms_solve, ms_p2 = [float(x) for x in stage]
    # the one collective of the step (packed sums + counts, qfa_accum_floats floats) between events on the launch stream
    ms_ar = float(np.mean([a.elapsed_time(b) for a, b in ar_evs])) if ar_evs is not None else None

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
    # ---- the same step fed the factored-z input form (what DeviceDataloader batches carry): zabs is not read
    fz = None
    if nb > 0:
        for _ in range(2):
            model.step(opt, batch[0], batch[1], None, batch[3], zfac=zfac)
        fev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
        for es in fev:
            for e in es:
                e.record()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            model.step(opt, batch[0], batch[1], None, batch[3], events=fev[i], zfac=zfac)
        torch.cuda.synchronize()
        dtf = time.perf_counter() - t1
        if use_dist:
            t = torch.tensor([dtf], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtf = float(t.item())
        fst = np.array([[es[j].elapsed_time(es[j + 1]) for j in range(4)] for es in fev]).mean(axis=0)
        fz = {"ms_per_step": dtf / args.steps * 1e3, "value": world * B * args.steps / dtf,
              "alg_bytes_per_spectrum": npix * 9 + 8,
              "stage_ms": {"pf_image": float(fst[0]), "pass1_moments": float(fst[1]), "solve": float(fst[2]),
                           "pass2_grads": float(fst[3])},
              "note": "qfa_batch_t::zq1 / pix_ratio instead of zabs (B, Nb): 4 Nb bytes per spectrum and pass less, two "
                      "transcendentals per blue element instead of five; same results to float32 rounding "
                      "(tests/test_hip_parity.py::test_factored_z_input_form_matches_zabs_form_and_oracle)"}
    # ---- the same tensors with QFA.auto_factor_zabs off: the kernels that READ zabs (every step of rounds 1-4's `value`)
    zk = None
    auto_on = bool(model.auto_factor_zabs) and nb > 0
    ent = model._zf_seen.get(id(batch[2])) if auto_on else None
    factored_headline = bool(ent is not None and isinstance(ent[2], tuple))
    if auto_on:
        model.auto_factor_zabs = False
        for _ in range(2):
            model.step(opt, *batch)
        zev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
        for es in zev:
            for e in es:
                e.record()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            model.step(opt, *batch, events=zev[i])
        torch.cuda.synchronize()
        dtz = time.perf_counter() - t1
        if use_dist:
            t = torch.tensor([dtz], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtz = float(t.item())
        model.auto_factor_zabs = True
        zst = np.array([[es[j].elapsed_time(es[j + 1]) for j in range(4)] for es in zev]).mean(axis=0)
        # what the structure test costs when it runs (once per repeated tensor): HIP events around the entry point
        import ctypes as C
        from qfa_amd import _lib as _L
        zq_t, rt_t = torch.empty(B, device=dev), torch.empty(nb, device=dev)
        nbad_t = torch.empty(1, dtype=torch.int32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        chk = lambda: _L.check(_L.lib().qfa_zabs_factor_f32(C.c_void_p(batch[2].data_ptr()), B, nb, 4e-7, C.c_void_p(zq_t.data_ptr()),
                                                           C.c_void_p(rt_t.data_ptr()), C.c_void_p(nbad_t.data_ptr()),
                                                           _L.current_stream(dev)), "qfa_zabs_factor_f32")
        chk()
        e0.record(); chk(); e1.record()
        torch.cuda.synchronize()
        zk = {"ms_per_step": dtz / args.steps * 1e3, "value": world * B * args.steps / dtz,
              "stage_ms": {"pf_image": float(zst[0]), "pass1_moments": float(zst[1]), "solve": float(zst[2]), "pass2_grads": float(zst[3])},
              "structure_test_ms": e0.elapsed_time(e1), "structure_test_bad_elements": int(nbad_t.item()),
              "note": "QFA.auto_factor_zabs = False: pass 1 and pass 2 read zabs (B, Nb) and evaluate five transcendentals per blue "
                      "element; `value` of this line in rounds 1-4.  structure_test_ms: one qfa_zabs_factor_f32 call on this batch "
                      "(run once, at the second sighting of a tensor; not part of a timed step)"}
    # ---- SURVEY 8(d) "also a run at random_init_func values" (QFA/model.py:67-72: F ~ U(-0.5, 0.5), Psi = omega = 1, tau0 0.02,
    # c0 0.3, beta 2): the same batch through a freshly initialised model
    rinit = None
    if world == 1 and args.config in ("c3", "c4", "c2"):
        torch.manual_seed(20220700)
        m2 = QFA(nb, nr, nh, dev)
        m2.flags, m2.deterministic = args.flags, bool(args.deterministic)
        opt2 = Adam(m2.parameters, dev, scheduler=step_scheduler(0.9, 10), learning_rate=1e-3, weight_decay=1e-1)
        for _ in range(3):
            l2 = m2.step(opt2, *batch)
        rev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
        for es in rev:
            for e in es:
                e.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            l2 = m2.step(opt2, *batch, events=rev[i])
        torch.cuda.synchronize()
        dtr = time.perf_counter() - t1
        rst = np.array([[es[j].elapsed_time(es[j + 1]) for j in range(4)] for es in rev]).mean(axis=0)
        rinit = {"ms_per_step": dtr / args.steps * 1e3, "value": B * args.steps / dtr, "loss": float(l2.item()),
                 "stage_ms": {"pf_image": float(rst[0]), "pass1_moments": float(rst[1]), "solve": float(rst[2]), "pass2_grads": float(rst[3])},
                 "parameters": "random_init_func (reference QFA/model.py:57-72), three Adam steps in"}
        del m2, opt2
    f1, f2 = alg_flops(npix, nh)
    by = alg_bytes(npix, nb)
    rate = world * B * args.steps / dt
    rate_gpu = B * args.steps / dt
    fl = int(args.flags)
    fast = bool(fl & 0x4)                     # QFA_F_S3_FAST
    if nh > 16:
        p2_name = "k_s12_x+2*k_grads_s3"      # pass 2 at N_h = 17..32: three launches, timed together by the stage events
    else:                                     # the same rule as pass2_use_xdl (qfa_host.h)
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        xdl_form = not (fl & 0x1)
        # ... and pass2_use_pixres: the pixel-resident form (k_grads_t) from 128 (N_h <= 8: 512) spectra on for N_pix >= 1024, else 96 per CU
        auto_t = B >= 96 * ncu or (npix >= 1024 and B >= (512 if nh <= 8 else 128))      # qfa_host.h, pass2_use_pixres
        pixres = xdl_form and npix >= 16 and not (fl & (0x1 | 0x10 | 0x4)) and (bool(fl & 0x40) or (not (fl & 0x2) and auto_t))
        p2_name = ("k_grads_t" if pixres else "k_grads_x") if xdl_form else "k_grads"
    dominant = p2_name if ms_p2 >= ms_p1 else "k_moments_x"
    dom_ms, dom_flops = (ms_p2, f2) if dominant == p2_name else (ms_p1, f1)
    # The contractions are ISSUED as bf16 piece products on the XDL pipe: per spectrum
    #   pass 1                 4 n k^2 x 6
    #   pass 2                 n k^2 x 3 (stage 1, diag Sigma^-1: TWO float16 pieces per operand, three products -- round 5)
    #                          + 2 n k^2 x 6 (stage 3, M Z: three bf16 pieces, six products; x 3 with QFA_F_S3_FAST)
    #                            k_grads_t and k_grads_s3 (round 5): stage 3 on two float16 pieces as well: 2 n k^2 x 3
    # `roofline` prices the issued 16-bit MFMA flops of the dominant kernel against the dense bf16 / f16 MFMA peak (the same
    # 2.5 PF).  Until round 5 pass 2 issued six bf16 products everywhere (18 n k^2; k_grads_t now 9 n k^2): the step got faster by
    # issuing FEWER flops, so `frac` of this round is not comparable with the earlier rounds' (`frac_vs_fp32_roof` and ms_per_step are).
    nk2 = npix * nh * nh
    s3 = 3 if fast else 6
    if dominant in ("k_grads_t", "k_s12_x+2*k_grads_s3"):
        xdl_flops = (3 * 1 + (3 if s3 == 6 else s3) * 2) * nk2      # (k_grads_s3<32, 6> runs three float16 products; the FAST form three bf16)
    elif dominant == "k_grads_x":
        xdl_flops = (3 * 1 + s3 * 2) * nk2
    elif dominant == "k_moments_x":
        xdl_flops = 6 * 4 * nk2
    else:
        xdl_flops = None          # k_grads (N_h <= 8): stage 1 on the float32 MFMA -- priced against the float32 roof
    ach32 = dom_flops * B / (dom_ms * 1e-3) / 1e12
    traffic = traffic_step = None
    traffic_src = None
    tfile = os.path.join(REPO, "profiles", f"traffic_{args.config}.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            if tj.get("B") == B:
                # the byte counts are REPLAYED from the builder's PMC run (rocprofv3 cannot run inside this process): say so,
                # and say whether that run measured the library loaded here
                import hashlib
                from qfa_amd import _lib as _L
                sha = hashlib.sha256(open(_L.LIB_PATH, "rb").read()).hexdigest()
                traffic_src = {"traffic_source": f"profiles/traffic_{args.config}.json (builder's rocprofv3 --pmc run, tools/profile_round.sh)",
                               "traffic_stale": tj.get("libqfa_hip_sha256") != sha,
                               "traffic_measured_on_sha256": tj.get("libqfa_hip_sha256"), "libqfa_hip_sha256": sha}
                zs_ = "_zfac" if factored_headline else ""
                p2key = p2_name
                traffic = tj.get((("k_moments_x" + zs_) if factored_headline else "k_moments") + "_hbm_bytes_per_launch") if dominant == "k_moments_x" \
                    else tj.get(p2key + zs_ + "_hbm_bytes_per_launch")
                parts = [tj.get(kk + "_hbm_bytes_per_launch") for kk in (("k_moments_x_zfac" if factored_headline else "k_moments"), "k_solve", p2key + zs_)]
                traffic_step = sum(parts) if all(x is not None for x in parts) else None
                if zk is not None:
                    zparts = [tj.get(kk + "_hbm_bytes_per_launch") for kk in ("k_moments", "k_solve", p2key)]
                    if all(x is not None for x in zparts):
                        zk["measured_hbm_bytes_per_step"] = sum(zparts)
                        zk["traffic_ratio"] = sum(zparts) / (by * B)
                p2z = p2key + "_zfac"
                if fz is not None and all((kk + "_hbm_bytes_per_launch") in tj for kk in ("k_moments_x_zfac", p2z, "k_solve")):
                    fz["measured_hbm_bytes_per_step"] = sum(tj[kk + "_hbm_bytes_per_launch"] for kk in ("k_moments_x_zfac", p2z, "k_solve"))
                    fz["traffic_ratio"] = fz["measured_hbm_bytes_per_step"] / (fz["alg_bytes_per_spectrum"] * B)
        except Exception:
            traffic = None
    if xdl_flops:
        ach = xdl_flops * B / (dom_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": dominant, "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "kernel_ms": dom_ms,
                "flops_per_spectrum": xdl_flops,
                "priced": "16-bit piece-product flops issued on the XDL pipe (pass 1; stage 3 of k_grads_x: three bf16 pieces, "
                          "six products per float32 product; stage 1 of pass 2 and stage 3 of k_grads_t / k_grads_s3: two float16 pieces, three products"
                          + (", three in stage 3: QFA_F_S3_FAST" if fast else "")
                          + ") over the dense bf16 MFMA peak",
                "alg_flops_per_spectrum": dom_flops, "achieved_alg_fp32": ach32,
                "frac_vs_fp32_roof": ach32 / PEAK_FP32_TFLOPS,
                "frac_note": "`frac` counts ISSUED piece products: the round-5 kernels issue 9 (k_grads_t, k_grads_s3) or 15 n k^2 where "
                             "rounds 1-4 issued 18 for the same float32 result, so `frac` fell while the kernel got faster; compare rounds "
                             "by kernel_ms / ms_per_step or by frac_vs_fp32_roof (algorithmic flops, DESIGN.md section 5)",
                "clock_capped_stream": {"tflops": CAPPED_STREAM_TFLOPS, "frac": ach / CAPPED_STREAM_TFLOPS,
                                        "source": "profiles/r5_ubench_xdl_power.txt (mode 4: the chip holds 1.49 GHz, not 2.4, "
                                                  "under a stall-free MFMA + LDS + VALU stream on random operands)"}}
    else:
        roof = {"bound": "mfma", "kernel": dominant, "achieved": ach32, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": ach32 / PEAK_FP32_TFLOPS, "traffic": traffic, "kernel_ms": dom_ms,
                "flops_per_spectrum": dom_flops, "priced": "algorithmic float32 flops over the float32 MFMA / VALU peak",
                "alg_flops_per_spectrum": dom_flops}
    if traffic_src:
        roof.update(traffic_src)
    hbm_frac = rate_gpu * by / (PEAK_HBM_GBS * 1e9)
    step_ms = dt / args.steps * 1e3
    out = {
        "metric": "spectra/sec per EM step", "value": rate, "unit": "spectra/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {B} spectra/GPU x N_pix={npix} (N_b={nb}), N_h={nh}, "
                               f"{'random pixel masks' if masks else 'no masks'}, becker tau, "
                               f"forward + {'RCCL all-reduce + ' if world > 1 else ''}Adam + clip",
                   "spectra_per_gpu": B, "n_pix": npix, "n_b": nb, "n_h": nh, "parallelism": f"dp{world}",
                   "input_form": ("the reference's forward signature (delta, error, zabs, mask: QFA/model.py:74).  The zabs tensor came back "
                                  "unchanged, was tested ONCE for the structure the reference's loader gives it (1 + zabs = (1 + z_qso) wav / "
                                  "1215.67, QFA/dataloader.py:102; qfa_zabs_factor_f32, during warm-up) and is served by the factored-z "
                                  "kernels since (QFA.auto_factor_zabs; `zabs_kernels` = the same steps with that switched off)")
                                 if factored_headline else "the reference's forward signature (delta, error, zabs, mask); the kernels read zabs",
                   "arithmetic": "float32 throughout; the contractions are issued as 16-bit XDL MFMAs with float32 accumulate: pass 1 "
                                 "(and stage 3 of the small-batch kernel) over operands split exactly into three bf16 pieces, "
                                 "six piece products per float32 product; stage 1 of pass 2 and of the posterior writer and stage 3 of "
                                 "k_grads_t / k_grads_s3 over two float16 pieces scaled by powers of two, three products (error vs float64 at or below the f32 "
                                 "MFMA's for both, tools/ubench/bf16x3_numerics.hip; gradients vs the float64 oracle unchanged, "
                                 "profiles/r5_ab_f16_stage1.txt)"
                                 + (" EXCEPT stage 3 of pass 2, run here with three (--flags 0x4: operands carried to ~17 bits)" if fast else "")
                                 + "; k x k solve and scalar-gradient sums in float64", "flags": fl},
        "roofline": roof,
        "stage_ms": {"pf_image": ms_prep, "pass1_moments": ms_p1, "solve": ms_solve, "pass2_grads": ms_p2,
                     "rest_of_step": dt / args.steps * 1e3 - float(stage.sum()) - (ms_ar or 0.0),
                     **({"allreduce": ms_ar, "allreduce_bytes": 4 * (npix * nh + 3 * npix + nb + 8),
                         "allreduce_backend": os.environ.get("QFA_BENCH_BACKEND", "nccl")} if ms_ar is not None else {})},
        "step_roofline": {"achieved_hbm_frac": hbm_frac, "achieved_hbm_GBs": rate_gpu * by / 1e9,
                          "alg_bytes_per_spectrum": by, "alg_flops_per_spectrum": f1 + f2,
                          "achieved_fp32_frac": rate_gpu * (f1 + f2) / (PEAK_FP32_TFLOPS * 1e12),
                          # measured HBM bytes of the step's kernels (profiles/traffic_<config>.json: rocprofv3 PMC) over the
                          # algorithmic bytes: > 1 = re-reads (both passes read the spectra: the k x k solve sits between)
                          "measured_hbm_bytes_per_step": traffic_step,
                          "traffic_ratio": traffic_step / (by * B) if traffic_step else None,
                          "measured_hbm_frac": traffic_step / (step_ms * 1e-3) / (PEAK_HBM_GBS * 1e9) if traffic_step else None,
                          "binding_roof": "XDL issue / latency (see roofline.frac) and HBM together: the step moves "
                                          f"{hbm_frac * 100:.1f}% of HBM peak in algorithmic bytes"
                                          + (f" and {traffic_step / (step_ms * 1e-3) / (PEAK_HBM_GBS * 1e9) * 100:.0f}% in measured bytes" if traffic_step else "")},
        "loss": float(losses[-1].item()),
    }
    if fz is not None:
        out["factored_z"] = fz
    if zk is not None:
        out["zabs_kernels"] = zk
    if rinit is not None:
        out["random_init_values"] = rinit
    # who ran: the collective's own view of the job, and every rank's clock (the line's value uses the slowest)
    rk = {"rank": rank, "device": int(torch.cuda.current_device()), "device_name": torch.cuda.get_device_name(dev),
          "ms_per_step": dt_rank / args.steps * 1e3, "stage_ms": [ms_prep, ms_p1, ms_solve, ms_p2]}
    if use_dist:
        allr = [None] * world
        dist.all_gather_object(allr, rk)
        out["ranks"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "per_rank": allr,
                        "max_over_ranks_ms": max(r["ms_per_step"] for r in allr), "min_over_ranks_ms": min(r["ms_per_step"] for r in allr)}
    else:
        out["ranks"] = {"world_size": 1, "backend": None, "per_rank": [rk], "max_over_ranks_ms": rk["ms_per_step"],
                        "min_over_ranks_ms": rk["ms_per_step"]}
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
    if world == 1 and not args.no_epoch:
        nbat = args.epoch_batches or max(4, -(-131072 // B))
        graph = (B <= 2048) if args.epoch_graph < 0 else bool(args.epoch_graph)
        ep_epochs = max(2, min(50, int(1.0 / max(nbat * step_ms * 1e-3, 1e-3))))
        ep = epoch_leg(params, mu, wav, nb, nr, nh, B, masks, dev, 20220900 + cfg_index, nbat, ep_epochs, graph)
        ep["vs_step_only"] = ep["value"] / rate
        if fz is not None:
            ep["vs_step_only_factored_z"] = ep["value"] / fz["value"]
        out["epoch"] = ep
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(params, batch, n_cpu, npix, args.config)
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
