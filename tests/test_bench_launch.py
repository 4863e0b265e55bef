"""bench.py's launch contract without a GPU: `python bench.py --gpus N` (no launcher) starts its N ranks itself as fresh
child processes and relays exactly one JSON line; under torch.distributed.run the ranks rendezvous on 127.0.0.1.  The
`--dry-run` rehearsal does the rendezvous, the barriers and the MAX all-reduce of the real run over gloo and processes
no spectra (its line says so); the real multi-rank step is covered by tests/test_data_parallel.py."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=300)
    return r.returncode, r.stdout.decode(), r.stderr.decode()


def test_bench_self_launches_two_ranks_and_prints_one_line():
    rc, out, err = _run(["--gpus", "2", "--config", "c1", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert rc == 0, err[-2000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["dry_run"] is True
    assert d["max_over_ranks"] == 2.0                      # MAX over ranks of (1 + rank)


def test_bench_single_rank_dry_run_needs_no_launcher():
    rc, out, err = _run(["--config", "c1", "--dry-run"])
    assert rc == 0, err[-2000:]
    d = json.loads(out.strip())
    assert d["n_gpus"] == 1 and d["dry_run"] is True


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_over_gloo_prints_one_line():
    """The real step under `python bench.py --gpus 2` (self-launched): two ranks share the one GPU of the box, the packed
    buffer is all-reduced over gloo (RCCL refuses two ranks on one device) -- a protocol check, not a scaling number."""
    rc, out, err = _run(["--gpus", "2", "--config", "c1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                         "--no-predict", "--sustain", "0"], {"QFA_BENCH_BACKEND": "gloo"})
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["value"] > 0 and d["scaling"] == "weak"
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d
