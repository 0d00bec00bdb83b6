"""CPU: bench.py's own multi-process control flow -- self-spawn under torch.distributed.run, rendezvous on 127.0.0.1, barrier,
max over ranks, ONE JSON line from rank 0 -- through `--dry-run` (gloo, a stand-in step: rank r sleeps r + 1 ms).  The real run
needs one GPU per rank; this covers the launcher path the driver uses for N = 2, 4, 8."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_self_spawn_two_ranks_one_line_max_over_ranks():
    d = _run(["--gpus", "2", "--steps", "20", "--warmup", "2", "--dry-run"])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 2 and d["scaling"] == "weak"
    # rank 0 sleeps 1 ms per step, rank 1 2 ms: the reported time is the slower rank's
    assert 2.0 <= d["ms_per_step"] <= 4.0, d
    assert d["per_rank_ms_per_step"]["min"] <= d["per_rank_ms_per_step"]["max"]      # (both ranks time between the same two barriers: equal up to rounding)
    assert abs(d["value"] - 64 * 2 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]


def test_launcher_invocation_as_the_driver_does():
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ..."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run"],
                         capture_output=True, text=True, timeout=300, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_single_process_dry_run():
    d = _run(["--dry-run", "--steps", "5", "--warmup", "1"])
    assert d["n_gpus"] == 1 and 0.9 <= d["ms_per_step"] <= 3.0


def test_eight_rank_launch_as_the_driver_will_run_it():
    """`bench.py --gpus 8` -- the configuration BASELINE.json's config 4 names and no round has had hardware for -- through its own
    self-spawn path with eight gloo ranks: rendezvous, barrier, max over eight ranks, one JSON line whose value is the whole job's
    (8 x 64 frame-pairs per step), per-rank spread reported.  (Rank r sleeps r + 1 ms per step: the slowest rank, 8 ms, is the step.)"""
    d = _run(["--gpus", "8", "--steps", "5", "--warmup", "1", "--dry-run"])
    assert d["n_gpus"] == 8 and d["steps"] == 5 and d["scaling"] == "weak"
    assert 8.0 <= d["ms_per_step"] <= 14.0, d
    assert d["per_rank_ms_per_step"]["min"] <= d["per_rank_ms_per_step"]["max"]
    assert abs(d["value"] - 64 * 8 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    assert d["config"]["global_batch"] == 512


def test_round6_evidence_helpers():
    """bench.py's evidence helpers that need no GPU: the rocprofv3 summary lookup behind roofline.profile_frac (the newest committed
    profiles/rNN_bench_default_kernel_stats.txt), the whole-host CPU baseline (P single-thread processes over one window; here P = 2,
    2 s) and the stdout guard that keeps RCCL's banner out of the one JSON line."""
    sys.path.insert(0, ROOT)
    import bench
    us, name = bench.profile_kernel_avg_us("cost_volume_split_kernel<false>")
    assert name and name.startswith("r") and 100.0 < us < 1000.0, (us, name)
    assert bench.profile_kernel_avg_us("no_such_kernel") == (None, None)
    r = bench.cpu_throughput_baseline(256, 2, seconds=2.0, startup_s=60.0)
    assert r["cores"] == 2 and r["processes_failed"] == 0 and r["value"] > 0.5, r
    out = subprocess.run([sys.executable, "-c", "import os, sys; sys.path.insert(0, %r); import bench\n"
                          "import ctypes\nwith bench._quiet_stdout():\n    os.write(1, b'banner\\n'); ctypes.CDLL(None).puts(b'buffered C stdio banner')\n"
                          "print('{\"ok\": 1}')" % ROOT],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == '{"ok": 1}', (out.stdout, out.stderr[-500:])
