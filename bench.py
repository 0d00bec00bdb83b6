#!/usr/bin/env python3
"""Headline benchmark: radar frame-pairs/s of the RaTrack backbone forward (eval) at B=64, N=256 per GPU.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under
                                                             torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one Track4D.backbone() pass (PointNet++ MSG encoder over both frames, kNN cost volume,
decoder PNHead, GRU, scene-flow + motion-segmentation heads) over one synthetic batch of B frame-pairs
already resident in HBM.  One process per GPU; frame-pairs are independent, so ranks run disjoint
batches with no data-path collective (weak scaling); the timed region is bracketed by a barrier +
synchronize on both sides and the maximum over ranks is reported.

The timed region is EXACTLY `steps` steps with steps = max(K asked for, the steps HEADLINE_SECONDS = 11 s take) -- a K = 20 window is
15 ms, which nothing outside this process can check (round 4's 1.5 s window was invisible to a 10 s utilisation sampler too); the
K-step window itself is reported as `k_step_window`.  NBATCH distinct resident batches rotate
through the steps (no step re-reads its predecessor's inputs).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  configs             BASELINE.json's other single-GPU forward configurations with the same method: config 2 (B=32, N=256) and
                      config 5 (B=32, N=1024): ms/step, pairs/s, the dominant kernel alone against the split matrix peak;
  roofline            the dominant kernel (cost_volume_split_kernel) against its matrix peak (dense fp16 / 3 products per fp32
                      product; the fp32-input MFMA kernel it replaced is the tests' comparison implementation); duration measured IN SITU
                      (the individual event timings go to profiles/ when --evidence NAME is given):
                      a second timed region replays the same pipelined workload with the graph split around the kernel,
                      which is launched eagerly between two HIP events on its own launch stream;
  whole_path          pairs/s against both rooflines (HBM on SURVEY's algorithmic bytes, fp32 on the reference
                      formulation's FLOPs and on the FLOPs this design executes, counted from the launch shapes at run time);
  train               BASELINE config 3: the captured train step (forward + multi-task loss + backward + gradient
                      all-reduce + Adam) at the same B, N, with the roofline of ITS dominant kernel (cost_volume_bwd_kernel);
  roofline_irregular  FPS / ball query / three-NN / kNN / gather-scatter gradients: algorithmic bytes / live duration vs 8 TB/s;
  cpu_baseline        the CPU oracle (oracle/track4d_ref.py: C restatement of the native ops + PyTorch-CPU dense layers)
                      on this box's host cores, a bounded sample (~20 s): B=1 x {1, all} threads, B=32 x 32 threads, medians; and
                      `throughput`: P = min(64, host CPUs) single-thread processes looping the B=1 forward over one 8 s window.
Round 6: roofline.profile_frac (the same FLOPs / the kernel's average in the newest committed rocprofv3 summary of this bench, next
to the in-situ frac), kernels_alone as the median of 7 eager passes, train.allreduce_us_1rank (the gradient bucket through a 1-rank
RCCL group), roofline_irregular with the product path's two geometry launches.

`--mode train` makes the train step the headline line instead (same fields).
"""
import argparse
import json
import os
import socket
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per frame-pair forward (fp32, weights excluded)
ALG_BYTES_PER_PAIR = {256: 14154240, 1024: 25293312}
ALG_FLOPS_PER_PAIR = {256: 4.003e9, 1024: 10.790e9}
FP32_PEAK_TFLOPS = 157.3      # MI355X fp32 vector == fp32-input MFMA peak (MI355X_MICROARCH.md)
BF16_PEAK_TFLOPS = 2500.0     # dense bf16 = fp16 MFMA peak (MI355X_MICROARCH.md)
# the split path (csrc/split_mfma.h) pays THREE fp16 MFMA products for one fp32 product (rounds 2-4: six bf16 products): its matrix
# roofline in fp32-equivalent terms
SPLIT_PRODUCTS = 3
SPLIT_PEAK_TFLOPS = BF16_PEAK_TFLOPS / SPLIT_PRODUCTS
DTYPE = "f32 (2xfp16 split MFMA: three fp16 products per fp32 product, fp32 accumulate, power-of-two scales per matrix and position)"
HEADLINE_SECONDS = 11.0       # the headline timed region (a utilisation sampler outside the process must see it)
CONFIG_SECONDS = 3.0          # configs 2 and 5
HBM_PEAK_GBS = 8000.0
NBATCH = 8                    # distinct synthetic batches resident in HBM, rotated through the timed steps
PMC = {"forward": "pmc_cost_volume.json", "train": "pmc_cost_volume_bwd.json", "irregular": "irregular_hbm.json"}      # profiles/rNN_<name>
ROUNDS = ["r%02d" % r for r in range(6, 0, -1)]                                                                       # newest first


def train_roofline(a, kms, kflops, ach, pm, ms_step):
    """Roofline object of the train step's dominant kernel.  fp32-input MFMA kernel (train_ops.CV_SPLIT = False, tests only): matrix-bound against 157.3
    TFLOP/s.  Split kernel: its matrix floor (flops / (2500 / 3) TFLOP/s) is far below its HBM floor, so HBM is the binding
    roofline: algorithmic bytes = per (point, neighbour) position a3 + two mask words read, dz1, dz2, dz3, dq3, d4, dt2 written
    (5232 B), per query point dout read, dp1, dpd, bias rows written (7168 B)."""
    from ratrack_amd import train_ops
    common = {"traffic": pm["traffic_bytes_per_launch"] if pm else None, "traffic_source": (pm or {}).get("source"), "kernel_ms": round(kms, 4),
              "flops_per_launch": kflops,
              "share_of_step": round(kms / ms_step, 3)}
    if not train_ops._cv_split(a.batch * a.npoints):
        return dict({"kernel": "cost_volume_bwd_kernel", "bound": "mfma", "achieved": round(ach, 2), "peak": FP32_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(ach / FP32_PEAK_TFLOPS, 4)}, **common)
    m = a.batch * a.npoints * 16
    nbytes = m * 5232 + (m // 16) * 7168
    gbs = nbytes / (kms * 1e-3) / 1e9
    return dict({"kernel": "cost_volume_bwd_split_kernel", "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_launch": nbytes,
                 "mfma": {"achieved": round(ach, 2), "peak": round(SPLIT_PEAK_TFLOPS, 1), "unit": "TFLOP/s (fp32-equivalent; fp16 peak / 3)",
                          "frac": round(ach / SPLIT_PEAK_TFLOPS, 4), "vs_fp32_mfma_peak": round(ach / FP32_PEAK_TFLOPS, 4)}}, **common)


def cost_volume_flops_per_pair(n, k=16):
    """Work of the stage the dominant kernel implements, counted in the reference's own arithmetic
    (model_utils.py:226-236; SURVEY.md Appendix B 'stage 1' minus the layer-1 feature part, which this
    design evaluates per point in separate launches): per (point, neighbour) pair
    layers 2+3 (2 x 256 x 256) + layer-1 direction term (3 x 256) + WeightNet (2136) + weighted sum (256) MACs."""
    macs = k * n * (2 * 256 * 256 + 3 * 256 + 2136 + 256)
    return 2.0 * macs


def _pmc(kind, batch, n):
    """Fabric-side bytes per launch from the committed PMC passes (profiles/, same workload only)."""
    try:
        if batch == 64 and n == 256:
            for name in ("%s_%s" % (r, PMC[kind]) for r in ROUNDS):
                p = os.path.join(ROOT, "profiles", name)
                if os.path.exists(p):
                    d = json.load(open(p))
                    d["source"] = "profiles/" + name
                    return d
    except Exception:
        pass
    return None


def _train_total_traffic(a):
    """HBM-side bytes of the whole train step from the committed all-kernel PMC pass (tools/pmc_train_total.py) and their ratio to
    SURVEY 8(d)'s 3 x forward algorithmic bytes."""
    name = next((n for n in ("%s_pmc_train_total.json" % r for r in ROUNDS) if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
    if a.batch != 64 or a.npoints != 256 or name is None:
        return {}
    p = os.path.join(ROOT, "profiles", name)
    try:
        d = json.load(open(p))
        return {"traffic_bytes_per_step": d["bytes_per_step"], "traffic_ratio": d["traffic_ratio"],
                "traffic_source": "profiles/%s (rocprofv3 --pmc passes over every kernel of the step)" % name}
    except Exception:
        return {}


def profile_kernel_avg_us(pattern, stats="bench_default_kernel_stats.txt"):
    """Average duration of the kernel whose name contains `pattern` in the newest committed rocprofv3 summary (profiles/rNN_<stats>, written
    by tools/prof_summary.py from `rocprofv3 --kernel-trace --stats` of this bench) -> (us, file name) or (None, None)."""
    for r in ROUNDS:
        p = os.path.join(ROOT, "profiles", "%s_%s" % (r, stats))
        if not os.path.exists(p):
            continue
        try:
            for line in open(p):
                if pattern in line:
                    f = line[78:].split()
                    return float(f[2]), os.path.basename(p)          # calls, total_us, avg_us, ...
        except Exception:
            pass
    return None, None


class _quiet_stdout:
    """RCCL prints a version banner on file descriptor 1 when its first communicator comes up; this file's contract is ONE JSON line on
    stdout.  Sends fd 1 to /dev/null for the duration (Python-level prints inside are lost too: there are none)."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        self._null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self._null, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:      # the banner is written with C stdio, which buffers when fd 1 is a pipe or a file: flush it while fd 1 is still /dev/null
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        os.close(self._null)
        return False


def allreduce_1rank_us(nfloats, dev, iters=20):
    """The gradient bucket's all-reduce through a ONE-rank RCCL group on this GPU: the collective's launch + kernel latency, the only
    term of DESIGN section 6's estimate a 1-GPU box can measure (no link traffic: with one rank RCCL copies in place)."""
    import torch.distributed as dist
    own = not dist.is_initialized()
    if own:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with _quiet_stdout():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        buf = torch.zeros(nfloats, dtype=torch.float32, device=dev)
        with _quiet_stdout():
            for _ in range(3):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    finally:
        if own:
            dist.destroy_process_group()


def measure_traffic(batch, n, timeout_s=240):
    """HBM-side bytes per launch of the two dominant kernels, measured NOW: two child runs of tools/pmc_workload.py --dominant under
    `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (separate passes, as MI355X_MICROARCH.md
    prescribes; FETCH_SIZE x2 on gfx950, the factor checked on a 256 MiB copy inside the same child).  -> {"forward": bytes,
    "train": bytes, "calibration": str} or None when rocprofv3 is absent / a pass fails (the caller then falls back to the committed
    profiles and says so)."""
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_report
    tmp = tempfile.mkdtemp(prefix="rtk_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    dbs = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rp, "--pmc", counter, "--kernel-trace", "-d", out, "-o", "t", "--", sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py"),
                   "--dominant", "--batch", str(batch), "--npoints", str(n)]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            found = [os.path.join(r, f) for r, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
            if not found:
                return None
            dbs[counter] = pmc_report.table(found[0], True)
        F, W = dbs["FETCH_SIZE"], dbs["WRITE_SIZE"]
        dm = pmc_report.demangle(sorted(set(F) | set(W)))
        cal = [k for k in W if "copy" in dm[k].lower()]
        scale, note = 2.0, "FETCH_SIZE x 2 (guide); no calibration copy seen"
        if cal:
            k = max(cal, key=lambda k: W[k][4])
            f_kib = F.get(k, (0, 0, 0, 0, 0))[4]
            if f_kib:
                scale = 262144.0 / f_kib
                note = "256 MiB copy in the same child: FETCH_SIZE %.0f KiB -> x%.3f, WRITE_SIZE %.0f KiB" % (f_kib, scale, W[k][4])

        def per_launch(pat):
            ks = [k for k in set(F) | set(W) if pat in dm[k]]
            if not ks:
                return None
            k = max(ks, key=lambda k: F.get(k, (0, 0, 0.0))[2])
            return int((F.get(k, (0, 0, 0.0))[2] * scale + W.get(k, (0, 0, 0.0))[2]) * 1024)
        return {"forward": per_launch("cost_volume_split_kernel<false>") or per_launch("cost_volume_kernel<"),
                "train": per_launch("cost_volume_bwd_split_kernel") or per_launch("cost_volume_bwd_kernel"), "calibration": note}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (SURVEY 8(d) protocol)
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_runs(fn, warmups, min_runs, max_runs, budget_s):
    for _ in range(warmups):
        fn()
    ts, t_all = [], time.perf_counter()
    while len(ts) < max_runs and (len(ts) < min_runs or time.perf_counter() - t_all < budget_s):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return ts


def cpu_baseline(n, budget_s=20.0):
    """The CPU oracle's backbone forward on this box's host cores, a BOUNDED sample (about 20 s of CPU work, so that the GPU is what
    the run spends its time on): B=1 with 1 thread (2 warm-ups, median of 10 runs -- SURVEY 8(d)'s protocol, and the fastest setting on
    every box measured so far), B=1 with all threads (1 warm-up, median of 5) and B=32 with min(32, cpus) threads (1 warm-up, median of
    3; 8/16/32 threads measured within 10 % of each other in round 3, all-thread and 1-thread settings 3-10x slower)."""
    from oracle import track4d_ref as R
    from ratrack_amd import synth
    from ratrack_amd.track4d import Args, Track4D
    net = Track4D(Args())
    synth.fill_state_dict(net.state_dict())
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    ncpu = os.cpu_count() or 1
    all_threads = torch.get_num_threads()
    data = {}
    for b in (1, 32):
        d = synth.make_frame_pairs(b, n, case_id=99)
        data[b] = {k: torch.from_numpy(v) for k, v in d.items() if k != "gt_cls"}
    rows, t_start = [], time.perf_counter()

    def measure(b, threads, warmups, runs):
        if time.perf_counter() - t_start > budget_s:
            return
        torch.set_num_threads(threads)
        t = data[b]
        with torch.no_grad():
            ts = _cpu_runs(lambda: R.backbone(sd, t["pc1"], t["pc2"], t["feature1"], t["feature2"], None), warmups, runs, runs, 0.0)
        med = statistics.median(ts)
        rows.append({"batch": b, "threads": threads, "warmups": warmups, "runs": len(ts), "median_s": round(med, 4),
                     "pairs_per_s": round(b / med, 2)})

    try:
        measure(1, 1, 2, 10)
        measure(1, all_threads, 1, 5)
        measure(32, min(32, ncpu), 1, 3)
    finally:
        torch.set_num_threads(all_threads)
    best = max(rows, key=lambda r: r["pairs_per_s"])
    try:      # ... and the whole host: P single-thread processes at once (the throughput the box's cores give, not one core's latency)
        throughput = cpu_throughput_baseline(n, min(64, ncpu))
    except Exception as e:
        throughput = {"error": repr(e)[:200]}
    return {"value": best["pairs_per_s"], "unit": "frame-pairs/s", "cores": best["threads"], "kind": "port", "host_cpus": ncpu,
            "throughput": throughput,
            "sample": "CPU oracle backbone forward, N=%d: best of (B=1, 1 thread), (B=1, %d threads), (B=32, %d threads) = B=%d with %d "
                      "threads (median of %d runs); %.0f s of CPU work in total"
                      % (n, all_threads, min(32, ncpu), best["batch"], best["threads"], best["runs"], time.perf_counter() - t_start),
            "runs": rows}


def _cpu_throughput_worker(n, rendezvous, seconds):
    """`python bench.py --cpu-worker N DIR SECONDS`: one single-thread B=1 oracle forward loop; announces itself in DIR (ready.<pid>), starts
    when DIR/go appears, prints "<forwards> <seconds>"."""
    torch.set_num_threads(1)
    from oracle import track4d_ref as R
    from ratrack_amd import synth
    from ratrack_amd.track4d import Args, Track4D
    net = Track4D(Args())
    synth.fill_state_dict(net.state_dict())
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    d = synth.make_frame_pairs(1, n, case_id=99)
    t = {k: torch.from_numpy(v) for k, v in d.items() if k != "gt_cls"}
    with torch.no_grad():
        R.backbone(sd, t["pc1"], t["pc2"], t["feature1"], t["feature2"], None)      # warm-up
        open(os.path.join(rendezvous, "ready.%d" % os.getpid()), "w").close()
        t_wait = time.time()
        while not os.path.exists(os.path.join(rendezvous, "go")) and time.time() - t_wait < 300:
            time.sleep(0.01)
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds:
            R.backbone(sd, t["pc1"], t["pc2"], t["feature1"], t["feature2"], None)
            k += 1
        print(k, time.perf_counter() - t0, flush=True)


def cpu_throughput_baseline(n, procs, seconds=8.0, startup_s=120.0):
    """What the host can do when every core works: `procs` independent single-thread processes, each looping the CPU oracle's B=1 forward
    (the fastest setting per core) over the same `seconds` window -- started together, once all of them have loaded and warmed up (or
    `startup_s` have passed).  -> pairs/s summed over the processes."""
    import shutil
    import subprocess
    import tempfile
    rdv = tempfile.mkdtemp(prefix="rtk_cpu_", dir="/tmp")
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(n), rdv, repr(seconds)], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(procs)]
        t0 = time.time()
        while time.time() - t0 < startup_s and sum(f.startswith("ready.") for f in os.listdir(rdv)) < procs:
            time.sleep(0.05)
        open(os.path.join(rdv, "go"), "w").close()
        done, late = [], 0
        for p_ in ps:
            try:
                out, _ = p_.communicate(timeout=startup_s + seconds * 3 + 60)
                k, el = out.split()
                done.append((int(k), float(el)))
            except Exception:
                p_.kill()
                late += 1
    finally:
        shutil.rmtree(rdv, ignore_errors=True)
    if not done:
        raise RuntimeError("no CPU worker finished")
    rate = sum(k / el for k, el in done)
    return {"value": round(rate, 1), "unit": "frame-pairs/s", "cores": len(done), "processes_failed": late,
            "sample": "%d independent single-thread processes, each the CPU oracle's B=1 forward in a loop for %.0f s (same window), "
                      "N=%d: %d forwards in all" % (len(done), seconds, n, sum(k for k, _ in done))}


def cpu_train_baseline(batch, n, budget_s=15.0):
    """The CPU oracle's train step (train-mode forward + multi-task loss + backward, torch-CPU autograd over the C ops)."""
    from oracle import track4d_ref as R
    from ratrack_amd import loss as L
    from ratrack_amd import synth
    from ratrack_amd.track4d import Args, Track4D
    net = Track4D(Args())
    synth.fill_state_dict(net.state_dict())
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in net.state_dict().items()}
    d = synth.make_frame_pairs(batch, n, case_id=98)
    t = {k: torch.from_numpy(v) for k, v in d.items()}

    def step():
        flow, h, cls, *_ = R.backbone(sd, t["pc1"], t["pc2"], t["feature1"], t["feature2"], None, training=True)
        total, _ = L.backbone_loss(t["pc1"] + flow, cls, t["gt_warp"], t["gt_cls"])      # a few reductions, device-agnostic
        total.backward()
    ts = _cpu_runs(step, 1, 3, 10, budget_s)
    med = statistics.median(ts)
    return {"value": round(batch / med, 3), "unit": "frame-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "median of %d train steps (forward + loss + backward) of the CPU oracle at B=%d, N=%d after 1 warm-up (%.1f s)"
                      % (len(ts), batch, n, sum(ts))}


# ---------------------------------------------------------------------------------------------------------------------
# train step (BASELINE config 3 / 4)
# ---------------------------------------------------------------------------------------------------------------------
def time_cost_volume_bwd(batch, n, dev, iters=10):
    """Dominant kernel of the train step (cost_volume_bwd_kernel) timed live with HIP events at the bench shape.
    Returns (ms per launch, FLOPs per launch: forward recompute + both 256x256 input gradients + the in-kernel weight-gradient
    contractions, counted from the kernel's own arithmetic)."""
    from ratrack_amd import train_ops
    return train_ops.time_cost_volume_bwd(batch, n, dev, iters)


def run_train(a, net, d, dev, dist, world, rank, steps, warmup, live_traffic=None):
    """Train step per rank on B frame-pairs: train-mode forward (training path: fused HIP operators under autograd),
    multi-task loss, backward, ONE flat gradient all-reduce over RCCL, Adam.  Weak scaling (B per GPU fixed).
    world 1: one hipGraph; world > 1: graph (forward..gradient pack) -> eager RCCL all-reduce -> graph (Adam)."""
    from ratrack_amd.ddp import broadcast_parameters
    from ratrack_amd.train import Trainer
    broadcast_parameters(net)
    use_graph = not a.no_graph
    tr = Trainer(net, graph=use_graph, graph_collective=a.graph_collective, deterministic=a.deterministic)
    ds = d if isinstance(d, (list, tuple)) else [d]      # distinct resident batches, rotated through the steps
    ts = [{k: torch.from_numpy(v).to(dev) for k, v in x.items()} for x in ds]
    h = torch.zeros(5, a.batch, 128, device=dev)
    rot = {"i": 0}

    def step():
        t = ts[rot["i"] % len(ts)]
        rot["i"] += 1
        return tr.step(t["pc1"], t["pc2"], t["feature1"], t["feature2"], t["gt_warp"], t["gt_cls"], h)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(warmup, 5 if use_graph else 1)):      # >= 3 eager warm-ups + the capture + one replay
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    mine = time.perf_counter() - t0
    el, per_rank = reduce_times(mine, dist, dev, world)
    # the gradient all-reduce alone (the bucket as the step leaves it), back to back between two HIP events
    ar_us = None
    if world > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            tr.reducer.all_reduce()
        barrier()
        e0.record()
        for _ in range(20):
            tr.reducer.all_reduce()
        e1.record()
        torch.cuda.synchronize()
        ar_us = e0.elapsed_time(e1) / 20 * 1e3
    ar1_us = None
    if world == 1:      # a 1-rank RCCL group: the collective's launch + kernel latency on this box
        try:
            ar1_us = allreduce_1rank_us(sum(p.numel() for p in net.parameters() if p.requires_grad) + 5, dev)
        except Exception:
            ar1_us = None
    res = None
    if rank == 0:
        ms_step = el / steps * 1e3
        kms, kflops = time_cost_volume_bwd(a.batch, a.npoints, dev)
        ach = kflops / (kms * 1e-3) / 1e12
        pm = _pmc("train", a.batch, a.npoints)
        if live_traffic:
            pm = {"traffic_bytes_per_launch": live_traffic, "source": "measured in this run (rocprofv3 --pmc child passes)"}
        kernels = None
        try:      # device kernels of ONE eager step (what a replay of the captured graph executes), counted by the profiler
            from torch.profiler import ProfilerActivity, profile
            tr.graph, g_saved = False, tr.graph
            step()                                    # one unprofiled eager step first (allocator, optimizer table), then the
            torch.cuda.synchronize()                  # maximum over two profiled ones: the tracer occasionally drops part of a window
            counts = []
            for _ in range(2):
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    step()
                    torch.cuda.synchronize()
                counts.append(sum(1 for e in prof.events() if str(e.device_type).endswith("CUDA") and "Memcpy" not in e.name
                                  and "Memset" not in e.name))
            tr.graph = g_saved
            kernels = max(counts)
        except Exception:
            pass
        res = {"ms_per_step": round(ms_step, 3), "pairs_per_s": round(a.batch * world * steps / el, 1), "steps": steps,
               "hipGraph": ("one graph" if world == 1 or not tr.split else "graph | RCCL all-reduce | graph") if use_graph else False,
               "deterministic": bool(a.deterministic),
               "kernels_per_step": kernels,
               "workload": "Track4D.backbone train step (fwd + multi-task loss + bwd + grad all-reduce + Adam), B=%d x N=%d per GPU, %d distinct "
                           "resident batches in rotation" % (a.batch, a.npoints, len(ts)),
               "allreduce_bytes": tr.reducer.bucket_bytes or 4 * sum(p.numel() for p in net.parameters() if p.requires_grad),
               "allreduce_gradient_bytes": tr.reducer.payload_bytes or 4 * sum(p.numel() for p in net.parameters() if p.grad is not None),
               "allreduce_us": None if ar_us is None else round(ar_us, 1),
               "allreduce_us_1rank": None if ar1_us is None else round(ar1_us, 1),
               "per_rank_ms_per_step": {"min": round(min(per_rank) / steps * 1e3, 3), "max": round(max(per_rank) / steps * 1e3, 3)},
               "roofline": train_roofline(a, kms, kflops, ach, pm, ms_step),
               "whole_step": {**_train_total_traffic(a),"hbm_frac_algorithmic_3x": round(3 * ALG_BYTES_PER_PAIR.get(a.npoints, 0) * a.batch / (ms_step * 1e-3)
                                                               / (HBM_PEAK_GBS * 1e9), 5),
                              "fp32_frac_algorithmic_3x": round(3 * ALG_FLOPS_PER_PAIR.get(a.npoints, 0) * a.batch / (ms_step * 1e-3)
                                                                / (FP32_PEAK_TFLOPS * 1e12), 5)}}
    return res


def reduce_times(mine, dist, dev, world):
    """-> (max over ranks, list of every rank's time): the headline uses the maximum, the spread is reported."""
    if dist is None:
        return mine, [mine]
    t = torch.tensor([mine], device=dev, dtype=torch.float64)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t)
    per_rank = [float(x.item()) for x in allt]
    return max(per_rank), per_rank


def dry_run(a, world, rank):
    """`--dry-run`: the launcher / rendezvous / barrier / max-over-ranks / one-JSON-line logic of this file on CPU (gloo) with a stand-in
    step (rank r sleeps (r + 1) ms) -- what tests/test_bench_cpu.py runs with --gpus 2, since the real thing needs GPUs."""
    dist = None
    dev = torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    step = lambda: time.sleep(1e-3 * (rank + 1))

    def barrier():
        if dist is not None:
            dist.barrier()
    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    el, per_rank = reduce_times(time.perf_counter() - t0, dist, dev, world)
    if rank == 0:
        print(json.dumps({"metric": "dry run (CPU stand-in step)", "value": round(a.batch * world * a.steps / el, 1), "unit": "frame-pairs/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(el / a.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
                          "config": {"workload": "dry run", "global_batch": a.batch * world},
                          "per_rank_ms_per_step": {"min": round(min(per_rank) / a.steps * 1e3, 3), "max": round(max(per_rank) / a.steps * 1e3, 3)}}),
              flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def forward_config(net, batch, n, depth, dev, seconds=CONFIG_SECONDS, case0=3000):
    """One of BASELINE.json's other single-GPU forward configurations with the headline's method (captured graphs, `depth` batches in
    flight, NBATCH distinct resident batches in rotation, >= `seconds` timed after 0.3 s of untimed steps), plus its dominant kernel
    alone between HIP events."""
    from ratrack_amd import fused, synth
    hosts = [synth.make_frame_pairs(batch, n, case_id=case0 + i) for i in range(NBATCH)]
    bs = [tuple(torch.from_numpy(x[k]).to(dev) for k in ("pc1", "pc2", "feature1", "feature2")) for x in hosts]
    h = torch.zeros(5, batch, 128, device=dev)
    with torch.no_grad():
        net.backbone(*bs[0], h)
        eng = net._fused_engine()
        pipe = fused.GraphPipeline(eng, (*bs[0], h), depth=depth)
        i = 0

        def run(steps):
            nonlocal i
            for _ in range(steps):
                i += 1
                pipe.submit(*bs[i % NBATCH], h)
            pipe.drain()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(8)
        per = (time.perf_counter() - t0) / 8
        run(max(8, int(0.3 / per)))
        steps = max(20, int(seconds / per))
        t0 = time.perf_counter()
        run(steps)
        el = time.perf_counter() - t0
        pipe = None
        net.backbone(*bs[0], h)      # (eager: the engine's last cost-volume operands)
        alone = eng.time_dominant_kernel(10)
        alone_ms = sum(s.elapsed_time(e) for s, e in alone) / len(alone)
    cv_flops = cost_volume_flops_per_pair(n) * batch
    pps = batch * steps / el
    out = {"workload": "Track4D.backbone forward, B=%d x N=%d, %d batches in flight, %d resident batches in rotation" % (batch, n, depth, NBATCH),
           "steps": steps, "ms_per_step": round(el / steps * 1e3, 4), "pairs_per_s": round(pps, 1),
           "dominant_kernel": {"kernel": "cost_volume_split_kernel", "alone_ms": round(alone_ms, 4), "flops_per_launch": cv_flops,
                               "frac_of_split_peak_alone": round(cv_flops / (alone_ms * 1e-3) / 1e12 / SPLIT_PEAK_TFLOPS, 4)}}
    if n in ALG_BYTES_PER_PAIR:
        out["hbm_frac_algorithmic"] = round(pps * ALG_BYTES_PER_PAIR[n] / (HBM_PEAK_GBS * 1e9), 5)
    return out


# ---------------------------------------------------------------------------------------------------------------------
def _self_spawn(a):
    """`python bench.py --gpus N` with no launcher: become `python -m torch.distributed.run ... bench.py <same args>`."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def main():
    if len(sys.argv) == 5 and sys.argv[1] == "--cpu-worker":
        return _cpu_throughput_worker(int(sys.argv[2]), sys.argv[3], float(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="frame-pairs per GPU per step")
    ap.add_argument("--npoints", type=int, default=256, help="radar points per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the train-step leg of the default (forward) run")
    ap.add_argument("--no-irregular", action="store_true", help="skip the irregular-op roofline leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the forward legs of BASELINE's other single-GPU configs (B=32 at N=256 / 1024)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--pipeline", type=int, default=4, help="captured graphs in flight (batch-level pipelining on streams)")
    ap.add_argument("--train-steps", type=int, default=20)
    ap.add_argument("--seconds", type=float, default=HEADLINE_SECONDS, help="length of the headline timed region (never fewer than --steps steps)")
    ap.add_argument("--evidence", default=None, help="rank 0 writes the in-situ event timings of the dominant kernel behind roofline.kernel_ms to "
                    "profiles/<NAME>_insitu_cost_volume.txt")
    ap.add_argument("--graph-collective", action="store_true", help="world > 1: capture the RCCL all-reduce inside the train graph (one graph) "
                                                                     "instead of graph | eager all-reduce | graph")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo run of the distributed control flow with a stand-in step (tests)")
    ap.add_argument("--traffic", choices=["auto", "profiles", "off"], default="auto",
                    help="roofline.traffic: auto = measure now with two rocprofv3 --pmc child passes when rocprofv3 is present, else take the "
                         "committed profiles/ value (and say so); profiles = always the committed value")
    ap.add_argument("--deterministic", action="store_true", help="train legs: Trainer(deterministic=True) -- order-independent sums, bit-reproducible steps")
    ap.add_argument("--mode", choices=["forward", "train"], default="forward",
                    help="forward = the headline metric (eval backbone, fused kernels) with the train step embedded as `train`; "
                         "train = the train step (BASELINE config 3/4) as the headline line")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_spawn(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.dry_run:
        assert a.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (a.gpus, world)
        return dry_run(a, world, rank)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _quiet_stdout():
            dist.init_process_group("nccl", device_id=dev)      # RCCL
            dist.barrier()                                      # (the communicator, and its banner, come up with the first collective)
    assert a.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (a.gpus, world)

    from ratrack_amd import synth
    from ratrack_amd.track4d import Args, Track4D

    net = Track4D(Args()).to(dev).eval()
    synth.fill_state_dict(net.state_dict())
    net.invalidate_fused()
    # every rank its own batches; NBATCH distinct ones resident in HBM, rotated through the steps so that no step re-reads its
    # predecessor's inputs (round-3 review: one re-submitted batch stays L2/MALL-hot)
    hosts = [synth.make_frame_pairs(a.batch, a.npoints, case_id=1000 + rank + 100 * i) for i in range(NBATCH)]
    d = hosts[0]
    batches = [tuple(torch.from_numpy(x[k]).to(dev) for k in ("pc1", "pc2", "feature1", "feature2")) for x in hosts]
    pc1, pc2, f1, f2 = batches[0]
    h = torch.zeros(5, a.batch, 128, device=dev)
    rot = {"i": 0}

    def next_batch():
        rot["i"] += 1
        return batches[rot["i"] % NBATCH]

    def finish(res):
        if rank == 0:
            print(json.dumps(res), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    if a.mode == "train":
        tr = run_train(a, net, hosts[:4], dev, dist, world, rank, a.steps, a.warmup)
        res = None
        if rank == 0:
            res = {"metric": "radar frame-pairs/sec (train step) at B=%d,N=%d per GPU" % (a.batch, a.npoints), "value": tr["pairs_per_s"],
                   "unit": "frame-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": tr["ms_per_step"],
                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
                   "config": {"workload": tr["workload"] + ", hipGraph=%s" % tr["hipGraph"], "global_batch": a.batch * world,
                              "parallelism": "dp%d, one flat RCCL all-reduce of %d bytes per step" % (world, tr["allreduce_bytes"])},
                   "roofline": tr["roofline"], "whole_step": tr["whole_step"],
                   "train": {"hipGraph": tr["hipGraph"], "deterministic": tr["deterministic"], "kernels_per_step": tr["kernels_per_step"],
                             "allreduce_bytes": tr["allreduce_bytes"],
                             "allreduce_gradient_bytes": tr["allreduce_gradient_bytes"], "allreduce_us": tr["allreduce_us"],
                             "allreduce_us_1rank": tr["allreduce_us_1rank"]},
                   "per_rank_ms_per_step": tr["per_rank_ms_per_step"]}
            if world == 1 and not a.no_cpu_baseline:
                try:
                    res["cpu_baseline"] = cpu_train_baseline(min(a.batch, 4), a.npoints)
                except Exception as e:          # the baseline is a reported extra, never a reason to lose the measurement
                    res["cpu_baseline"] = {"error": repr(e)[:200]}
        return finish(res)

    # ---- forward (headline) -------------------------------------------------------------------------------------------
    from ratrack_amd import fused
    with torch.no_grad():
        net.backbone(pc1, pc2, f1, f2, h)
        eng = net._fused
        assert eng, "fused engine not active"
        # multiply-adds this design executes per step, from the launch shapes of one eager pass (data dependent through the
        # exhausted-cloud counters, hence measured, not hard-coded)
        with fused.trace_work() as tw:
            net.backbone(pc1, pc2, f1, f2, h)
        exec_macs, exec_by_kernel = tw.executed_macs()
        # every launch of one eager pass between two HIP events, one stream, nothing else on the GPU: per kernel family the time
        # and -- for the FLOP-carrying ones -- the executed rate against the split-path matrix peak
        from ratrack_amd import _lib as rtk_lib
        eng.use_side_stream, side_saved = False, eng.use_side_stream
        net.backbone(pc1, pc2, f1, f2, h)
        torch.cuda.synchronize()
        fam_of = {"rtk_pointwise_mlp": "pointwise", "rtk_sa_scale": "sa_scale", "rtk_sa_scale_split": "sa_scale", "rtk_cost_volume_split": "cost_volume",
                  "rtk_cost_volume": "cost_volume", "rtk_patch_cost": "patch_cost", "rtk_gru_step": "gru"}
        ALONE_PASSES = 7
        passes = []
        for _ in range(ALONE_PASSES):      # the MEDIAN over the passes, family by family (round 5 committed one pass: a 155 us outlier in it)
            rtk_lib.TIMING = timing = []
            net.backbone(pc1, pc2, f1, f2, h)
            rtk_lib.TIMING = None
            torch.cuda.synchronize()
            fam = {}
            for nm, e0, e1 in timing:
                f = fam.setdefault(fam_of.get(nm, nm.replace("rtk_", "")), [0, 0.0])
                f[0] += 1
                f[1] += e0.elapsed_time(e1) * 1e3
            passes.append(fam)
        eng.use_side_stream = side_saved
        by_family = {k: [passes[0][k][0], statistics.median(p_[k][1] for p_ in passes if k in p_)] for k in passes[0]}

        spread = {}

        def timed(step, drain, steps, warmup, warm_seconds=0.0, tag=None):
            """`warmup` untimed steps (and, with warm_seconds, as many more as that takes: clocks and caches of a fresh box settle
            in ~0.5 s, not in the 5-10 steps = 5-10 ms a caller asks for), then EXACTLY `steps` steps between barrier + synchronize."""
            def barrier():
                drain()                                            # every submitted batch finishes inside the timed region
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
            t_w = time.perf_counter()
            n_w = 0
            while n_w < warmup or time.perf_counter() - t_w < warm_seconds:
                step()
                n_w += 1
                if n_w % 64 == 0:
                    drain()
                    torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            barrier()
            el, per_rank = reduce_times(time.perf_counter() - t0, dist, dev, world)
            if tag:
                spread[tag] = {"min": round(min(per_rank) / steps * 1e3, 4), "max": round(max(per_rank) / steps * 1e3, 4), "untimed_warmup_steps": n_w}
            return el

        depth = max(1, a.pipeline)
        if a.no_graph:
            step_fn, drain_fn = (lambda: net.backbone(*next_batch(), h)), (lambda: None)
            kstep = timed(step_fn, drain_fn, a.steps, a.warmup, 0.5, "k_steps")
            # the headline window: the same region over >= HEADLINE_SECONDS (never fewer than the K steps asked for) -- with the driver's
            # --steps 20 the K-step window is 15 ms, which no clock outside this process resolves
            timed_steps = max(a.steps, int(a.seconds / max(kstep / a.steps, 1e-6)))
            elapsed = timed(step_fn, drain_fn, timed_steps, 0, 0.0, "headline")
            eng.kernel_events = events = []
            insitu = timed(step_fn, drain_fn, a.steps, 2)
            eng.kernel_events = None
        else:
            pipe = fused.GraphPipeline(eng, (pc1, pc2, f1, f2, h), depth=depth)
            step_fn = lambda: pipe.submit(*next_batch(), h)      # inputs are copied into the slot's static buffers
            kstep = timed(step_fn, pipe.drain, a.steps, a.warmup, 0.5, "k_steps")
            timed_steps = max(a.steps, int(a.seconds / max(kstep / a.steps, 1e-6)))
            elapsed = timed(step_fn, pipe.drain, timed_steps, 0, 0.0, "headline")
            # second timed region, same workload and concurrency, graphs split around the dominant kernel: its duration in situ
            pipe2 = fused.GraphPipeline(eng, (pc1, pc2, f1, f2, h), depth=depth, split_cost_volume=True)
            events = []
            pipe2.set_kernel_events(events)
            insitu = timed(lambda: pipe2.submit(*next_batch(), h), pipe2.drain, max(a.steps, 50), 3)
            pipe2.set_kernel_events(None)
        torch.cuda.synchronize()
        events = events[-max(a.steps, 50):] if not a.no_graph else events[-a.steps:]
        insitu_steps = len(events)
        kern_each = [s.elapsed_time(e) for s, e in events]
        kern_ms = sum(kern_each) / max(len(kern_each), 1)
        # the same kernel on the same operands with nothing else on the GPU (what a serialising profiler such as
        # rocprofv3 --kernel-trace reports for it)
        alone = eng.time_dominant_kernel(20)
        alone_ms = sum(s.elapsed_time(e) for s, e in alone) / len(alone)
        # ... and as the pipeline launches it from depth 3: on a share of the CUs (fused.cv_shared_workgroups)
        shared_wgs = fused.cv_shared_workgroups(a.batch, a.npoints, dev) if (depth > 2 and not a.no_graph) else 0
        alone_shared_ms = None
        if shared_wgs:
            eng.cv_shared = True
            ev = eng.time_dominant_kernel(20)
            eng.cv_shared = False
            alone_shared_ms = sum(s.elapsed_time(e) for s, e in ev) / len(ev)

    res = None
    if rank == 0 and a.evidence and kern_each:
        with open(os.path.join(ROOT, "profiles", a.evidence + "_insitu_cost_volume.txt"), "w") as f:
            f.write("# bench.py --evidence %s: cost_volume_split_kernel IN SITU, B=%d N=%d, %d batches in flight -- one line per launch,\n"
                    "# HIP events around the launch on its own launch stream inside a timed region of the pipelined workload (graphs split around\n"
                    "# the kernel; the measured kernels of the batches in flight chained by events).  roofline.kernel_ms = their mean.\n"
                    % (a.evidence, a.batch, a.npoints, depth))
            for k, v in enumerate(kern_each):
                f.write("%3d  %.4f ms\n" % (k, v))
            f.write("# mean %.4f ms  median %.4f  min %.4f  max %.4f  (n = %d);  alone (nothing else in flight): %.4f ms\n"
                    % (kern_ms, statistics.median(kern_each), min(kern_each), max(kern_each), len(kern_each), alone_ms))
    if rank == 0:
        pairs_per_s = a.batch * world * timed_steps / elapsed
        cv_flops = cost_volume_flops_per_pair(a.npoints) * a.batch
        achieved = cv_flops / (kern_ms * 1e-3) / 1e12
        pm = _pmc("forward", a.batch, a.npoints)
        traffic = {"bytes": pm["traffic_bytes_per_launch"] if pm else None, "source": (pm or {}).get("source")}
        live = None
        if a.traffic == "auto" and world == 1:
            live = measure_traffic(a.batch, a.npoints)
            if live and live.get("forward"):
                traffic = {"bytes": live["forward"], "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes (%s)" % live["calibration"]}
        per_gpu = pairs_per_s / world
        split = bool(getattr(eng, "cv_split", False))
        prof_us, prof_file = profile_kernel_avg_us("cost_volume_split_kernel<false>" if split else "cost_volume_kernel<") if (a.batch, a.npoints) == (64, 256) else (None, None)
        cv_peak = SPLIT_PEAK_TFLOPS if split else FP32_PEAK_TFLOPS
        exec_flops_per_pair = 2.0 * exec_macs / a.batch
        res = {
            "metric": "radar frame-pairs/sec (backbone forward, eval) at B=%d,N=%d per GPU" % (a.batch, a.npoints),
            "value": round(pairs_per_s, 1),
            "unit": "frame-pairs/s",
            "n_gpus": world, "steps": timed_steps, "steps_requested": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / timed_steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic",
            "per_rank_ms_per_step": spread.get("headline"),
            "timed_region": "EXACTLY `steps` steps between barrier + synchronize, max over ranks; steps = max(K asked for, what %g s take): " % a.seconds +
                            "the K-step window alone is `k_step_window`; %d distinct resident batches rotate through the steps" % NBATCH,
            "k_step_window": {"steps": a.steps, "ms_per_step": round(kstep / a.steps * 1e3, 4),
                              "value": round(a.batch * world * a.steps / kstep, 1), "per_rank_ms_per_step": spread.get("k_steps"),
                              "what": "the first K steps after the warm-up (>= 0.5 s of untimed steps)"},
            "config": {"workload": "Track4D.backbone forward, B=%d frame-pairs x N=%d points per GPU, S=512 centroids, "
                                   "eval-mode BN, random-init weights, hipGraph=%s, batches in flight=%d"
                                   % (a.batch, a.npoints, not a.no_graph, 1 if a.no_graph else depth),
                       "global_batch": a.batch * world, "parallelism": "replicas x%d (no collective on the forward path)" % world},
            "roofline": {"kernel": "cost_volume_split_kernel" if split else "cost_volume_kernel", "bound": "mfma",
                         "achieved": round(achieved, 2), "peak": round(cv_peak, 1),
                         "unit": "TFLOP/s", "frac": round(achieved / cv_peak, 4),
                         "peak_is": ("fp32-equivalent: dense fp16 MFMA peak (2500) / 3 -- the kernel takes every fp32 product as three fp16 "
                                     "MFMA products of two operand pieces each (csrc/split_mfma.h); `achieved` counts the algorithmic fp32 "
                                     "flops, so `frac` is also the executed fp16 rate over 2500.  Against the fp32-input MFMA peak "
                                     "(157.3, the roofline of the round-1/2 kernel) the same launch is at %.2f in situ / %.2f alone"
                                     % (achieved / FP32_PEAK_TFLOPS, cv_flops / (alone_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS))
                                    if split else "fp32-input MFMA peak",
                         "traffic": traffic["bytes"], "traffic_source": traffic["source"],
                         "kernel_ms": round(kern_ms, 4), "flops_per_launch": cv_flops,
                         "profile_frac": (round(cv_flops / (prof_us * 1e-6) / 1e12 / cv_peak, 4) if prof_us else None),
                         "profile_avg_us": prof_us, "profile_source": ("profiles/%s (rocprofv3 --kernel-trace --stats of this bench: dispatches "
                                                                       "serialised by the profiler)" % prof_file) if prof_file else None,
                         "alone": {"kernel_ms": round(alone_ms, 4), "frac": round(cv_flops / (alone_ms * 1e-3) / 1e12 / cv_peak, 4),
                                   "what": "the same launch with nothing else in flight (back-to-back launches between HIP events; what "
                                           "rocprofv3 --kernel-trace, which serialises dispatches, shows): in situ the kernel shares the "
                                           "CUs with the kernels of the other batches in flight, which is what the pipelining is for"},
                         "alone_as_launched": ({"workgroups": shared_wgs, "kernel_ms": round(alone_shared_ms, 4),
                                                "what": "the pipeline's launch (rtk_cost_volume_split_shared: the kernel keeps its CUs whole, so "
                                                        "with several batches in flight it is given 3/4 of them) with nothing else in flight"}
                                               if alone_shared_ms else None),
                         "measured": "in situ: %d launches between HIP events inside a timed region of the same pipelined workload "
                                     "(graphs split around the kernel, the measured kernels of the batches in flight chained by events "
                                     "so that they do not time-share the CUs; %.4f ms/step there)" % (len(events), insitu / max(insitu_steps, 1) * 1e3),
                         "kernel_ms_each": {"min": round(min(kern_each), 4), "median": round(statistics.median(kern_each), 4),
                                            "max": round(max(kern_each), 4), "n": len(kern_each)} if kern_each else None},
            # whole path per GPU against both rooflines (SURVEY.md H1 asks for both).  "algorithmic" = the reference
            # formulation's 14.15 MB / 4.003 GFLOP per pair; "executed" = the multiply-adds this design issues (per-point
            # layer-1 projections, duplicate centroids skipped), counted from this run's launch shapes.
            "whole_path": {"hbm_frac_algorithmic": round(per_gpu * ALG_BYTES_PER_PAIR.get(a.npoints, 0) / (HBM_PEAK_GBS * 1e9), 5),
                           "fp32_frac_algorithmic": round(per_gpu * ALG_FLOPS_PER_PAIR.get(a.npoints, 0) / (FP32_PEAK_TFLOPS * 1e12), 5),
                           "fp32_frac_executed": round(per_gpu * exec_flops_per_pair / (FP32_PEAK_TFLOPS * 1e12), 5),
                           # against the matrix roofline of this design's own arithmetic (fp16 peak / 3 products per fp32 product):
                           # the ceiling of the path as formulated, and what the north star's "30 % of HBM" would need of it
                           "split_frac_executed": round(per_gpu * exec_flops_per_pair / (SPLIT_PEAK_TFLOPS * 1e12), 5),
                           "split_ceiling_pairs_per_s": round(SPLIT_PEAK_TFLOPS * 1e12 / exec_flops_per_pair, 0),
                           "hbm_30pct_pairs_per_s": round(0.3 * HBM_PEAK_GBS * 1e9 / ALG_BYTES_PER_PAIR.get(a.npoints, 1), 0),
                           "executed_gflop_per_pair": round(exec_flops_per_pair / 1e9, 4),
                           "executed_gflop_per_pair_by_kernel": {k: round(2.0 * v / a.batch / 1e9, 4) for k, v in exec_by_kernel.items()}},
            # one eager pass, every launch between HIP events on one stream with nothing else in flight (what a serialising profiler
            # shows); FLOP-carrying families: executed multiply-adds x 2 / time against the split matrix peak (fp16 / 3)
            "kernels_alone_method": "median over %d eager passes, every launch between HIP events on one stream with nothing else in flight" % ALONE_PASSES,
            "kernels_alone": {k: dict({"launches": v[0], "us": round(v[1], 1)},
                                      **({"gflop": round(2.0 * exec_by_kernel[k] / 1e9, 2),
                                          "tflops": round(2.0 * exec_by_kernel[k] / (v[1] * 1e-6) / 1e12, 1),
                                          "frac_of_split_peak": round(2.0 * exec_by_kernel[k] / (v[1] * 1e-6) / 1e12 / SPLIT_PEAK_TFLOPS, 3)}
                                         if k in exec_by_kernel and v[1] > 0 else {}))
                              for k, v in sorted(by_family.items(), key=lambda kv: -kv[1][1])},
        }
    # ---- BASELINE.json's other single-GPU forward configurations (config 2: B=32, N=256; config 5: B=32, N=1024) -----------
    if rank == 0 and world == 1 and not a.no_configs and (a.batch, a.npoints) == (64, 256):
        res["configs"] = {}
        for name, (b_, n_) in (("config2_B32_N256", (32, 256)), ("config5_B32_N1024", (32, 1024))):
            try:
                pipe = pipe2 = None
                res["configs"][name] = forward_config(net, b_, n_, depth, dev)
            except Exception as e:
                res["configs"][name] = {"error": repr(e)[:300]}
    # ---- train step (config 3; config 4 when world > 1) ---------------------------------------------------------------
    if not a.no_train:
        try:
            # the forward leg's pipelines (eight captured graphs with their private pools, events, streams) are released first: the
            # train leg starts from the allocator state a fresh `--mode train` process has
            pipe = pipe2 = eng = events = alone = None
            net._fused = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            tr = run_train(a, net, hosts[:4], dev, dist, world, rank, a.train_steps, 5, live_traffic=(live or {}).get("train") if rank == 0 else None)
        except Exception as e:                  # never lose the headline over the extra leg
            tr = {"error": repr(e)[:300]}
        if rank == 0:
            res["train"] = tr
        net.eval()
    if rank == 0 and not a.no_irregular:
        try:
            from ratrack_amd import benchutil
            pm = _pmc("irregular", a.batch, a.npoints)
            res["roofline_irregular"] = benchutil.irregular_ops(a.batch, a.npoints, dev, pmc=(pm or {}).get("traffic_bytes_per_launch"))
        except Exception as e:
            res["roofline_irregular"] = {"error": repr(e)[:300]}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(a.npoints)
        except Exception as e:
            res["cpu_baseline"] = {"error": repr(e)[:300]}
    finish(res)


if __name__ == "__main__":
    main()
