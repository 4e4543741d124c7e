#!/usr/bin/env python3
"""Throughput of the scoring hot path: XLS-R-300M + SLS head forward on batches
of synthetic 4.04 s / 16 kHz clips (64600 samples), 64 clips per GPU per step.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run, one rank per GPU; utterance batches
are sharded across ranks with no data-path collective (weak scaling); the only
exchange is the score gather, done once after the timed region.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import slsmi  # noqa: E402
from slsmi import _lib, ops  # noqa: E402
from slsmi.config import XLSR_300M, flops_per_utt  # noqa: E402
from slsmi.model import Model  # noqa: E402
from slsmi.weights import synth_state_dict  # noqa: E402

PEAK_BF16_DENSE_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (vendor spec, no sparsity)
BATCH = 64
SAMPLES = 64600


def gemm_shapes(batch: int):
    """(M, N, K, lda, mode, count) of every launch of the 128-wide GEMM kernel
    in one forward of `batch` utterances (conv layers as flat GEMMs over the
    padded row layout 208 * 2^(6-i) rows per utterance)."""
    T, rows = 201, 208
    shapes = []
    for i, k in enumerate((3, 3, 3, 3, 2, 2)):
        shapes.append((batch * rows * 2 ** (5 - i), 512, k * 512, 1024, _lib.GEMM_BF16, 1))
    M = batch * T
    shapes.append((M, 1024, 512, 512, _lib.GEMM_F32, 1))
    shapes.append((M, 3072, 1024, 1024, _lib.GEMM_BF16, 24))
    shapes.append((M, 1024, 1024, 1024, _lib.GEMM_F32_RESID, 24))
    shapes.append((M, 4096, 1024, 1024, _lib.GEMM_BF16_GELU, 24))
    shapes.append((M, 1024, 4096, 4096, _lib.GEMM_F32_RESID, 24))
    return shapes


def gemm_algorithmic_bytes(batch: int) -> float:
    """HBM bytes one step's GEMM launches must move at least: every operand row and weight
    read once (overlapping conv rows counted once), every output written once, the fp32
    residual stream read and written by the two in-place GEMMs of a layer."""
    total = 0.0
    for (M, N, K, lda, mode, count) in gemm_shapes(batch):
        a_bytes = ((M - 1) * lda + K) * 2.0 if lda < K else M * K * 2.0
        c_elem = 2.0 if mode in (_lib.GEMM_BF16, _lib.GEMM_BF16_GELU) else 4.0
        c_bytes = M * N * c_elem * (2.0 if mode == _lib.GEMM_F32_RESID else 1.0)
        total += count * (a_bytes + N * K * 2.0 + c_bytes)
    return total


PMC_FILE = os.path.join("profiles", "r03_pmc_bench.json")


def library_sha16() -> str:
    """First 16 hex digits of the SHA-256 of the loaded libslsmi.so: names the build a counter file belongs to."""
    import hashlib
    with open(_lib.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def pmc_traffic(kernel_prefix: str):
    """HBM-side bytes per launch and MFMA-busy fraction of the dominant kernel from the rocprofv3 counter
    passes in profiles/r03_pmc_bench.json (tools/gpu_pmc_bench.sh: FETCH_SIZE doubled as the gfx950 guide
    prescribes, plus WRITE_SIZE).  Counters cannot be read from inside the timed run, so the file is used
    only when it was collected on the very library build that is loaded now (it records the build's hash);
    otherwise (None, None, reason)."""
    path = os.path.join(ROOT, PMC_FILE)
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("libslsmi_sha16") != library_sha16():
            return None, None, f"{PMC_FILE} was collected on another build of libslsmi.so"
        rec = next(v for name, v in doc["kernels"].items() if name.startswith(kernel_prefix))
        return rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"], rec.get("mfma_busy_frac"), None
    except (OSError, KeyError, StopIteration, ValueError) as e:
        return None, None, f"{PMC_FILE}: {type(e).__name__}"


def time_gemm_kernel(dev, batch: int, dtype, reps: int = 3):
    """Times every GEMM shape of one step (same kernel, random operands) with
    HIP events recorded on the launch stream inside the library.  Returns
    (flops_per_step, seconds_per_step, launches_per_step, per-shape detail)."""
    total_flops, total_sec, launches, detail = 0.0, 0.0, 0, []
    for (M, N, K, lda, mode, count) in gemm_shapes(batch):
        elems = (M - 1) * lda + K + 64
        a = (torch.randn(elems, device=dev) * 0.5).to(dtype)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dtype)
        b = torch.zeros(N, device=dev)
        odt = dtype if mode in (_lib.GEMM_BF16, _lib.GEMM_BF16_GELU) else torch.float32
        out = torch.zeros(M, N, dtype=odt, device=dev)
        sec = ops.time_gemm_bf16(a, w, b, out, mode, M, K, lda, reps=max(reps, min(count, 8)))
        fl = 2.0 * M * N * K
        detail.append({"M": M, "N": N, "K": K, "count": count, "us": 1e6 * sec, "tflops": fl / sec / 1e12})
        total_flops += fl * count
        total_sec += sec * count
        launches += count
        del a, w, out
    return total_flops, total_sec, launches, detail


def cpu_baseline(wav_host):
    """The CPU oracle (fp32 PyTorch restatement of the reference forward) on the first clips of the timed
    batch, all host threads of this process: single clips (B = 1) and the reference's evaluation batch
    (B = 20, main.py:161); about 10-20 s of CPU work in all.  `value` is the better of the two.  Also
    returns the oracle's scores of the B = 20 run (the checker of `score_check`)."""
    from oracle import reference_cpu as orc
    sd = synth_state_dict(XLSR_300M, "sls", 201, 1234)
    wav = wav_host[:20]
    cfg = orc.OracleConfig()
    res = {}
    logp = None
    for bsz, runs in ((1, 16), (min(20, len(wav)), 4)):
        x = wav[:bsz]
        for _ in range(2 if bsz == 1 else 1):
            orc.model_forward(sd, x, cfg, "sls")
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            logp = orc.model_forward(sd, x, cfg, "sls")
            ts.append(time.perf_counter() - t0)
        ts.sort()
        res[bsz] = ts[len(ts) // 2]
    thr = {b: b / t for b, t in res.items()}
    best = max(thr, key=thr.get)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    big = max(res)
    return ({"value": thr[best], "unit": "utt/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model, "kind": "port",
             "sample": f"fp32 oracle on 64600-sample clips: 16 runs at B=1 and 4 runs at B={big} (median each, after "
                       f"warm-up); value = B={best}",
             "utt_per_s_b1": thr[1], f"utt_per_s_b{big}": thr[big], "median_s_b1": res[1], f"median_s_b{big}": res[big]},
            torch.exp(logp)[:, 1].float())


def launch_ranks(args) -> int:
    """`--gpus N` with N > 1 outside torch.distributed.run: start N ranks (one per GPU, RCCL) as a CHILD
    process before this process makes any GPU call, and return the child's exit code.  Replaces the
    reference's single-process nn.DataParallel (main.py:518)."""
    import socket
    import subprocess
    rehearsal = os.environ.get("SLSMI_BENCH_REHEARSAL") == "1"
    visible = torch.cuda.device_count()   # counts devices without initialising one
    if not rehearsal and visible < args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this machine shows {visible} "
              "(SLSMI_BENCH_REHEARSAL=1 runs every rank on cuda:0 over gloo to rehearse the code path)",
              file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def init_ranks(args):
    """(world, rank, device, dist or None, device collectives run on).  Every rank takes the GPU of its
    LOCAL_RANK; N > 1 joins an RCCL ("nccl") process group.  SLSMI_BENCH_REHEARSAL=1: every rank on cuda:0
    with gloo, to run the N > 1 code path on a one-GPU machine (the numbers it prints mean nothing)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    rehearsal = os.environ.get("SLSMI_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local}, {torch.cuda.device_count()} GPUs visible")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    return world, rank, dev, dist, (torch.device("cpu") if rehearsal else dev)


def max_over_ranks(elapsed: float, dist, comm_dev) -> float:
    if dist is None:
        return elapsed
    t = torch.tensor([elapsed], device=comm_dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def bench_train(args):
    """Training throughput (a second, separate line; the headline metric is the eval forward):
    one step = forward + backward + Adam on `batch` clips per GPU (main.py:224-238; batch 5 per GPU as in
    the reference's recipe), gradients averaged over the ranks with one flat all-reduce per step."""
    from slsmi import training
    world, rank, dev, dist, comm_dev = init_ranks(args)
    batch = args.batch if args.batch != BATCH else 5
    thead = "sls" if args.train_model == "sls" else "sae"
    if thead == "sls":   # the upstream model, Model(args, device): README.md:46
        model = Model(None, dev, cp_path=None, precision=args.precision).train()
    else:                # the in-tree model main.py builds by default (model.py:150-193)
        model = Model(None, dev, cp_path=None, precision=args.precision, use_sae=True).train()
    model.load_state_dict(synth_state_dict(XLSR_300M, thead, 201, 1234))
    model.to(dev)
    training.freeze_unused(model)
    params = [p for p in model.parameters() if p.requires_grad]
    fused_adam = world == 1 and not (args.no_fused_adam or args.sharded or args.torch_adam)
    if args.sharded:   # reduce-scatter, Adam on the rank's shard, all-gather (RCCL; not for gloo rehearsals through host memory)
        opt = training.ShardedAdam(params, lr=1e-6, weight_decay=1e-4, model=model,
                                   comm_dtype=torch.bfloat16 if args.sharded == "bf16" else torch.float32)
    elif args.torch_adam:
        opt = torch.optim.Adam(params, lr=1e-6, weight_decay=1e-4, fused=True)   # main.py:526 (torch's fused kernel)
    else:
        # the same step, every tensor in one HIP launch; on one GPU the encoder layers' matrices take it inside the
        # backward, where their gradient is produced (same bits; --no-fused-adam: backward, then the step)
        opt = training.Adam(params, lr=1e-6, weight_decay=1e-4, fuse_backward=model if fused_adam else None)
    # (a rehearsal's gloo group exchanges through host memory)
    sync = training.GradSync(params, comm_device=None if comm_dev == dev else comm_dev) if world > 1 and not args.sharded else None
    g = torch.Generator().manual_seed(1234 + rank)
    wav = (0.1 * torch.randn(batch, SAMPLES, generator=g)).clamp(-1, 1).to(dev)
    y = (torch.arange(batch, device=dev) % 2)
    for _ in range(args.warmup):
        training.train_step(model, opt, wav, y, grad_sync=sync)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _, _ = training.train_step(model, opt, wav, y, grad_sync=sync)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, comm_dev)
    if rank == 0:
        fl = 3.0 * flops_per_utt(XLSR_300M, SAMPLES, thead)   # forward + two backward contractions per forward one
        print(json.dumps({
            "metric": "training utterances/sec whole-node (XLS-R-300M + " + ("SLS" if thead == "sls" else "TopK-SAE") +
                      " head, forward + backward + Adam)",
            "value": world * batch * args.steps / elapsed, "unit": "utt/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision + " forward, bf16 backward",
            "rccl_ranks": world if dist is not None and dist.get_backend() == "nccl" else 0,
            "data": "synthetic",
            "config": {"workload": f"training step, batch={batch} clips of {SAMPLES} samples per GPU, " +
                                   ("upstream SLS model" if thead == "sls" else "in-tree model") + ", "
                                   "Adam lr 1e-6 wd 1e-4 (" + ("slsmi.training.ShardedAdam" if args.sharded else
                                                               "torch fused" if args.torch_adam else
                                                               "slsmi.training.Adam, encoder-layer matrices stepped inside the backward"
                                                               if fused_adam else "slsmi.training.Adam") +
                                   "), seeded random XLS-R-300M weights",
                       "global_batch": world * batch, "parallelism": f"data-parallel x{world}, " + ("gradients reduce-scattered, optimizer state sharded, parameters all-gathered"
                                                                     if args.sharded else "flat gradient all-reduce")},
            "approx_step_tflops": batch * fl / (elapsed / args.steps) / 1e12, "final_loss": float(loss)}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_varlen(args):
    """Variable-length scoring (BASELINE's In-the-Wild configuration; a separate line, the headline is the fixed-length
    forward): `--clips` clips per GPU with lengths uniform in [1 s, 10 s] (seed 1234 + rank), cut into
    length-bucketed padded batches of at most `batch` clips (data.bucket_batches), every batch through
    Model.forward(batch, lengths=) with the SAE head (the SLS head's fc1 fixes 201 frames).  One step = one pass
    over all of a rank's batches; inputs are resident in HBM."""
    import numpy as np
    from slsmi import data
    world, rank, dev, dist, comm_dev = init_ranks(args)
    model = Model(None, dev, cp_path=None, precision=args.precision, use_sae=True).eval()
    model.load_state_dict(synth_state_dict(XLSR_300M, "sae", 201, 1234))
    model.to(dev)
    rng = np.random.default_rng(1234 + rank)
    lengths = rng.integers(16000, 160001, size=args.clips)
    batches = data.bucket_batches(lengths, batch_size=args.batch if args.max_frames <= 0 else 4 * args.batch,
                                  max_frames=args.max_frames if args.max_frames > 0 else None)
    g = torch.Generator().manual_seed(1234 + rank)
    resident = []
    for idx, padded in batches:
        wav = (0.1 * torch.randn(len(idx), padded, generator=g)).clamp(-1, 1).to(dev)
        resident.append((wav, torch.from_numpy(lengths[idx].astype(np.int32)).to(dev)))
    padded_samples = sum(len(i) * p for i, p in batches)

    def one_pass():
        out = None
        for wav, lens in resident:
            out = model(wav, return_sae_loss=False, lengths=lens)
        return out

    with torch.no_grad():
        for _ in range(args.warmup):
            one_pass()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = one_pass()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        elapsed = max_over_ranks(time.perf_counter() - t0, dist, comm_dev)
    if rank == 0:
        utt = world * args.clips * args.steps
        print(json.dumps({
            "metric": "variable-length utterances/sec whole-node (XLS-R-300M + TopK-SAE head, eval, clips of 1-10 s)",
            "value": utt / elapsed, "unit": "utt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "rccl_ranks": world if dist is not None and dist.get_backend() == "nccl" else 0,
            "config": {"workload": f"eval forward over {args.clips} clips per GPU, lengths uniform in [16000, 160000] "
                                   f"samples, {len(batches)} length-bucketed batches of at most " +
                                   (f"{args.batch} clips" if args.max_frames <= 0 else
                                    f"{args.max_frames} frames (and {4 * args.batch} clips)") + ", head=sae",
                       "global_batch": world * args.clips, "parallelism": f"utterance-sharded x{world}"},
            "audio_seconds_per_second": world * float(lengths.sum()) / 16000.0 * args.steps / elapsed,
            "padding_overhead": padded_samples / float(lengths.sum()) - 1.0,
            "outputs_finite": bool(torch.isfinite(last).all().item())}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def timed_forward(model, wav, steps, warmup, dev, dist):
    """`warmup` untimed and `steps` timed forwards, bracketed by a barrier + device synchronise on both
    sides.  Returns (host seconds, device seconds between two events on the launch stream, last output,
    all outputs finite)."""
    outs = []
    with torch.no_grad():
        for _ in range(warmup):
            model(wav, return_sae_loss=False)
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(torch.cuda.current_stream(dev))
        for _ in range(steps):
            outs.append(model(wav, return_sae_loss=False))
        e1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
    finite = bool(torch.isfinite(torch.stack(outs)).all().item())
    return elapsed, e0.elapsed_time(e1) * 1e-3, outs[-1], finite


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--head", default="sls", choices=["sls", "sae"])
    ap.add_argument("--precision", default="f16", choices=["f16", "bf16"],
                    help="16-bit operand format of the MFMA GEMMs (fp32 accumulate either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the second (other operand format) run")
    ap.add_argument("--low-latency", action="store_true",
                    help="small batches: let the scoring forward cut GEMMs along K (slsmi_set_low_latency)")
    ap.add_argument("--no-fused-adam", action="store_true",
                    help="--mode train, one GPU: slsmi.training.Adam without fuse_backward=model (the encoder layers' weight "
                         "matrices then get a stored gradient and step with the others, instead of in the epilogue of their "
                         "weight-gradient kernel)")
    ap.add_argument("--sharded", nargs="?", const="f32", default=None, choices=["f32", "bf16"],
                    help="train mode: slsmi.training.ShardedAdam (gradient bytes on the links in fp32 or bfloat16)")
    ap.add_argument("--train-model", default="intree", choices=["intree", "sls"],
                    help="--mode train: the in-tree model (TopK-SAE head, what main.py builds) or the upstream SLS model")
    ap.add_argument("--torch-adam", action="store_true",
                    help="--mode train: step with torch.optim.Adam(fused=True) instead of slsmi.training.Adam")
    ap.add_argument("--clips", type=int, default=1024, help="--mode varlen: clips per GPU and pass")
    ap.add_argument("--max-frames", type=int, default=12896,
                    help="--mode varlen: frame budget of a batch (data.bucket_batches max_frames); 0 = batches of --batch clips")
    ap.add_argument("--mode", default="eval", choices=["eval", "train", "varlen"],
                    help="eval (default): the scoring forward, the headline metric.  train: the training step of "
                         "main.py (forward + backward + Adam, in-tree model, batch 5 per GPU unless --batch is given).  "
                         "varlen: scoring clips of 1-10 s in length-bucketed batches")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))   # no GPU call has been made in this process
    if args.mode == "train":
        return bench_train(args)
    if args.mode == "varlen":
        return bench_varlen(args)

    world, rank, dev, dist, comm_dev = init_ranks(args)
    sd = synth_state_dict(XLSR_300M, args.head, 201, 1234)
    model = Model(None, dev, cp_path=None, head=args.head, precision=args.precision,
                  low_latency=args.low_latency).eval()
    model.load_state_dict(sd)
    model.to(dev)
    g = torch.Generator().manual_seed(1234 + rank)
    wav_host = (0.1 * torch.randn(args.batch, SAMPLES, generator=g)).clamp(-1, 1)
    wav = wav_host.to(dev)  # resident in HBM

    elapsed, dev_elapsed, last, finite = timed_forward(model, wav, args.steps, args.warmup, dev, dist)
    elapsed = max_over_ranks(elapsed, dist, comm_dev)
    if dist is not None:
        # the path's one exchange step, outside the timed region: gather the last step's scores
        from slsmi.scoring import gather_scores
        gathered = gather_scores(torch.exp(last)[:, 1].to(comm_dev), world * args.batch)
        assert (gathered is not None and len(gathered) == world * args.batch) if rank == 0 else gathered is None

    # the other 16-bit operand format on the same clips (BASELINE's second configuration names bf16; f16 is
    # the default because only it meets the 1e-3 score tolerance): rank 0 of a single-GPU run only
    alt = None
    if world == 1 and not args.no_alt and not args.low_latency:
        other = "bf16" if args.precision == "f16" else "f16"
        m2 = Model(None, dev, cp_path=None, head=args.head, precision=other).eval()
        m2.load_state_dict(sd)
        m2.to(dev)
        a_el, _, a_last, a_fin = timed_forward(m2, wav, max(3, args.steps // 4), 2, dev, None)
        alt = {"dtype": other, "value": args.batch * max(3, args.steps // 4) / a_el, "unit": "utt/s",
               "ms_per_step": 1e3 * a_el / max(3, args.steps // 4), "outputs_finite": a_fin}
        alt_scores = torch.exp(a_last)[:, 1].float().cpu()
        del m2
    del sd
    main_scores = torch.exp(last)[:, 1].float().cpu()

    if rank == 0:
        utt = world * args.batch * args.steps
        value = utt / elapsed
        fl_utt = flops_per_utt(XLSR_300M, SAMPLES, args.head)
        step_tflops = args.batch * fl_utt / (dev_elapsed / args.steps) / 1e12
        # the MFMA GEMM launches timed where they run: a few more forwards (outside the timed region) with every
        # GEMM launch bracketed by two HIP events on its launch stream
        in_place_steps = max(1, min(args.steps, 5))
        with torch.no_grad(), ops.gemm_timing(dev, max_launches=256 * in_place_steps) as gt:
            for _ in range(in_place_steps):
                model(wav, return_sae_loss=False)
        gemm_in_place = {
            "launches_per_step": gt.launches / in_place_steps, "untimed_launches": gt.dropped,
            "avg_launch_us": 1e6 * gt.seconds / max(gt.launches, 1),
            "ms_per_step": 1e3 * gt.seconds / in_place_steps,
            "gflop_per_launch": gt.flops / max(gt.launches, 1) / 1e9,
            "tflops": gt.flops / gt.seconds / 1e12 if gt.seconds > 0 else None,
            "frac": gt.flops / gt.seconds / 1e12 / PEAK_BF16_DENSE_TFLOPS if gt.seconds > 0 else None,
            "method": f"{in_place_steps} forwards after the timed region, every MFMA GEMM launch between two HIP events "
                      "recorded on its launch stream (slsmi_debug_gemm_timing_begin / _end); sum of the layers' 2 M N K "
                      "(a launch on split hi + lo operands counts its layer's K once) over the sum of the launches' durations"}
        gflops, gsec, launches, gdetail = time_gemm_kernel(dev, args.batch, ops.op_dtype(args.precision))
        # what back-to-back MFMAs from registers reach on this box (the chip clocks down under full matrix load):
        # the bound of any GEMM K loop here, reported beside the fraction of the data-sheet peak
        mfma_loop = ops.mfma_rate(dev, ops.op_dtype(args.precision))
        # the path's roofline figure: whole-job utterances/s x algorithmic FLOP per utterance over the dense
        # 16-bit MFMA peak of the GPUs used (every kernel, launch gap and the host loop included)
        ach = value * fl_utt / world / 1e12
        traffic, mfma_busy, pmc_note = (pmc_traffic("gemm16") if args.batch == BATCH and args.precision == "f16"
                                        else (None, None, "counters are collected at batch 64, f16 only"))
        out = {
            "metric": "4 s/16 kHz utterances/sec whole-node (XLS-R-300M + SLS forward, eval)",
            "value": value, "unit": "utt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "rccl_ranks": world if dist is not None and dist.get_backend() == "nccl" else 0,
            "config": {"workload": f"eval forward, batch={args.batch} clips of {SAMPLES} samples per GPU, "
                                   f"head={args.head}, seeded random XLS-R-300M weights",
                       "global_batch": world * args.batch, "samples_per_clip": SAMPLES,
                       "parallelism": f"utterance-sharded x{world}"},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_BF16_DENSE_TFLOPS,
                         "definition": "utt/s x 148.82 GFLOP per utterance / (n_gpus x 2.5 PFLOP/s): the whole path, "
                                       "host loop and every non-GEMM kernel included",
                         "traffic": traffic, "traffic_note": pmc_note,
                         "traffic_unit": f"HBM-side bytes per GEMM launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, {PMC_FILE})",
                         "algorithmic_bytes_per_launch": gemm_algorithmic_bytes(args.batch) / launches,
                         "mfma_busy_pmc": mfma_busy, "libslsmi_sha16": library_sha16(),
                         "kernel": "the MFMA GEMM kernels (all launches of one step)",
                         "launches_per_step": launches, "avg_launch_us_isolated": 1e6 * gsec / launches,
                         "algorithmic_gflop_per_step": gflops / 1e9,
                         "gemm_in_place": gemm_in_place,
                         "gemm_isolated_tflops": gflops / gsec / 1e12,
                         "gemm_isolated_frac": gflops / gsec / 1e12 / PEAK_BF16_DENSE_TFLOPS,
                         "device_step_tflops": step_tflops,
                         "mfma_loop_tflops": mfma_loop,
                         "mfma_loop_note": "register-only 16x16x32 MFMA loop on every SIMD of this GPU, measured in this "
                                           "run (slsmi_debug_mfma_rate); achieved / mfma_loop_tflops = "
                                           f"{ach / mfma_loop:.3f}",
                         "gflop_per_utt": fl_utt / 1e9, "shapes": gdetail},
            "outputs_finite": finite,
        }
        if not args.no_cpu_baseline and world == 1:
            cpu, oracle_scores = cpu_baseline(wav_host)
            out["cpu_baseline"] = cpu
            n = len(oracle_scores)
            out["score_check"] = {"clips": n, "against": "fp32 CPU oracle on the first clips of the timed batch",
                                  args.precision: float((main_scores[:n] - oracle_scores).abs().max())}
            if alt is not None:
                out["score_check"][alt["dtype"]] = float((alt_scores[:n] - oracle_scores).abs().max())
        if alt is not None:
            out["alt"] = alt
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
