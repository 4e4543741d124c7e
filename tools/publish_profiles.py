#!/usr/bin/env python3
"""Copies the summaries tools/gpu_profiles.sh left under gpurun_out/ into profiles/r03_*, each with a header that
says what was run and which library build it belongs to.  usage: publish_profiles.py <ms per bench step> <ms per train step>"""
import hashlib
import os
import re
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
pr = os.path.join(root, "profiles")
sha = hashlib.sha256(open(os.path.join(root, "slsforasvspoof-2021-df_amd", "libslsmi.so"), "rb").read()).hexdigest()[:16]
bench_ms = sys.argv[1] if len(sys.argv) > 1 else "11.7"
train_ms = sys.argv[2] if len(sys.argv) > 2 else "11.3"


def body(path):
    return open(path).read()


def launches(text):
    m = re.search(r"(\d+) launches", text)
    return m.group(1) if m else "?"


one = body(os.path.join(go, "prof_final", "one_forward.md"))
open(os.path.join(pr, "r03_one_forward.md"), "w").write(
    "# One scoring forward (B = 64 clips of 64600 samples, SLS head, f16 operands) cut out of the kernel trace of the bench command\n\n"
    "`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt`, "
    "then `tools/one_forward.py <kernel_trace.csv> --seq`: the launches of ONE forward only (no isolated-shape timings, no "
    f"`mfma_rate_kernel`, no weight set-up), first summed by kernel, then in launch order. Library sha `{sha}`. Template arguments of "
    "`gemm16_wide_kernel<format, mode, LayerNorm fold, periodic A>`: mode 0 16-bit out (q/k/v, conv layers 1-2), 1 16-bit GELU (fc1), "
    "2 fp32 out (conv layers 3-5 on split weights, the projection), 3 fp32 residual (out-proj, fc2); fold 1 = emits the next "
    "LayerNorm's operand, 2 = applies the previous one. Gaps between launches are wider under the profiler than in the timed bench "
    f"run ({bench_ms} ms per step on this kind of box).\n\n" + one)
ks = body(os.path.join(go, "prof_final", "kernel_stats.md"))
open(os.path.join(pr, "r03_bench_kernel_stats_v1.md"), "w").write(
    "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt\n\n"
    "Whole run (12 forwards: warm-up, timed steps, the in-place GEMM timing forwards; plus the isolated-shape timings, two "
    "`mfma_rate_kernel` launches and weight set-up): per-forward figures are in `r03_one_forward.md`. "
    f"Library sha `{sha}`.\n\n" + ks)
shutil.copy(os.path.join(go, "prof_final", "r_kernel_stats.csv"), os.path.join(pr, "r03_bench_kernel_stats_v1.csv"))
for tag, name, what in (("prof_train", "r03_train_one_step.md", "in-tree model"),
                        ("prof_train_sls", "r03_train_sls_one_step.md", "upstream SLS model, `--train-model sls`")):
    st = body(os.path.join(go, tag, "one_step.md"))
    flag = " --train-model sls" if "sls" in tag else ""
    open(os.path.join(pr, name), "w").write(
        f"# One training step ({what}, batch 5 clips of 64600 samples, f16 forward / bf16 backward, slsmi.training.Adam(fuse_backward=model): the encoder layers' weight matrices step inside the backward) cut out "
        "of the kernel trace\n\n"
        f"`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --mode train{flag} --steps 5 --warmup 2`, then "
        f"`tools/train_step_prof.py <kernel_trace.csv> 45`: the launches between two loss evaluations. Library sha `{sha}`. bench.py "
        f"without the profiler on the same kind of box: {train_ms} ms per step (the span below is stretched by the profiler's "
        f"per-launch cost over {launches(st)} launches).\n\n" + st)
shutil.copy(os.path.join(go, "prof_train", "tr_kernel_stats.csv"), os.path.join(pr, "r03_train_kernel_stats.csv"))
print("published for library", sha)
