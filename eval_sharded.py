#!/usr/bin/env python3
"""Multi-GPU form of the reference's evaluation run (main.py --is_eval: main.py:628-653 and
produce_evaluation_file, main.py:158-199), one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        eval_sharded.py --track DF --model_path models/best.pth --cp_path xlsr2_300m.sd.pt \
        --protocols_path .../ASVspoof2021.DF.cm.eval.trl.txt --database_path .../ASVspoof2021_DF_eval \
        --eval_output scores/scores_DF.txt
    python evaluate_2021_DF.py scores/scores_DF.txt ./keys eval      # the reference's scorer, unchanged

Same flags as main.py where they reach the model or the data (main.py:402-437).  Rank r scores trials
r, r + W, ... (native FLAC / WAV decode into pinned memory ahead of the forward), the ranks' scores
are gathered with one all_gather over RCCL and rank 0 writes "utt score" lines in protocol order.
--variable_length scores every clip at its own length (up to --max_seconds) in length-bucketed
padded batches instead of cutting / tiling it to 64600 samples (in-tree heads only).
A single process (no torchrun) scores everything on cuda:0.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import slsmi  # noqa: E402,F401
from slsmi import data, scoring  # noqa: E402
from slsmi.model import Model, WindowTopKModel  # noqa: E402


def build_model(args, device):
    kw = dict(cp_path=args.cp_path, precision=args.precision)
    if args.head == "sls":
        return Model(args, device, **kw)
    kw.update(use_sae=True, use_sparse_features=args.use_sparse_features, sae_dict_size=args.sae_dict_size,
              sae_k=args.sae_k, sae_weight=args.sae_weight)
    if args.use_window_topk:
        return WindowTopKModel(args, device, sae_window_size=args.sae_window_size, **kw)
    return Model(args, device, **kw)


def load_trained(model, path, backbone_given: bool, allow_partial: bool, verbose: bool = True) -> int:
    """main.py:562-592: load the trained checkpoint strictly, fall back to a non-strict load -- but never
    score with weights no file supplied: a key the model needs that the checkpoint lacks is an error
    (exit code 2) unless --allow_partial, or unless it is a backbone key and --cp_path supplied the backbone.
    The file is read as main.py:753-764 writes it (numpy scalars beside the tensors) without executing
    anything from it."""
    from slsmi import checkpoint
    ckpt = checkpoint.safe_load(path)
    try:
        model.load_state_dict(ckpt, strict=True)
        if verbose:
            print("Model loaded with strict=True")
        return 0
    except RuntimeError as e:
        if verbose:
            print(f"Strict loading failed: {e}\nLoading with strict=False...")
    missing, unexpected = model.load_state_dict(ckpt, strict=False)
    needed = [k for k in missing if not (backbone_given and k.startswith("ssl_model."))
              and not k.endswith("num_batches_tracked")]
    if verbose:
        print(f"load_state_dict: {len(missing)} missing, {len(unexpected)} unexpected keys")
    if needed and not allow_partial:
        if verbose:
            print(f"Error: {path} lacks {len(needed)} weights the model needs (first: {needed[:4]}); they would "
                  "stay at their random initial values.  Pass --allow_partial to score anyway.")
        return 2
    return 0


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--database_path", required=True)
    ap.add_argument("--protocols_path", required=True)
    ap.add_argument("--track", default="DF", choices=["LA", "In-the-Wild", "DF"])
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--cp_path", default=None, help="XLS-R backbone weights (plain state dict); omit to take the "
                                                    "backbone from --model_path alone")
    ap.add_argument("--model_path", default=None, help="trained checkpoint (main.py:753-776 format or a state dict)")
    ap.add_argument("--head", default="sae", choices=["sae", "sls"],
                    help="sae: the in-tree model main.py builds; sls: the upstream Model(args, device)")
    ap.add_argument("--sae_weight", type=float, default=0.1)
    ap.add_argument("--sae_dict_size", type=int, default=4096)
    ap.add_argument("--sae_k", type=int, default=128)
    ap.add_argument("--use_window_topk", action="store_true")
    ap.add_argument("--sae_window_size", type=int, default=8)
    ap.add_argument("--use_sparse_features", action="store_true", default=True)
    ap.add_argument("--eval_output", default=None)
    ap.add_argument("--quick_test", action="store_true", help="first 5 batches per rank only (main.py:173-175)")
    ap.add_argument("--precision", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--log_scores", action="store_true", help="write log-probabilities (the upstream SLS convention)")
    ap.add_argument("--variable_length", action="store_true")
    ap.add_argument("--max_seconds", type=float, default=10.0)
    ap.add_argument("--max_frames", type=int, default=0,
                    help="with --variable_length: frame budget of a batch (clips x frames of its padded length; batches of "
                         "up to 4 x --batch_size clips); 12896 fills one round of the encoder GEMM tiles; 0 = --batch_size clips")
    ap.add_argument("--decode_threads", type=int, default=16)
    ap.add_argument("--synthetic_weights", action="store_true", help="seeded random weights (smoke runs)")
    ap.add_argument("--range_report", action="store_true",
                    help="print, per layer, the largest |value| of every 16-bit store of the forward and how many values "
                         "saturated (fixed-length runs; this rank's share): whether --precision f16 holds for this checkpoint")
    ap.add_argument("--allow_partial", action="store_true",
                    help="score even if --model_path lacks weights the model needs (they keep their initial values)")
    args = ap.parse_args()
    if not args.model_path and not args.synthetic_weights:
        print("Error: --model_path is required for evaluation mode")  # main.py:630-632
        return 1

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    model = build_model(args, device).eval()
    if args.model_path:
        rc = load_trained(model, args.model_path, backbone_given=args.cp_path is not None,
                          allow_partial=args.allow_partial, verbose=rank == 0)
        if rc:
            return rc
    model.to(device)

    ids = data.genSpoof_list(args.protocols_path, is_train=False, is_eval=True)
    ds_cls = data.Dataset_in_the_wild_eval if args.track == "In-the-Wild" else data.Dataset_ASVspoof2021_eval
    ds = ds_cls(ids, args.database_path)
    out_path = args.eval_output or os.path.join("scores", f"scores_{args.track}.txt")
    if rank == 0:
        os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)

    t0 = time.perf_counter()
    if args.variable_length:
        loader = data.VariableLengthLoader(ds, args.batch_size if args.max_frames <= 0 else 4 * args.batch_size, device,
                                           args.decode_threads, max_len=int(args.max_seconds * data.SAMPLE_RATE),
                                           rank=rank, world=world, max_frames=args.max_frames if args.max_frames > 0 else None)
        if args.quick_test:
            loader.batches = loader.batches[:5]
            ids_scored = sorted(j for idx, _ in loader.batches for j in idx)
            assert world == 1, "--quick_test with --variable_length runs on one rank"
            scores = scoring.score_variable_length(model, loader, ids, None, args.log_scores)
            scoring.write_score_file(out_path, [ids[j] for j in ids_scored], scores[ids_scored])
        else:
            scores = scoring.score_variable_length(model, loader, ids, out_path if rank == 0 else None, args.log_scores)
    else:
        mine = scoring.shard_indices(len(ids), rank, world)
        if args.quick_test:
            mine = mine[:5 * args.batch_size]
        n_scored = len(mine) * world if args.quick_test else len(ids)
        loader = data.BatchLoader(ds, args.batch_size, device, args.decode_threads, indices=mine)
        chunks = []
        tracker = scoring.RangeTracker(model._get_engine(device)) if args.range_report else None
        with torch.no_grad():
            for batch, _ in loader:
                logp = model(batch, return_sae_loss=False)
                chunks.append(logp[:, 1] if args.log_scores else torch.exp(logp)[:, 1])
                if tracker is not None:
                    tracker.update()
        if tracker is not None:
            tracker.close()
            if rank == 0:
                limit = 65504.0 if args.precision == "f16" else 3.39e38
                print("\n".join(tracker.lines(limit)))
                bad = tracker.saturated()
                print(f"range report: {len(bad)} of {len(tracker.rows)} stores saturated" +
                      (f" (first: {bad[0][0]}, {bad[0][1]} values): consider --precision bf16" if bad else
                       f": --precision {args.precision} holds on this rank's {len(mine)} clips"))
        local_scores = torch.cat(chunks) if chunks else torch.zeros(0, device=device)
        if args.quick_test and world > 1:
            raise SystemExit("--quick_test runs on one rank")
        scores = scoring.gather_scores(local_scores, n_scored if not args.quick_test else len(mine))
        if rank == 0:
            if os.path.exists(out_path):
                os.remove(out_path)  # main.py:646-649
            scoring.write_score_file(out_path, [ids[j] for j in (mine if args.quick_test else range(len(ids)))], scores)
    if rank == 0:
        dt = time.perf_counter() - t0
        print(f"Scores saved to: {out_path}  ({len(scores)} trials, {len(scores) / dt:.0f} utt/s on {world} GPU(s))")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
