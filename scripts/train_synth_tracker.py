#!/usr/bin/env python3
"""Stand-in weights for the closed-loop tests: se(3)-TrackNet trained (PyTorch autograd through the oracle's forward, on the CPU or --
as the committed fixture was -- on the MI355X box through PyTorch-ROCm) on the synthetic tracking problem of oracle/synth_track.py,
so that the loop of predict.py:416-420 is contractive as it is with the reference's pretrained weights (which are not available
offline).  TEST-FIXTURE GENERATOR, not product code; training itself is out of the hot path's scope.

Only a SUBSET of the state_dict is trained (stems, the 64-channel blocks, convAB1, convAB2, every BN affine, the two FC layers: 1.7 M
parameters); every other tensor (the heads' 512-channel convs: 10.6 M of the 13.5 M parameters) stays O.make_state_dict(BASE_SEED).
The fixture tests/golden/synth_tracker.npz holds the trained tensors rounded to float16 (both sides load the same float32 values), the
mean / std of the training set and the held-out errors.  What it took (profiles/r06_synth_tracker_training.log): translation learns
from 20 k pairs in ~400 steps; rotation memorises 20 k pairs (train loss 0, held-out residual ratio 0.75) and needs FRESH data: with
--regen-every the training set is replaced every 4,000 steps (320 k distinct pairs): held-out ratios 0.061 / 0.205.

    python scripts/train_synth_tracker.py --device cuda --samples 20000 --val 500 --steps 64000 --batch 64 --lr 2e-3 --rot-weight 2 \
        --workers 15 --eval-every 4000 --regen-every 4000 --out tests/golden/synth_tracker.npz          # 16 minutes on the GPU box"""
import argparse
import os
import re
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

BASE_SEED = 0
TRAINABLE = r"^(convA1|convB1|convA2|convB2|convB3|convAB1|convAB2)\.|\.(bn1|bn2|1)\.(weight|bias)$|^(trans_out|rot_out)\."


REGIME = None          # set from --regime


def gen_data(n, K, workers, per_job=100, first_seed=0, pool=None):
    from oracle import free_run as FR, synth_track as ST
    jobs = [dict(seed=first_seed + j, n=min(per_job, n - j * per_job), K=K, regime=REGIME) for j in range((n + per_job - 1) // per_job)]
    if pool is not None:
        parts = list(pool.map(ST.training_samples, jobs))
    else:
        with FR._pool(workers) as own:
            parts = list(own.map(ST.training_samples, jobs))
    return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=6000)
    ap.add_argument("--val", type=int, default=400)
    ap.add_argument("--steps", type=int, default=900)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--data", default="/tmp/synth_train.npz")
    ap.add_argument("--out", default="tests/golden/synth_tracker.npz")
    ap.add_argument("--regime", default="ycbineoat_30deg", choices=["ycbineoat_30deg", "ycb_video_5deg"],
                    help="normalisers the labels are in: predict.py:586 (0.03 m, 30 degrees) or predict.py:128 (0.03 m, 5 degrees)")
    ap.add_argument("--device", default="cpu", help="cpu | cuda: torch device of the training loop (the fixture was trained on the MI355X "
                                                     "box through PyTorch-ROCm: a test-fixture generator may use any tool; the PRODUCT never sees this)")
    ap.add_argument("--eval-every", type=int, default=100)
    ap.add_argument("--regen-every", type=int, default=0,
                    help="> 0: every this many steps the training set is REPLACED by --samples freshly generated ones (new seeds; generated "
                         "by the worker pool while the previous chunk trains): the network sees steps / regen-every x samples distinct pairs "
                         "-- 20 k samples alone were memorised (train loss 0, held-out rotation residual ratio 0.75)")
    ap.add_argument("--resume", default=None)
    ap.add_argument("--trainable", default=None, help="regex of the trained state_dict keys (default: TRAINABLE above)")
    ap.add_argument("--clip", type=float, default=0.0, help="> 0: clip the gradient norm (a run under the 5-degree normaliser collapsed to the zero "
                                                           "predictor at the peak of the one-cycle schedule without it)")
    ap.add_argument("--rot-weight", type=float, default=1.0, help="weight of the rotation loss (problems.py:90-91 loss_weights)")
    ap.add_argument("--rot-fc-scale", type=float, default=1.0, help="multiply the (resumed) rot_out FC weights once, before training")
    args = ap.parse_args()
    global REGIME
    REGIME = args.regime
    import torch
    from oracle import free_run as FR, se3_oracle as O, synth_track as ST
    torch.set_num_threads(args.threads)
    K = FR.camera_matrix()
    if os.path.exists(args.data):
        z = np.load(args.data)
        data = {k: z[k] for k in z.files}
    else:
        t0 = time.time()
        data = gen_data(args.samples + args.val, K, args.workers)
        np.savez(args.data, **data)
        print("generated %d samples in %.0f s" % (len(data["zA"]), time.time() - t0), flush=True)
    n = len(data["zA"]) - args.val
    dev = torch.device(args.device)

    def to_dev(d):
        return {k: torch.from_numpy(v if v.dtype != np.uint16 else v.astype(np.int32)).to(dev) for k, v in d.items()}
    T = to_dev(data)
    VAL = {k: v[n:].clone() for k, v in T.items()}         # the held-out pairs stay the same through every regeneration

    def tensors(idx, S=None):
        S = T if S is None else S
        dA = ST.offset_depth_torch(S["depthA"][idx], S["zA"][idx], torch)
        dB = ST.offset_depth_torch(S["depthB"][idx], S["zA"][idx], torch)
        A = torch.cat([S["rgbA"][idx].to(torch.float32), dA[..., None]], 3)
        B = torch.cat([S["rgbB"][idx].to(torch.float32), dB[..., None]], 3)
        return A, B

    # mean / std per channel over (a sample of) the training set, A then B (mean.npy / std.npy layout, predict.py:657-658)
    A, B = tensors(torch.arange(0, min(n, 1000), device=dev))
    mean = torch.cat([A.mean((0, 1, 2)), B.mean((0, 1, 2))]).double().cpu().numpy()
    std = torch.cat([A.std((0, 1, 2)), B.std((0, 1, 2))]).double().cpu().numpy()
    print("mean", mean.round(2), "std", std.round(2), flush=True)
    mean_t, std_t = torch.from_numpy(mean).float().to(dev), torch.from_numpy(std).float().to(dev)

    def batch(idx, S=None):
        A, B = tensors(idx, S)
        S = T if S is None else S
        A = ((A - mean_t[:4]) / std_t[:4]).permute(0, 3, 1, 2).contiguous()
        B = ((B - mean_t[4:]) / std_t[4:]).permute(0, 3, 1, 2).contiguous()
        return A, B, S["trans"][idx], S["rot"][idx]

    sd = O.make_state_dict(BASE_SEED)
    for h in ("trans_out", "rot_out"):        # start the (trainable) FC layers small: tanh away from saturation
        sd[h + ".0.weight"] = sd[h + ".0.weight"] * 0.02
        sd[h + ".0.bias"] = sd[h + ".0.bias"] * 0.0
    if args.resume:
        z = np.load(args.resume)
        for k in z.files:
            if k.startswith("w:"):
                sd[k[2:]] = torch.from_numpy(z[k].astype(np.float32))
    if args.rot_fc_scale != 1.0:
        sd["rot_out.0.weight"] = sd["rot_out.0.weight"] * args.rot_fc_scale
    sd = type(sd)((k, v.to(dev)) for k, v in sd.items())
    pat = re.compile(args.trainable or TRAINABLE)
    params = []
    for k, v in sd.items():
        if v.dtype == torch.float32 and pat.search(k) and "running" not in k:
            v.requires_grad_(True)
            params.append(v)
    print("trainable tensors %d, parameters %d" % (len(params), sum(p.numel() for p in params)), flush=True)
    fwd = O.forward.__wrapped__
    opt = torch.optim.Adam(params, lr=args.lr)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=args.lr, total_steps=args.steps, pct_start=0.15)
    rng = np.random.default_rng(0)
    val_idx = torch.arange(0, args.val, device=dev)

    def evaluate():
        with torch.no_grad():
            et, er = [], []
            for i in range(0, args.val, 50):
                A, B, lt, lr_ = batch(val_idx[i:i + 50], VAL)
                o = fwd(sd, A, B)
                et.append((o["trans"] - lt).abs())
                er.append((o["rot"] - lr_).abs())
            et, er = torch.cat(et), torch.cat(er)
            lt, lr_ = VAL["trans"], VAL["rot"]
            # residual after one step relative to the residual before it: < 1 means the loop contracts
            ratio_t = float(et.norm(dim=1).mean() / lt.norm(dim=1).mean())
            ratio_r = float(er.norm(dim=1).mean() / lr_.norm(dim=1).mean())
            return float(et.mean()), float(er.mean()), ratio_t, ratio_r

    def save(path, val):
        out = {"mean": mean, "std": std, "base_seed": BASE_SEED, "regime": args.regime, "val": np.array(val), "trainable": args.trainable or TRAINABLE}
        for k, v in sd.items():
            if v.requires_grad:
                out["w:" + k] = v.detach().cpu().numpy().astype(np.float16)
        np.savez_compressed(path, **out)

    t0 = time.time()
    regen_pool = regen_future = None
    if args.regen_every > 0:
        from concurrent.futures import ThreadPoolExecutor
        regen_pool = FR._pool(args.workers)
        regen_thread = ThreadPoolExecutor(1)
        chunk = 1
        regen_future = regen_thread.submit(gen_data, n, K, args.workers, 100, 10000 * chunk, regen_pool)
    for step in range(args.steps):
        if regen_future is not None and step > 0 and step % args.regen_every == 0:
            fresh = regen_future.result()
            T = to_dev(fresh)
            chunk += 1
            print("   chunk %d: %d fresh pairs at step %d" % (chunk, len(fresh["zA"]), step), flush=True)
            regen_future = regen_thread.submit(gen_data, n, K, args.workers, 100, 10000 * chunk, regen_pool)
        idx = torch.from_numpy(rng.choice(n, args.batch, replace=False)).to(dev)
        A, B, lt, lr_ = batch(idx)
        o = fwd(sd, A, B)
        loss_t = ((o["trans"] - lt) ** 2).mean()
        loss_r = ((o["rot"] - lr_) ** 2).mean()
        loss = loss_t + args.rot_weight * loss_r
        opt.zero_grad()
        loss.backward()
        if args.clip > 0:
            torch.nn.utils.clip_grad_norm_(params, args.clip)
        opt.step()
        sched.step()
        if step % (10 if args.device == "cpu" else 100) == 0:
            print("step %4d  loss trans %.4f rot %.4f   %.0f s" % (step, float(loss_t.detach()), float(loss_r.detach()), time.time() - t0), flush=True)
        if step % args.eval_every == args.eval_every - 1 or step == args.steps - 1:
            val = evaluate()
            print("   val |d trans| %.4f |d rot| %.4f  residual ratio trans %.3f rot %.3f" % val, flush=True)
            save(args.out, val)
    print("saved", args.out, os.path.getsize(args.out), "bytes", flush=True)
    if regen_pool is not None:
        regen_thread.shutdown(wait=False, cancel_futures=True)
        regen_pool.shutdown(wait=False, cancel_futures=True)
        os._exit(0)                                       # (the generation of a chunk nobody will use is still running)


if __name__ == "__main__":
    main()
