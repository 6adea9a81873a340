#!/usr/bin/env python3
"""bench.py -- throughput of the se(3)-TrackNet hot path on MI355X.

Metric (BASELINE.json): RGB-D pair inferences / s at 176x176.  A "step" is one pass of the hot path
over one batch of synthetic pairs already resident in HBM:
    se3tn_preprocess (crop + nearest-resize + depth offset + normalise, A and B)
 -> se3tn_infer      (two-branch CNN, 5.527 GFLOP / pair, exact-f32 MFMA)  + pose update
 -> (N>1) all-gather of the poses over RCCL.
N=1 workload = BASELINE configs[1]: batch 64 pairs, random-init weights on the reference's
state_dict surface (pretrained 003_cracker_box weights are not available offline).
N>1: weak scaling, 64 pairs per GPU, weights broadcast once from rank 0 over RCCL.  `python bench.py --gpus N`
launches its own N ranks (torch.distributed.run, one per GPU) when it is not already running under a launcher.

The JSON line carries, besides the contract fields:
  roofline      executed MFMA flops of the 3x3 conv family / its HIP-event time / 157.3 TF (frac <= 1); the
                algorithmic (direct-convolution) rate is reported beside it as effective_vs_direct
  parity        the TIMED batch checked against the CPU oracle: all pairs, pre-processing bit-exact, (trans, rot)
                and the composed pose against the north-star tolerances
  track         300-frame closed-loop Tracker.on_track stand-in for configs[2]: `per_step_parity` (the oracle evaluated at the HIP
                track's pose every frame) and `free_running` (two independent closed loops, HIP / oracle, 3 seeds: 1000 frames on a
                synthetic tracking problem with ground truth and trained stand-in weights, under both normaliser regimes -- ADD / ADD-S
                AUC of both tracks -- and 300 on the random-init stand-in, each with the oracle-vs-itself control)
  cpu_baseline  the oracle timed on this box's host cores (bounded sample)

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64] [--stage full|net]
"""
import argparse
import contextlib
import json
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_PAIR = 5527115776            # BASELINE.md section 3 (17 convs + 2 FC)
CONV3_FLOP_PER_PAIR = 2 * (2763557888 - 194281472 - 3072)  # the ten 3x3 conv launches (no stem/FC)
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MIN_TIMED_SECONDS = 1.0               # a timed region shorter than this is re-run with more steps
NET_TOL, POSE_TOL = 1e-4, 1e-5        # north-star tolerances (BASELINE.json)


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: become `torch.distributed.run --nproc-per-node N bench.py ...`."""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not args.dry_run_gloo:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this box -- N=%d is UNMEASURED here "
                         "(no extrapolation)" % (args.gpus, ndev, args.gpus))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU per step")
    ap.add_argument("--stage", default="full", choices=["full", "net"],
                    help="full = preprocess + network + pose update (default); net = network + pose on "
                         "pre-normalised NHWC pairs")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3"],
                    help="f32 = exact float32 MFMA (default, the parity configuration); f16x3 = split-f16 MFMA for the "
                         "256/512-channel layers (float32-class error, see DESIGN.md)")
    ap.add_argument("--winograd", type=int, default=None, metavar="MIN_BATCH",
                    help="se3tn_set_winograd threshold (0 = direct kernels only; default: the library's)")
    ap.add_argument("--winograd-tile", type=int, default=0, choices=[0, 2, 4, 6, 46],
                    help="F(tile x tile,3x3); 46 = SE3TN_WINOGRAD_TILE_AUTO (4 below 14 pairs, 6 from there); 0 = library default")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="N > 1: all-gather the poses after every step (overlapped with the next step) instead of once "
                         "at the end of the timed region")
    ap.add_argument("--host-frames", action="store_true",
                    help="PCIe-inclusive variant (never the headline): every step's camera frames start in pinned host "
                         "memory and are uploaded on a copy stream, double-buffered against the previous step's compute")
    ap.add_argument("--streams", type=int, default=2,
                    help="lanes of the throughput mode (se3.PipelinedEngine): successive steps alternate over this many HIP "
                         "streams / activation workspaces, so the HBM-bound passes of one step run under the MFMA-bound kernels "
                         "of the other; 1 = strictly one step at a time (reported as `single_stream` in any case)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tracker-batch", action="store_true", help="skip the Tracker.on_track_batch leg (tracker_batch block)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed batch")
    ap.add_argument("--track-frames", type=int, default=300, help="closed-loop Tracker.on_track stand-in (0 = skip)")
    ap.add_argument("--free-frames", type=int, default=1000,
                    help="frames of the FREE-RUNNING two-track comparison on the synthetic tracking problem with ground truth and "
                         "trained stand-in weights (track.free_running.synthetic_tracking_trained_weights; 0 = skip)")
    ap.add_argument("--free-frames-random", type=int, default=300,
                    help="frames of the free-running comparison on the random-init stand-in (track.free_running.random_init; 0 = skip).  300 by "
                         "default: those tracks are unrelated after 5-20 frames whatever the implementation (1,000 frames x 3 seeds x every "
                         "control: profiles/r06_free_run.json)")
    ap.add_argument("--exact-steps", action="store_true",
                    help="never extend the timed region beyond --steps (default: a region shorter than 1 s is re-run "
                         "with enough steps and BOTH are reported)")
    ap.add_argument("--layers", action="store_true", help="print the per-launch time breakdown to stderr")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="NOT a measurement: the same self-launch, rank logic, barriers, weight-blob broadcast (the real packed "
                         "blob), pose all-gather and single JSON line on CPU with the gloo backend and every kernel stubbed out "
                         "(tests/test_bench_launch.py: the N > 1 code path must have run somewhere before it meets RCCL)")
    args = ap.parse_args()

    # the host driver of these boxes supports dmabuf IPC only: without this RCCL's buffer sharing between the ranks fails with
    # hipIpcGetMemHandle "invalid argument".  Exported by the image already; set here too, before the HIP runtime loads.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "RANK" not in os.environ and args.gpus > 1:
            self_launch(args)          # does not return
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    import se3tracknet_amd as se3
    from oracle import se3_oracle as O  # weights generator, and the CHECKER of the parity / cpu_baseline / track blocks

    DRY = args.dry_run_gloo
    if DRY:
        dev = "cpu"
        sync = lambda: None                                            # noqa: E731
        stream_ctx = lambda stream: contextlib.nullcontext()           # noqa: E731
        args.no_parity = args.no_cpu_baseline = True
        args.track_frames = 0
        os.environ["SE3TN_NO_ALT"] = "1"
    else:
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("rank %d: no GPU %d on this box (device_count = %d)" % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = "cuda:%d" % local_rank
        sync = torch.cuda.synchronize
        stream_ctx = torch.cuda.stream
    # SE3TN_FORCE_DIST=1: run the RCCL code path (weight broadcast, pose all-gather) even at world 1
    use_dist = world > 1 or (os.environ.get("SE3TN_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints its version banner on STDOUT when the communicator is created; stdout carries exactly one JSON
        # line, so file descriptor 1 points at stderr until the communicator exists (first collective below)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if DRY:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device(dev))
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)
            sync()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    dist_mod = __import__("importlib").import_module("iros20-6d-pose-tracking_amd.dist")

    nb = args.batch
    lanes = 1 if (args.host_frames or args.gather_every_step) else max(1, args.streams)
    # lane 0 is also the single-stream engine of the legs below
    pe = DryPipelinedEngine(se3, rank, nb, lanes) if DRY else se3.PipelinedEngine(local_rank, nb, depth=lanes)
    eng = pe.engines[0]
    sd = O.make_state_dict(0) if rank == 0 else None
    if use_dist:
        blob = dist_mod.load_weights_everywhere(eng, sd, device=dev)   # C1: RCCL broadcast of the packed blob
        pe.bind_blob(blob)                                     # every lane reads the same device copy
    else:
        pe.load_state_dict(sd)
    mean = np.array([110., 105., 100., 1000., 112., 104., 99., 1010.]); std = np.array([60., 58., 61., 300., 59., 60., 62., 310.])
    pe.set_normalization(mean, std)
    TN, RN = 0.03, 5 * np.pi / 180
    pe.set_normalizers(TN, RN)
    if args.precision == "f16x3":
        pe.set_precision(se3._lib.PREC_F16X3)
    if args.winograd is not None or args.winograd_tile:
        pe.set_winograd(args.winograd if args.winograd is not None else 8, args.winograd_tile)

    # ---- synthetic inputs, resident in HBM (seeded per rank) -------------------------------
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    H, W = 480, 640
    frames_rgb = torch.randint(0, 256, (nb, H, W, 3), generator=g, device=dev, dtype=torch.uint8)
    frames_d = torch.randint(300, 1500, (nb, H, W), generator=g, device=dev, dtype=torch.int16)
    rend_rgb = torch.randint(0, 256, (nb, 176, 176, 3), generator=g, device=dev, dtype=torch.uint8)
    rend_d = torch.randint(600, 1000, (nb, 176, 176), generator=g, device=dev, dtype=torch.int16)
    rng = np.random.default_rng(7 + rank)
    poses = np.tile(np.eye(4), (nb, 1, 1))
    poses[:, 0, 3] = rng.uniform(-0.15, 0.15, nb); poses[:, 1, 3] = rng.uniform(-0.1, 0.1, nb)
    poses[:, 2, 3] = rng.uniform(0.6, 1.0, nb)
    K = np.array([[1066.778, 0, 312.9869], [0, 1067.487, 241.3109], [0, 0, 1]])
    bboxes = [se3.compute_bbox(poses[i], K, 250.0) for i in range(nb)]
    windowsA = np.tile(np.array([0, 0, 176, 176]), (nb, 1))
    windowsB = np.array([se3.crop_window(bb) for bb in bboxes])
    z_mm = poses[:, 2, 3] * 1000.0

    # --host-frames: two device frame sets, filled alternately from pinned host memory by a copy stream
    frame_sets = [(frames_rgb, frames_d)]
    if args.host_frames:
        frame_sets.append((torch.empty_like(frames_rgb), torch.empty_like(frames_d)))
        host_rgb, host_d = frames_rgb.cpu().pin_memory(), frames_d.cpu().pin_memory()
        copy_stream = torch.cuda.Stream(device=dev)
        uploaded = [torch.cuda.Event(), torch.cuda.Event()]   # set k's upload finished
        consumed = [torch.cuda.Event(), torch.cuda.Event()]   # set k's last reader (preprocess) finished
        for e in consumed:
            e.record()
    step_no = [0]

    def upload(k):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[k])
            frame_sets[k][0].copy_(host_rgb, non_blocking=True)
            frame_sets[k][1].copy_(host_d, non_blocking=True)
            uploaded[k].record(copy_stream)

    def make_crops(k=0):   # per step, like a tracker would: 2 x nb descriptors, vectorised (no per-crop Python)
        if DRY:
            return None, None
        return (se3.pack_crops(rend_rgb, rend_d, windowsA, z_mm, 0),
                se3.pack_crops(frame_sets[k][0], frame_sets[k][1], windowsB, z_mm, 1))
    cropsA, cropsB = make_crops()
    poseA = torch.from_numpy(poses.reshape(nb, 16)).to(dev)
    poseB = torch.empty_like(poseA)
    poseB_alt = torch.empty_like(poseA)   # N > 1: poses of step k are all-gathered while step k+1 computes
    trans = torch.empty((nb, 3), device=dev); rot = torch.empty((nb, 3), device=dev)
    inA, inB = eng.input_buffer_ptr(0), eng.input_buffer_ptr(1)
    if args.stage == "net":
        eng.preprocess(cropsA, inA); eng.preprocess(cropsB, inB)

    pending = []   # [(gathered poses, work)] of the previous step

    def step():
        if args.stage == "full":
            k = 0
            if args.host_frames:
                k = step_no[0] & 1
                if step_no[0] == 0:
                    upload(0)
                upload(k ^ 1)                                   # next step's frames, under this step's compute
                torch.cuda.current_stream().wait_event(uploaded[k])
                step_no[0] += 1
            cA, cB = make_crops(k)
            eng.preprocess(cA, inA)
            eng.preprocess(cB, inB)
            if args.host_frames:
                consumed[k].record()
        if not (use_dist and args.gather_every_step):
            eng.infer(inA, inB, nb, se3.NHWC, trans, rot, poseA, poseB)
            return poseB
        # C2, overlapped: step k writes pose buffer k % 2; its all-gather runs on the RCCL stream while
        # step k+1 computes into the other buffer; the gather of step k-1 is retired first
        buf = poseB if len(pending) == 0 or pending[-1][2] is poseB_alt else poseB_alt
        eng.infer(inA, inB, nb, se3.NHWC, trans, rot, poseA, buf)
        while pending:
            pending.pop(0)[1].wait()
        out, work = dist_mod.gather_poses_async(buf)
        pending.append((out, work, buf))
        return out

    def drain():
        # the path shards by pairs and has no exchange step: by default the ranks' poses are
        # collected ONCE, here, inside the timed region
        if use_dist and not args.gather_every_step:
            return dist_mod.gather_poses(poseB)      # C2
        while pending:
            pending.pop(0)[1].wait()

    local_dt = [0.0]
    min_timed = 0.02 if DRY else MIN_TIMED_SECONDS

    def timed_loop(steps):
        sync()
        if use_dist:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        drain()
        sync()
        local_dt[0] = time.perf_counter() - t0      # this rank alone (before the closing barrier)
        if use_dist:
            dist.barrier()
        sync()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed_region(steps):
        """Warm-up is done by the caller.  Times exactly `steps` steps; if that region is shorter than
        MIN_TIMED_SECONDS it is timed AGAIN with enough steps to exceed it (every rank takes the same decision from
        the max-over-ranks time).  Returns (dt, steps_timed, short) with short = (dt, steps) of the first region or None."""
        slots = min(steps, 64)
        eng.profile_enable(slots)
        dt = timed_loop(steps)
        if args.exact_steps or dt >= min_timed:
            return dt, steps, None, slots
        more = min(20000, int(math.ceil(steps * min_timed * 1.1 / dt)))
        slots = min(more, 64)
        eng.profile_enable(slots)
        dt2 = timed_loop(more)
        return dt2, more, (dt, steps), slots

    # ---- throughput mode: successive steps alternate over the lanes of the PipelinedEngine --------------------------
    lane_out = [(torch.empty((nb, 3), device=dev), torch.empty((nb, 3), device=dev), torch.empty_like(poseA)) for _ in range(lanes)]
    last_lane = [0]

    def step_pipelined():
        e, stream = pe.next_lane()
        k = pe.engines.index(e)
        last_lane[0] = k
        with stream_ctx(stream):
            cA, cB = make_crops(0)
            e.preprocess(cA, e.input_buffer_ptr(0))
            e.preprocess(cB, e.input_buffer_ptr(1))
            e.infer(e.input_buffer_ptr(0), e.input_buffer_ptr(1), nb, se3.NHWC, lane_out[k][0], lane_out[k][1], poseA, lane_out[k][2])

    def verify_pipelined(steps, want):
        """Untimed: `steps` pipelined steps queued back to back, the outputs of EVERY step kept (cloned on its own
        stream) and compared with the single-stream results `want` = (trans, rot, pose).  Returns the number of
        steps whose outputs are not bit-identical."""
        sync()
        kept = []
        for _ in range(steps):
            step_pipelined()
            k = last_lane[0]
            with stream_ctx(pe.streams[k]):
                kept.append(tuple(x.clone() for x in lane_out[k]))
        sync()
        return sum(0 if all(torch.equal(a, b) for a, b in zip(o, want)) else 1 for o in kept)

    def timed_loop_pipelined(steps):
        sync()
        if use_dist:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_pipelined()
        for st_ in pe.streams:                                 # the collective (and the clock) wait for every lane
            if not DRY:
                torch.cuda.current_stream().wait_stream(st_)
        if use_dist:
            dist_mod.gather_poses(lane_out[last_lane[0]][2])   # C2, once per timed region as in the single-stream loop
        sync()
        local_dt[0] = time.perf_counter() - t0
        if use_dist:
            dist.barrier()
        sync()
        dtp = time.perf_counter() - t0
        if use_dist:
            t_ = torch.tensor([dtp], dtype=torch.float64, device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            dtp = float(t_.item())
        return dtp

    for _ in range(args.warmup):
        step()
    dt, steps_timed, short, slots = timed_region(args.steps)
    per_rank = [nb * steps_timed / local_dt[0]]
    if use_dist:
        t = torch.tensor([per_rank[0]], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allr, t)
        per_rank = [float(x.item()) for x in allr]

    dry_checks = None
    if DRY and use_dist:
        # the gathered poses carry every rank's marker (DryEngine.infer writes rank + 1 into element 15), the blob every rank
        # holds is byte-identical to rank 0's (checksum), and its size is what the library packs
        allp = dist_mod.gather_poses(poseB)
        marks = [float(allp[r * nb, 15]) for r in range(world)]
        csum = torch.tensor([int(blob.to(torch.int64).sum())])
        sums = [torch.zeros_like(csum) for _ in range(world)]
        dist.all_gather(sums, csum)
        dry_checks = {"gathered_rank_markers": marks, "gather_ok": marks == [float(r + 1) for r in range(world)],
                      "blob_bytes": int(blob.numel()), "blob_identical_on_all_ranks": len({int(x) for x in sums}) == 1}
        assert dry_checks["gather_ok"] and dry_checks["blob_identical_on_all_ranks"], dry_checks

    # ---- roofline of the dominant kernel family, from the HIP events of the timed region ----
    conv_ms, tot_ms, nconv = [], [], 0
    for s in range(slots):
        c, nconv, t = eng.profile_read(s)
        conv_ms.append(c); tot_ms.append(t)
    conv_ms_avg = float(np.mean(conv_ms))
    algorithmic = CONV3_FLOP_PER_PAIR * nb / (conv_ms_avg * 1e-3) / 1e12
    # MFMA flops the conv family actually executes: the Winograd layers do (tile+2)^2 multiplies per
    # tile x tile outputs (incl. the rows / columns computed past the map edge) instead of 9 per output
    wino_min, wino_tile_sel = eng.get_winograd()
    layers = eng.profile_launches(slots - 1)

    def tile_of(name):
        """Winograd tile of a conv launch, from the algorithm tag the library puts into its profile names ("[F(6x6)]"); 0 = direct."""
        import re
        m_ = re.search(r"\[F\((\d)x\1\)\]", name)
        return int(m_.group(1)) if m_ else 0
    WINO_LAYERS = {"convAB2.conv1": (22, 256, 256, 1), "convAB2.conv2": (22, 256, 256, 1),      # name prefix -> (hw, cin, cout, groups)
                   "trans|rot conv2.conv1": (11, 512, 512, 2), "trans|rot conv2.conv2": (11, 512, 512, 2)}
    executed_per_pair = CONV3_FLOP_PER_PAIR
    wino_tiles = {}
    for name, _ in layers:
        key = next((k for k in WINO_LAYERS if name.startswith(k)), None)
        t_ = tile_of(name)
        if key is None or not t_:
            continue
        hw, cin, cout, groups = WINO_LAYERS[key]
        wino_tiles[key] = t_
        executed_per_pair += groups * 2 * cin * cout * ((t_ + 2) ** 2 * (-(-hw // t_)) ** 2 - 9 * hw * hw)
    wino_on = bool(wino_tiles)
    wino_desc = "/".join("F(%dx%d)" % (t_, t_) for t_ in sorted(set(wino_tiles.values()), reverse=True))
    layers = eng.profile_launches(slots - 1)
    # the 64-channel trunk launches that took the fused Winograd F(2x2) kernel (the library decides per launch: whole rounds of
    # workgroups only; the launch name says so): 4 quadrants x 32 steps x (128 x 64 x 32) MACs per image and group instead of
    # 44 x 44 x 64 x 64 x 9
    TRUNK_FUSED_FLOP = 2 * 4 * 32 * 128 * 64 * 32
    TRUNK_DIRECT_FLOP = 2 * 64 * 64 * 9 * 44 * 44
    trunk_fused = [name for name, _ in layers if name.startswith("conv64") and "fused F(2x2)" in name]
    for name in trunk_fused:
        executed_per_pair += (2 if "|" in name else 1) * (TRUNK_FUSED_FLOP - TRUNK_DIRECT_FLOP)
    executed = executed_per_pair * nb / (conv_ms_avg * 1e-3) / 1e12
    # per-launch view: executed flops of every conv launch over its own HIP-event time (averaged over the recorded steps)
    per_slot = [dict(eng.profile_launches(s_)) for s_ in range(slots)]
    spec = {   # name prefix -> (cin, cout, groups, out_hw, winograd-capable)
        "conv64 A2.conv1": (64, 64, 2, 44, False), "conv64 A2.conv2": (64, 64, 2, 44, False),
        "conv64 B3.conv1": (64, 64, 1, 44, False), "conv64 B3.conv2": (64, 64, 1, 44, False),
        "convAB1": (128, 256, 1, 22, False), "convAB2.conv1": (256, 256, 1, 22, True), "convAB2.conv2": (256, 256, 1, 22, True),
        "trans|rot conv1": (256, 1024, 1, 11, False), "trans|rot conv2.conv1": (512, 512, 2, 11, True),
        "trans|rot conv2.conv2": (512, 512, 2, 11, True)}
    per_layer = []
    for name, _ in layers:
        key = next((k for k in spec if name.startswith(k)), None)
        if key is None:
            continue
        cin, cout, groups, hw, wcap = spec[key]
        ms = float(np.mean([d_[name] for d_ in per_slot if name in d_]))
        if name in trunk_fused:
            fl = float(TRUNK_FUSED_FLOP) * groups * nb
        elif wcap and tile_of(name):
            fl = 2.0 * cin * cout * groups * (tile_of(name) + 2) ** 2 * (-(-hw // tile_of(name))) ** 2 * nb
        else:
            fl = 2.0 * cin * cout * groups * 9 * hw * hw * nb
        per_layer.append({"launch": name, "ms": round(ms, 4), "gflop_executed": round(fl / 1e9, 2),
                          "tflops": round(fl / (ms * 1e-3) / 1e12, 1), "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)})
    dominant = max(per_layer, key=lambda d_: d_["ms"]) if per_layer else None
    eng.profile_enable(0)
    sync()
    pose_main = poseB.clone()
    trans_main, rot_main = trans.clone(), rot.clone()

    # ---- parity of the TIMED batch against the CPU oracle (rank 0; the oracle is the checker here) ----
    parity = None
    oracle_inputs = None
    if rank == 0 and not args.no_parity and args.precision == "f32":
        parity, oracle_inputs = check_timed_batch(np, torch, se3, O, eng, sd, nb, frames_rgb, frames_d, rend_rgb, rend_d,
                                                  poses, bboxes, mean, std, TN, RN, trans_main, rot_main, pose_main,
                                                  check_pre=(args.stage == "full"))

    # ---- the headline when lanes > 1: the same steps pipelined over the lanes (each step is still one batch of nb pairs) ----
    pipelined = None
    if lanes > 1 and args.stage == "full":
        for _ in range(max(args.warmup, 2 * lanes)):
            step_pipelined()
        steps_p = steps_timed if args.exact_steps else int(math.ceil(steps_timed * 1.25))   # also >= 1 s at the higher rate
        dtp = timed_loop_pipelined(steps_p)
        pr = [nb * steps_p / local_dt[0]]
        if use_dist:
            t = torch.tensor([pr[0]], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allr, t)
            pr = [float(x.item()) for x in allr]
        same = all(torch.equal(o[0], trans_main) and torch.equal(o[1], rot_main) and torch.equal(o[2], pose_main) for o in lane_out)
        n_verify = 16 * lanes
        n_bad = verify_pipelined(n_verify, (trans_main, rot_main, pose_main))
        same = same and n_bad == 0
        pipelined = {"lanes": lanes, "value": round(world * nb * steps_p / dtp, 1), "ms_per_step": round(dtp / steps_p * 1e3, 4),
                     "steps_timed": steps_p, "timed_seconds": round(dtp, 4), "per_rank_pairs_per_s": [round(v, 1) for v in pr],
                     "outputs_bit_identical_to_single_stream": bool(same),
                     "verified_steps": n_verify, "verified_steps_differing": n_bad}
        assert os.environ.get("SE3TN_NOCHECK") or same, "a lane's outputs differ from the single-stream run"

    # second arithmetic mode on the same inputs: timed the same way, reported beside the main value
    other = None
    if args.precision == "f32" and nb >= 32 and args.stage == "full" and not os.environ.get("SE3TN_NO_ALT"):
        eng.set_precision(se3._lib.PREC_F16X3)
        for _ in range(args.warmup):
            step()
        sl = min(steps_timed, 64)
        eng.profile_enable(sl)
        dt2 = timed_loop(steps_timed)
        c2 = float(np.mean([eng.profile_read(s_)[0] for s_ in range(sl)]))
        eng.profile_enable(0)
        other = {"precision": "f16x3", "what": "every 3x3 conv as hi*hi+hi*lo+lo*hi on v_mfma_f32_32x32x16_f16 with split-row "
                 "(f16 hi|lo) operands, f32 accumulate; the 512-channel head block as fused Winograd F(4x4) passes on the same "
                 "split operands (the 256-channel block stays direct: its batched GEMMs are bandwidth-bound at the f16 rate)",
                 "value": round(world * nb * steps_timed / dt2, 1), "unit": "pairs/s", "ms_per_step": round(dt2 / steps_timed * 1e3, 4),
                 "steps_timed": steps_timed, "conv_ms_per_step": round(c2, 4),
                 "conv_tflops_f32_equivalent": round(CONV3_FLOP_PER_PAIR * nb / (c2 * 1e-3) / 1e12, 1),
                 "mfma_frac_of_2.5PF": round(3 * CONV3_FLOP_PER_PAIR * nb / (c2 * 1e-3) / 2.5e15, 4),
                 "max_abs_pose_diff_vs_f32": float((poseB - pose_main).abs().max()),
                 "range_guard_fired": bool(eng.overflow())}
        if parity is not None:   # the same oracle outputs, this mode's device results
            other["parity"] = compare_with_oracle(np, parity["_oracle"], trans.cpu().numpy(), rot.cpu().numpy(),
                                                  poseB.cpu().numpy().reshape(nb, 4, 4))
        if lanes > 1:            # and pipelined over the lanes like the headline
            t16, r16, p16 = trans.clone(), rot.clone(), poseB.clone()
            pe.set_precision(se3._lib.PREC_F16X3)
            for _ in range(max(args.warmup, 2 * lanes)):
                step_pipelined()
            dt2p = timed_loop_pipelined(steps_timed)
            other["pipelined_value"] = round(world * nb * steps_timed / dt2p, 1)
            other["pipelined_ms_per_step"] = round(dt2p / steps_timed * 1e3, 4)
            n_bad16 = verify_pipelined(16 * lanes, (t16, r16, p16))
            other["pipelined_outputs_bit_identical"] = bool(n_bad16 == 0 and all(
                torch.equal(o[0], t16) and torch.equal(o[1], r16) and torch.equal(o[2], p16) for o in lane_out))
            other["pipelined_verified_steps"] = 16 * lanes
            other["pipelined_verified_steps_differing"] = n_bad16
            other["range_guard_fired"] = bool(other["range_guard_fired"] or any(e_.overflow() for e_ in pe.engines))
            pe.set_precision(se3._lib.PREC_F32)
        eng.set_precision(se3._lib.PREC_F32)
    # and the same float32 run with the Winograd layers switched back to the direct kernels
    direct = None
    if (wino_on or trunk_fused) and not os.environ.get("SE3TN_NO_ALT"):
        trunk_min, trunk_fill = eng.get_trunk_winograd()
        eng.set_winograd(0)
        eng.set_trunk_winograd(0)
        for _ in range(args.warmup):
            step()
        sl = min(steps_timed, 64)
        eng.profile_enable(sl)
        dt3 = timed_loop(steps_timed)
        c3 = float(np.mean([eng.profile_read(s_)[0] for s_ in range(sl)]))
        eng.profile_enable(0)
        direct = {"algorithm": "direct implicit GEMM for all ten 3x3 convs (se3tn_set_winograd(ctx, 0, 0), se3tn_set_trunk_winograd(ctx, 0))",
                  "value": round(world * nb * steps_timed / dt3, 1), "unit": "pairs/s", "ms_per_step": round(dt3 / steps_timed * 1e3, 4),
                  "steps_timed": steps_timed, "conv_ms_per_step": round(c3, 4),
                  "achieved": round(CONV3_FLOP_PER_PAIR * nb / (c3 * 1e-3) / 1e12, 2),
                  "frac": round(CONV3_FLOP_PER_PAIR * nb / (c3 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                  "max_abs_pose_diff_vs_main": float((poseB - pose_main).abs().max())}
        if parity is not None:
            direct["parity"] = compare_with_oracle(np, parity["_oracle"], trans.cpu().numpy(), rot.cpu().numpy(),
                                                   poseB.cpu().numpy().reshape(nb, 4, 4))
        eng.set_winograd(wino_min, wino_tile_sel)
        if trunk_fused and wino_on:
            # third point of the same line: the Winograd blocks as in the headline, the 64-channel trunk on the direct kernels (the
            # library before the fused trunk kernel): executed flops go UP by the trunk's 2.13 x and so does `frac`, the step gets slower
            for _ in range(args.warmup):
                step()
            eng.profile_enable(sl)
            dt4 = timed_loop(steps_timed)
            c4 = float(np.mean([eng.profile_read(s_)[0] for s_ in range(sl)]))
            eng.profile_enable(0)
            ex4 = executed_per_pair - sum((2 if "|" in n_ else 1) * (TRUNK_FUSED_FLOP - TRUNK_DIRECT_FLOP) for n_ in trunk_fused)
            direct["trunk_direct"] = {"algorithm": "Winograd %s blocks as in the headline, 64-channel trunk on the direct kernels "
                                                   "(se3tn_set_trunk_winograd(ctx, 0))" % wino_desc,
                                      "value": round(world * nb * steps_timed / dt4, 1), "unit": "pairs/s",
                                      "ms_per_step": round(dt4 / steps_timed * 1e3, 4), "conv_ms_per_step": round(c4, 4),
                                      "flop_per_step_executed": ex4 * nb,
                                      "achieved": round(ex4 * nb / (c4 * 1e-3) / 1e12, 2),
                                      "frac": round(ex4 * nb / (c4 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
        eng.set_trunk_winograd(trunk_min, trunk_fill)
    assert os.environ.get("SE3TN_NOCHECK") or torch.isfinite(poseB).all()
    assert os.environ.get("SE3TN_NOCHECK") or not eng.overflow(), "f16x3 range guard fired"

    if rank == 0:
        value_single = world * nb * steps_timed / dt
        value = pipelined["value"] if pipelined else value_single
        out = {
            "metric": "RGB-D pair inferences/sec (176x176)", "value": round(value, 1), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": pipelined["ms_per_step"] if pipelined else round(dt / steps_timed * 1e3, 4),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "steps_timed": pipelined["steps_timed"] if pipelined else steps_timed,
            "timed_seconds": pipelined["timed_seconds"] if pipelined else round(dt, 4),
            # `value`: K complete steps of one batch each, successive steps alternating over `lanes` HIP streams / activation
            # workspaces (PipelinedEngine) -- throughput of the job, not 1 / latency of a step.  `single_stream`: the same K
            # steps strictly one after the other on one stream; the roofline / layers_ms blocks are measured in THAT region
            # (kernel durations undisturbed by a concurrent lane).
            "steps_in_flight": lanes if pipelined else 1,
            "single_stream": {"value": round(value_single, 1), "ms_per_step": round(dt / steps_timed * 1e3, 4),
                              "timed_seconds": round(dt, 4), "steps_timed": steps_timed},
            "config": {"workload": "configs[1]: batch=%d synthetic 176x176 RGB-D pairs per GPU, random-init Se3TrackNet "
                                   "(reference state_dict surface), stage=%s" % (nb, args.stage),
                       "pairs_per_gpu": nb, "global_batch": world * nb, "stage": args.stage,
                       "frames": "pinned host memory, uploaded per step (PCIe-inclusive)" if args.host_frames else "resident in HBM",
                       "parallelism": "frame-sharded x%d, RCCL weight broadcast at start-up, pose all-gather %s; per GPU the steps "
                                      "alternate over %d HIP stream(s)" %
                                      (world, "every step (overlapped)" if args.gather_every_step else "once per timed region", lanes)},
            "rccl_ranks": dist.get_world_size() if use_dist else 1,
            "backend": dist.get_backend() if use_dist else None,
            "per_rank_pairs_per_s": pipelined["per_rank_pairs_per_s"] if pipelined else [round(v, 1) for v in per_rank],
            "tflops_total": round(value * FLOP_PER_PAIR / 1e12, 2),
            "roofline": {"bound": "mfma",
                         "kernel": "3x3 conv family, 10 convs/step on exact-f32 v_mfma_f32_32x32x2_f32: direct implicit GEMM "
                                   "(conv3x3_slab_kernel, conv3x3_gather_s2_kernel)" +
                                   (" + Winograd %s (3x3) fused residual blocks for AB2.* and trans|rot conv2.* (wino_input / gemm / mid / "
                                    "output | tail kernels)" % wino_desc if wino_on else "") +
                                   (" + fused Winograd F(2x2,3x3) for the 64-channel trunk (wino64_fused_kernel: %d of its 4 launches)"
                                    % len(trunk_fused) if trunk_fused else ""),
                         # achieved / frac = the MFMA flops the family EXECUTES (Winograd layers: (tile+2)^2 multiplies per
                         # tile x tile outputs) / its HIP-event time (transform passes included) / the f32-MFMA peak
                         "achieved": round(executed, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(executed / PEAK_F32_MFMA_TFLOPS, 4),
                         "flop_per_step_executed": executed_per_pair * nb,
                         # the ALGORITHMIC (direct-convolution) rate: what the Winograd layers buy
                         "achieved_algorithmic": round(algorithmic, 2), "flop_per_step_algorithmic": CONV3_FLOP_PER_PAIR * nb,
                         "effective_vs_direct": round(algorithmic / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic": None,
                         "conv_ms_per_step": round(conv_ms_avg, 4), "all_kernels_ms_per_step": round(float(np.mean(tot_ms)), 4),
                         "timing": "hipEvents around every launch on the launch stream inside the timed region "
                                   "(last %d steps)" % slots,
                         # the single longest launch of the family and every launch's own rate (Winograd entries include their
                         # transform passes in the time)
                         "dominant_launch": dominant, "launches": per_layer,
                         # whole-step view: every MFMA flop a step executes (the ten 3x3 convs as run + the two 7x7 stems,
                         # 194,281,472 MAC per pair) over the WALL time of a step, single-stream and pipelined
                         "whole_step": {"flop_per_step_executed": executed_per_pair * nb + 2 * 194281472 * nb,
                                        "frac_single_stream": round((executed_per_pair + 2 * 194281472) * nb / (dt / steps_timed) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                        "frac_pipelined": (round((executed_per_pair + 2 * 194281472) * nb / (pipelined["ms_per_step"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
                                                           if pipelined else None)}},
            "layers_ms": {n: round(ms, 4) for n, ms in layers},
        }
        if DRY:
            out["dry_run"] = ("gloo on CPU, every kernel stubbed out: `value`, `roofline` and `layers_ms` are NOT measurements; this "
                              "line only shows that the N-rank launch, broadcast, gather and reporting path runs")
            out["dry_run_checks"] = dry_checks
        if pipelined:
            out["pipelined"] = pipelined
        if short is not None:
            out["requested_steps_region"] = {"steps": short[1], "seconds": round(short[0], 4),
                                             "ms_per_step": round(short[0] / short[1] * 1e3, 4),
                                             "value": round(world * nb * short[1] / short[0], 1),
                                             "note": "the --steps region (single stream) was shorter than %.1f s; `single_stream` and "
                                                     "`value` come from regions of steps_timed steps timed right after it" % MIN_TIMED_SECONDS}
        if parity is not None:
            parity.pop("_oracle", None)
            out["parity"] = parity
        if other is not None:
            out["alt_precision"] = other
        if direct is not None:
            out["alt_algorithm"] = direct
        prof = pmc_traffic(nb)
        # the committed PMC passes are of the DEFAULT float32 configuration: another precision / algorithm setting keeps traffic null
        #  and carries the profile only as `traffic_from_profile`
        default_cfg = (args.precision == "f32" and args.winograd is None and args.winograd_tile == 0 and args.stage == "full"
                       and not any(k.startswith(("SE3TN_WINO", "SE3TN_TRUNK", "SE3TN_SPLITK", "SE3TN_GEMM")) for k in os.environ))
        if prof is not None and not default_cfg:
            out["roofline"]["traffic_from_profile"] = dict(prof, note=prof["note"] + "; NOT this run's configuration")
        elif prof is not None:
            # PMC counters need rocprofv3 around the process, so the line quotes the newest COMMITTED profile of this command (per step
            # of the conv family, FETCH_SIZE x 2 + WRITE_SIZE) and says so; `traffic_from_profile` has the details
            out["roofline"]["traffic"] = prof["bytes_per_step"]
            out["roofline"]["traffic_unit"] = "bytes per step (conv family), HBM + Infinity Cache"
            out["roofline"]["traffic_source"] = prof["source"] + " (committed rocprofv3 PMC passes of this command; not measured by this run)"
            out["roofline"]["traffic_from_profile"] = prof
            # per-kernel matrix-core occupancy of the same committed PMC passes beside the per-launch table
            out["roofline"]["mfma_busy_by_kernel"] = prof.get("mfma_busy_by_kernel", {})
            out["roofline"]["mfma_busy_source"] = prof["source"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(O, sd, nb, oracle_inputs)
        if world == 1 and args.track_frames > 0 and args.precision == "f32":
            from oracle import closed_loop
            out["track"] = closed_loop.run(se3, frames=args.track_frames, check=not args.no_parity)
            if not args.no_parity and (args.free_frames > 0 or args.free_frames_random > 0):
                # two INDEPENDENT closed loops (HIP tracker / CPU oracle), 3 seeds each, + the oracle-vs-itself control: what
                # "same track" means (oracle/free_run.py); the oracle tracks run in worker processes on the host cores
                from oracle import free_run
                try:
                    out["track"]["free_running"] = free_run.run_report(se3, args.free_frames, args.free_frames_random)
                except Exception as e:   # noqa: BLE001  (a checker leg on host worker processes must not take the bench line down)
                    out["track"]["free_running"] = {"error": repr(e)}
        if world == 1 and args.track_frames > 0 and args.precision == "f32" and not args.no_tracker_batch:
            # the path a many-tracks / many-objects deployment (configs[3] / [4]) calls per step: Tracker.on_track_batch, renderer- and
            # PCIe-inclusive (frames start in host memory), with a pair-by-pair oracle check of the same configuration
            from oracle import closed_loop
            try:
                sizes = {str(n): closed_loop.time_batch(se3, n, check_frames=0 if args.no_parity else 2) for n in (8, 21, 64)}
            except Exception as e:   # noqa: BLE001
                sizes = {"error": repr(e)}
            out["tracker_batch"] = {
                "what": "Tracker.on_track_batch: n independent closed-loop tracks per call = host float64 bboxes + image A of all n poses "
                        "rendered on the device + the n camera frames' crop windows staged from host memory and uploaded + both crops + "
                        "network + pose update + read-back; wall clock per call, synthetic frames, random-init weights",
                "sizes": sizes}
        if args.layers:
            for n, ms in layers:
                print("%-32s %8.3f ms" % (n, ms), file=sys.stderr)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


class DryEngine:
    """--dry-run-gloo only: stands where se3.Engine stands.  The weight blob is the REAL one (the library's host packer through a
    device -1 context); every device call is a no-op, `infer` writes poseA with this rank's marker into poseB."""

    def __init__(self, se3, rank, nb):
        self.host = se3.Engine(device=-1, max_batch=1)
        self.rank, self.max_batch, self.device, self.blob = rank, nb, -1, None
        self._slots = 0

    def pack_state_dict(self, sd):
        return self.host.pack_state_dict(sd)

    def packed_bytes(self):
        return self.host.packed_bytes()

    def bind_blob(self, blob):
        import numpy as np
        assert blob.numel() == self.packed_bytes() and bytes(np.asarray(blob[:4])) == b"T3ES", "bad blob header"
        self.blob = blob

    def input_buffer_ptr(self, which):
        return 0

    def preprocess(self, crops, out):
        pass

    def infer(self, A, B, n, layout, trans, rot, poseA, poseB):
        poseB.copy_(poseA)
        poseB[:, 15] = self.rank + 1
        trans.zero_(); rot.zero_()

    def profile_enable(self, slots=1):
        self._slots = slots

    def profile_read(self, slot=0):
        return 1.0, 10, 1.2

    def profile_launches(self, slot=0):
        return [("stem7x7_mfma", 0.2), ("convAB1 s2", 0.5), ("trans|rot conv1 s2", 0.5)]

    def get_winograd(self):
        return 8, 46

    def get_trunk_winograd(self):
        return 8, 55

    def overflow(self):
        return False

    def __getattr__(self, name):
        if name.startswith("set_"):
            return lambda *a, **k: None
        raise AttributeError(name)


class DryPipelinedEngine:
    def __init__(self, se3, rank, nb, lanes):
        self.engines = [DryEngine(se3, rank, nb) for _ in range(lanes)]
        self.streams = [None] * lanes
        self._i = 0

    def load_state_dict(self, sd):
        self.bind_blob(self.engines[0].pack_state_dict(sd))

    def bind_blob(self, blob):
        for e in self.engines:
            e.bind_blob(blob)

    def next_lane(self):
        k = self._i % len(self.engines)
        self._i += 1
        return self.engines[k], self.streams[k]

    def __getattr__(self, name):
        if name.startswith("set_"):
            return lambda *a, **k: None
        raise AttributeError(name)


def compare_with_oracle(np, orc, trans, rot, pose):
    e_net = max(float(np.abs(trans - orc["trans"]).max()), float(np.abs(rot - orc["rot"]).max()))
    e_pose = float(np.abs(pose - orc["pose"]).max())
    return {"pairs_checked": int(trans.shape[0]), "max_abs_trans_rot": e_net, "max_abs_pose": e_pose,
            "ok": bool(e_net <= NET_TOL and e_pose <= POSE_TOL)}


def check_timed_batch(np, torch, se3, O, eng, sd, nb, frames_rgb, frames_d, rend_rgb, rend_d, poses, bboxes, mean, std,
                      tn, rn, trans, rot, poseB, check_pre=True):
    """Every pair of the batch the timed region just ran, against the CPU oracle:
    (1) pre-processing: oracle crop_bbox + processData on the host copies of the frames vs the device's input
        tensors (bit-exact expected); (2) network + pose update: oracle forward on the device's own input tensors."""
    def interior(name):
        t = eng.debug_buffer(name, nb)[:, 3:-3, 3:-3, :]          # [nb,176,176,4] NHWC
        return t.permute(0, 3, 1, 2).contiguous().cpu()
    A, B = interior("inA"), interior("inB")
    pre_exact = None
    if check_pre:
        fr, fd = frames_rgb.cpu().numpy(), frames_d.cpu().numpy().view(np.uint16)
        rr, rd = rend_rgb.cpu().numpy(), rend_d.cpu().numpy().view(np.uint16)
        pre_exact = 0
        for i in range(nb):
            rgbB, depthB = O.crop_bbox(fr[i], fd[i], bboxes[i], (176, 176))
            a, b = O.process_data(rr[i], rd[i], poses[i], rgbB, depthB, mean, std)
            pre_exact += int(np.array_equal(a, A[i].numpy()) and np.array_equal(b, B[i].numpy()))
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    ref = O.forward(sd, A, B)
    want_pose = np.stack([O.process_predict(poses[i], ref["trans"][i].numpy(), ref["rot"][i].numpy(), tn, rn) for i in range(nb)])
    orc = {"trans": ref["trans"].numpy(), "rot": ref["rot"].numpy(), "pose": want_pose}
    res = compare_with_oracle(np, orc, trans.cpu().numpy(), rot.cpu().numpy(), poseB.cpu().numpy().reshape(nb, 4, 4))
    res.update(tol_trans_rot=NET_TOL, tol_pose=POSE_TOL,
               against="CPU oracle (torch-CPU fp32 + numpy restatement of the reference, pinned by tests/golden) on the timed batch")
    if pre_exact is not None:
        res["preprocess_bit_exact_pairs"] = pre_exact
        res["ok"] = bool(res["ok"] and pre_exact == nb)
    res["_oracle"] = orc
    return res, (A, B)


def pmc_traffic(nb):
    """HBM + Infinity-Cache bytes per step of the conv3x3 family from the newest COMMITTED rocprofv3 PMC summary that carries per-step
    totals (profiles/*_pmc.json `per_step`, scripts/summarize_profile.py: every dispatch of the FETCH_SIZE (x2 per MI355X_MICROARCH.md)
    and WRITE_SIZE passes of this same command at batch 64 in the default float32 configuration, summed and divided by the steps of
    the pass).  NOT measured by this run (PMC counters need rocprofv3 around the process): `roofline.traffic` quotes it with
    `traffic_source` naming the profile, and the profile's own bench value stands beside it."""
    import glob
    if nb != 64:
        return None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:   # noqa: BLE001
            continue
        ps = d.get("per_step")
        if not ps:
            continue
        tag = os.path.basename(f).replace("_pmc.json", "")
        bench_value = None
        bfile = os.path.join(ROOT, "profiles", tag + "_bench.json")
        if os.path.isfile(bfile):
            try:
                bench_value = json.load(open(bfile)).get("value")
            except Exception:   # noqa: BLE001
                bench_value = None
        # MFMA-busy per kernel from the SQ pass of the same profile: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
        busy = {}
        for kname, cnt in (d.get("sq") or {}).items():
            gui, mf = cnt.get("GRBM_GUI_ACTIVE", 0), cnt.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
            if gui and mf and kname in ps.get("launches_per_step", {}):
                busy[kname] = round(mf / (gui / 8 * 1024), 4)
        return {"bytes_per_step": int(ps["conv_family_bytes"]), "all_kernels_bytes_per_step": int(ps["all_kernels_bytes"]),
                "mfma_busy_by_kernel": busy,
                "source": "profiles/%s_pmc.json" % tag, "profile_bench_value": bench_value,
                "algorithmic_bytes_per_step": nb * 991256 + 54100000,
                "note": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) / WRITE_SIZE passes of a committed earlier run of this "
                        "command; includes Infinity-Cache hits"}
    return None


def cpu_model():
    model, sockets = "unknown", set()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                sockets.add(line.split(":", 1)[1].strip())
    except OSError:
        pass
    return model, max(1, len(sockets))


def physical_cores():
    """One logical CPU per physical core of this process's affinity mask, ordered by (package, core): SMT siblings are
    left idle (two oneDNN threads on one core run slower than one)."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, out = set(), []
    for c in allowed:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
            pkg = int(open("/sys/devices/system/cpu/cpu%d/topology/physical_package_id" % c).read())
        except OSError:
            sib, pkg = str(c), 0
        if (pkg, sib) in seen:
            continue
        seen.add((pkg, sib))
        out.append((pkg, c))
    return [c for _, c in sorted(out)]


def cpu_quota_cores():
    """CPU time this container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None if unlimited.  The GPU boxes of this
    pool expose 256 logical CPUs but run under a 16-core quota: more busy threads than that are CFS-throttled, which is what made
    the all-cores figures of rounds 1 / 2 collapse (256 threads: ~1 pair/s)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


CPU_SAMPLES = 3                      # samples per cpu_baseline configuration (median + spread are reported)


def _cpu_worker(cpus, threads, seconds, chunk, start, q):
    """One pinned oracle process of the cpu_baseline: the WHOLE per-pair path (crop_bbox + processData on numpy, the network
    on torch-CPU in forwards of `chunk` pairs, processPredict), for `seconds` after a common start."""
    try:
        os.sched_setaffinity(0, cpus)
        os.environ["OMP_NUM_THREADS"] = str(threads)
        import numpy as np
        import torch
        torch.set_num_threads(threads)
        from oracle import fixtures as Fx
        from oracle import se3_oracle as O
        sd = O.make_state_dict(0)
        mean, std = Fx.mean_std(0)
        rgb, depth = Fx.synthetic_frame(3); P = Fx.pose(3); rgbA, depthA = Fx.synthetic_render(103, 0.8)

        def one_chunk():
            a_, b_ = [], []
            for _ in range(chunk):
                bb = O.compute_bbox(P, Fx.K_YCB, 250.0, (1000, 1000, 1000))
                rgbB, depthB = O.crop_bbox(rgb, depth, bb, (176, 176))
                a, b = O.process_data(rgbA, depthA, P, rgbB, depthB, mean, std)
                a_.append(a); b_.append(b)
            out = O.forward(sd, torch.from_numpy(np.stack(a_)), torch.from_numpy(np.stack(b_)))
            for i in range(chunk):
                O.process_predict(P, out["trans"][i].numpy(), out["rot"][i].numpy())
        one_chunk()
        q.put(("ready", 0))
        if not start.wait(timeout=300):
            return
        wins = []
        for _ in range(CPU_SAMPLES):                     # consecutive windows of seconds / CPU_SAMPLES: one sample each
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < seconds / CPU_SAMPLES:
                one_chunk(); n += chunk
            wins.append((n, time.perf_counter() - t0))
        q.put(("done", wins))
    except Exception as e:   # noqa: BLE001
        q.put(("error", repr(e)))


def cpu_multiprocess_rate(procs, threads, seconds, chunk, cores):
    """`procs` pinned oracle processes x `threads` threads, disjoint blocks of physical cores, common start (barrier);
    returns CPU_SAMPLES whole-path rates [pairs/s], one per window = all pairs finished in it / the slowest worker's time."""
    import multiprocessing as mp
    import queue as queue_mod
    ctx = mp.get_context("spawn")
    start, q = ctx.Event(), ctx.Queue()
    ws = [ctx.Process(target=_cpu_worker, args=(set(cores[i * threads:(i + 1) * threads]), threads, seconds, chunk, start, q))
          for i in range(procs)]
    for w in ws:
        w.start()

    def collect(n, deadline):
        got = []
        while len(got) < n:
            try:
                got.append(q.get(timeout=1.0))
            except queue_mod.Empty:
                if time.time() > deadline or not all(w.is_alive() or w.exitcode == 0 for w in ws):
                    return got, "a worker died or timed out"
                continue
            if got[-1][0] == "error":
                return got, str(got[-1][1])
        return got, None
    _, err = collect(procs, time.time() + 240)          # every worker imported torch, pinned itself and warmed up
    res = []
    if err is None:
        start.set()                                     # common start
        res, err = collect(procs, time.time() + seconds + 120)
    for w in ws:
        if err is not None:
            w.kill()
        w.join(timeout=30)
    if err is not None:
        return None, err
    wins = [r[1] for r in res]
    return [sum(w[i][0] for w in wins) / max(w[i][1] for w in wins) for i in range(CPU_SAMPLES)], None


def _median_spread(samples):
    """median of the samples and (max - min) / median"""
    import numpy as np
    med = float(np.median(samples))
    return round(med, 2), [round(float(v), 2) for v in samples], round((max(samples) - min(samples)) / med, 4) if med > 0 else None


def reference_vs_port(O, sd, threads):
    """The UNMODIFIED reference Se3TrackNet on torch-CPU next to the oracle port, same weights / inputs / threads (forwards of 16 and
    of 1): only where the reference tree exists (the build container; never the GPU box)."""
    if not os.path.isdir("/root/reference"):
        return {"value": None, "reason": "/root/reference does not exist on this box (it never travels to the GPU box); measured on the "
                                         "build container: profiles/r05_cpu_reference_vs_port.txt (port / reference time 0.89 / 1.02 / 1.20 at "
                                         "batch 1 / 16 / 64, outputs bit-identical)"}
    try:
        import numpy as np
        import torch
        from oracle import fixtures as Fx, ref_shims
        from oracle.make_golden import ref_model
        model = ref_model(ref_shims.load(), sd)
        torch.set_num_threads(threads)
        out = {"threads": threads}
        for n in (1, 16):
            A, B = Fx.net_inputs(3, n)

            def med(fn, reps=5):
                fn(); ts = []
                for _ in range(reps):
                    t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
                return float(np.median(ts))
            with torch.no_grad():
                tr = med(lambda: model(A, B))
                r = model(A, B)
            tp = med(lambda: O.forward(sd, A, B))
            o = O.forward(sd, A, B)
            out["batch_%d" % n] = {"reference_pairs_per_s": round(n / tr, 2), "port_pairs_per_s": round(n / tp, 2),
                                   "port_over_reference_time": round(tp / tr, 3),
                                   "max_abs_output_diff": max(float((r["trans"] - o["trans"]).abs().max()), float((r["rot"] - o["rot"]).abs().max()))}
        return out
    except Exception as e:   # noqa: BLE001   (the baseline must not take the bench line down)
        return {"value": None, "reason": "reference import failed: %r" % (e,)}


def cpu_baseline(O, sd, nb, inputs=None):
    """The CPU oracle (torch-CPU fp32 restatement of the reference network + numpy pre/post) timed on this box's host cores
    on a bounded sample (~30 s).  `value` = the MEDIAN of CPU_SAMPLES samples of the best whole-path configuration measured in THIS
    run (every configuration is timed for the same duration, split into CPU_SAMPLES consecutive windows; `samples` / `spread`
    = (max - min) / median beside it -- the host is shared, one window is thin).
    The usable cores are min(physical cores, the container's cgroup CPU quota): configurations are one process with Q and Q/2
    threads, K pinned processes x T threads with K T = Q on disjoint physical cores, and ONE over-quota configuration (2 Q
    threads) that documents the throttling.  Batch 1 (what the reference's live tracker runs) beside it."""
    import numpy as np
    import torch
    from oracle import fixtures as Fx
    ncpu = os.cpu_count() or 1
    model, sockets = cpu_model()
    cores = physical_cores()
    P = len(cores)
    quota = cpu_quota_cores()
    Q = max(1, min(P, int(quota) if quota else P))   # cores this container can keep busy
    A, B = inputs if inputs is not None else Fx.net_inputs(3, nb)
    SECONDS, CHUNK = 4.5, 16
    mean, std = Fx.mean_std(0)
    rgb, depth = Fx.synthetic_frame(3); Pz = Fx.pose(3); rgbA, depthA = Fx.synthetic_render(103, 0.8)
    t0 = time.perf_counter()
    for _ in range(8):
        bb = O.compute_bbox(Pz, Fx.K_YCB, 250.0, (1000, 1000, 1000))
        rgbB, depthB = O.crop_bbox(rgb, depth, bb, (176, 176))
        O.process_data(rgbA, depthA, Pz, rgbB, depthB, mean, std)
        O.process_predict(Pz, np.zeros(3, np.float32), np.zeros(3, np.float32))
    pp_s_per_pair = (time.perf_counter() - t0) / 8
    configs = {}
    # single process: network on the timed batch's own pairs in forwards of 16, + the measured numpy pre/post per pair
    old_aff = os.sched_getaffinity(0)
    for th in sorted({t for t in (Q, max(1, Q // 2), 2 * Q) if t <= P} or {P}):
        try:
            os.sched_setaffinity(0, set(cores[:th]))
        except OSError:
            pass
        torch.set_num_threads(th)
        O.forward(sd, A[:CHUNK], B[:CHUNK])
        nets, c0 = [], 0
        for _ in range(CPU_SAMPLES):
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < SECONDS / CPU_SAMPLES:
                O.forward(sd, A[c0:c0 + CHUNK], B[c0:c0 + CHUNK]); n += min(CHUNK, nb - c0)
                c0 = (c0 + CHUNK) % max(CHUNK, nb - nb % CHUNK)
            nets.append(n / (time.perf_counter() - t0))
        med, smp, spread = _median_spread([1.0 / (1.0 / net + pp_s_per_pair) for net in nets])
        configs["1 x %d" % th] = {"processes": 1, "threads": th, "value": med, "samples": smp, "spread": spread,
                                  "network_only": round(float(np.median(nets)), 2)}
    # batch 1: best of a small thread sweep, 10 forwards each
    b1, b1_samples = {}, {}
    for th in sorted({t for t in (max(1, Q // 2), Q) if t <= P} or {P}):
        try:
            os.sched_setaffinity(0, set(cores[:th]))
        except OSError:
            pass
        torch.set_num_threads(th)
        O.forward(sd, A[:1], B[:1])
        smp = []
        for _ in range(CPU_SAMPLES):
            t0 = time.perf_counter()
            for _ in range(8):
                O.forward(sd, A[:1], B[:1])
            smp.append(8.0 / (time.perf_counter() - t0))
        b1[th] = round(float(np.median(smp)), 2)
        b1_samples[th] = [round(v, 2) for v in smp]
    try:
        os.sched_setaffinity(0, old_aff)
    except OSError:
        pass
    # K pinned processes x T threads over the physical cores (whole path inside every worker)
    errors = {}
    for k in (2, 4):
        th = Q // k
        if th < 2:
            continue
        rates, err = cpu_multiprocess_rate(k, th, SECONDS, CHUNK, cores)
        if rates is None:
            errors["%d x %d" % (k, th)] = err
            continue
        med, smp, spread = _median_spread(rates)
        configs["%d x %d" % (k, th)] = {"processes": k, "threads": th, "value": med, "samples": smp, "spread": spread}
    best = max(configs, key=lambda k_: configs[k_]["value"])
    single = max((k_ for k_ in configs if configs[k_]["processes"] == 1), key=lambda k_: configs[k_]["value"])
    b1_best = max(b1, key=b1.get)
    torch.set_num_threads(min(32, P))
    out = {"value": configs[best]["value"], "unit": "pairs/s", "cores": configs[best]["processes"] * configs[best]["threads"],
           "samples": configs[best]["samples"], "spread": configs[best]["spread"],
           "kind": "port", "configuration": best + " (processes x threads, pinned to disjoint physical cores)",
           "cpu_model": model, "sockets": sockets, "logical_cpus": ncpu, "physical_cores": P,
           "cgroup_cpu_quota_cores": quota, "usable_cores": Q,
           "configurations_pairs_per_s": configs,
           "single_process_best": {"configuration": single, "value": configs[single]["value"], "cores": configs[single]["threads"]},
           "batch1": {"value": round(1.0 / (1.0 / b1[b1_best] + pp_s_per_pair), 2), "cores": b1_best,
                      "network_only_by_threads": b1, "network_only_samples_by_threads": b1_samples, "unit": "pairs/s"},
           "prepost_ms_per_pair": round(pp_s_per_pair * 1e3, 3),
           "sample": "oracle (torch-CPU fp32 port of the reference network + numpy crop / normalise / pose update; %s x%d sockets, "
                     "%d physical cores, cgroup CPU quota %s cores): every configuration timed for %.1f s = %d consecutive samples, in forwards of %d "
                     "pairs; value = the median of the best configuration's samples, spread = (max - min) / median"
                     % (model, sockets, P, ("%.0f" % quota) if quota else "none", SECONDS, CPU_SAMPLES, CHUNK)}
    out["reference_vs_port"] = reference_vs_port(O, sd, Q)
    if errors:
        out["configurations_failed"] = errors
    return out


if __name__ == "__main__":
    main()
