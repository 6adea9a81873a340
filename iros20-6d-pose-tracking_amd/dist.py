"""Frame-level sharding of the hot path across the GPUs of one node: one process per GPU,
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The path is embarrassingly parallel over pairs: the only exchanges are
  (C1) a one-off broadcast of the packed weight blob (54 MB: the BN-folded float32 panels, blob v7; every rank derives its
       Winograd planes and, if selected, its f16x3 split panels on its own device) from rank 0, and
  (C2) a per-step all-gather of the [n_local,16] float64 poses (KBs, latency-bound).
No all-reduce exists anywhere on the path."""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous [lo, hi) slice of `n_total` pairs owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_blob(blob_or_none, nbytes, device, src=0, group=None):
    """C1: rank `src` passes the packed blob (uint8 tensor, any device); every rank returns a uint8
    tensor of `nbytes` on `device` holding the same bytes."""
    if dist.get_rank(group) == src:
        assert blob_or_none is not None and blob_or_none.numel() == nbytes
        buf = blob_or_none.to(device).contiguous()
    else:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src, group=group)
    return buf


def load_weights_everywhere(engine, state_dict_or_none, src=0, group=None, device=None):
    """rank `src` folds+packs the reference state_dict; all ranks bind the broadcast blob.  `device`: where the blob is
    received (default: the engine's GPU; "cpu" only in the gloo dry run of bench.py)."""
    blob = engine.pack_state_dict(state_dict_or_none) if dist.get_rank(group) == src else None
    buf = broadcast_blob(blob, engine.packed_bytes(), device or "cuda:%d" % engine.device, src, group)
    engine.bind_blob(buf)
    return buf


def gather_poses(local_poses, group=None):
    """C2: all-gather equally sized [n_local,16] pose shards -> [world*n_local,16] on every rank."""
    world = dist.get_world_size(group)
    out = torch.empty((world * local_poses.shape[0],) + tuple(local_poses.shape[1:]),
                      dtype=local_poses.dtype, device=local_poses.device)
    dist.all_gather_into_tensor(out, local_poses.contiguous(), group=group)
    return out


def gather_poses_async(local_poses, group=None):
    """C2 without putting the collective on the compute stream's critical path: returns (out, work);
    the all-gather runs on the backend's own stream (after everything already queued on the current
    stream), the caller keeps computing into a DIFFERENT pose buffer and calls work.wait() before it
    reads `out` or reuses `local_poses`."""
    world = dist.get_world_size(group)
    out = torch.empty((world * local_poses.shape[0],) + tuple(local_poses.shape[1:]),
                      dtype=local_poses.dtype, device=local_poses.device)
    work = dist.all_gather_into_tensor(out, local_poses.contiguous(), group=group, async_op=True)
    return out, work


def shard_round_robin(items, rank, world):
    """Object-level parallelism (BASELINE configs[4]: 21 YCB objects, per-object weights, 8 GPUs): item i is
    owned by rank i % world -- objects differ ~3x in frame count, interleaving evens that out better than
    contiguous blocks."""
    return [it for i, it in enumerate(items) if i % world == rank]


def gather_objects(obj, group=None):
    """Every rank contributes one picklable object (e.g. {class_id: error arrays}); all ranks get the list
    in rank order.  Host-side, once per evaluation: uses all_gather_object (no device traffic)."""
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out
