"""Batch sharding of the frontend over the GPUs of one node.

Clips are independent (no cross-sample state; the EMA is per (clip, filter) row; parameters are ~1 KB and
replicated), so the path shards over the batch with NO data-path collective.  The only communication the
north star names is the "trivial gather" of the per-rank ``(B_r, F, T')`` outputs, done with one RCCL
``all_gather_into_tensor`` over xGMI (backend "nccl" on ROCm); the same code runs on ``gloo`` for the
CPU tests.  The reference has no counterpart (its only data-parallel code is the TPU trainer,
``train_xla.py:192-196,283``).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_clips: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of ``n_clips`` over ``world`` ranks; the first ``n_clips % world`` ranks get one more."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_clips, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Replicate the (tiny) parameter set from ``src`` so every rank computes with identical filters."""
    stage = dist.get_backend(group) == "gloo"       # a gloo (control-plane) group: through the host, no device collective
    with torch.no_grad():
        for p in module.parameters():
            buf = p.detach().cpu() if (stage and p.is_cuda) else p.detach().clone()
            dist.broadcast(buf, src=src, group=group)
            p.copy_(buf)            # in-place under no_grad: bumps p._version, so Leaf.cache_tables() rebuilds its tables


def gather_features(local: torch.Tensor, n_clips: int, group=None, out: Optional[torch.Tensor] = None,
                    async_op: bool = False):
    """All-gather per-rank ``(B_r, F, T')`` feature blocks into the full ``(n_clips, F, T')`` tensor.

    Shards follow ``shard_bounds``.  Equal shards use one ``all_gather_into_tensor`` straight into ``out``;
    ragged shards are padded to the largest shard, gathered, and trimmed.
    Returns ``out`` (and the work handle when ``async_op``).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_clips, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} clips, expected {hi - lo}")
    tail = tuple(local.shape[1:])
    local = local.contiguous()
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no device collectives: stage through the host (the dry run of bench.py on a box with fewer GPUs than
        # ranks; RCCL -- backend "nccl" -- gathers device to device)
        full = gather_features(local.cpu(), n_clips, group=group).to(local.device)
        if out is not None:
            out.copy_(full)
            full = out
        return (full, None) if async_op else full
    if n_clips % world == 0:
        if out is None:
            out = local.new_empty((n_clips,) + tail)
        work = dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)
        return (out, work) if async_op else out
    biggest = -(-n_clips // world)
    padded = local.new_zeros((biggest,) + tail)
    padded[: hi - lo] = local
    flat = local.new_empty((world * biggest,) + tail)
    dist.all_gather_into_tensor(flat, padded, group=group)
    pieces = []
    for r in range(world):
        a, b = shard_bounds(n_clips, r, world)
        pieces.append(flat[r * biggest: r * biggest + (b - a)])
    full = torch.cat(pieces, dim=0)
    if out is not None:
        out.copy_(full)
        full = out
    return (full, None) if async_op else full


def forward_sharded(frontend, x_full: torch.Tensor, group=None, gather: bool = True) -> torch.Tensor:
    """Run ``frontend`` on this rank's contiguous slice of ``x_full`` (B,1,T) and optionally gather all outputs."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(x_full.shape[0], rank, world)
    if hi > lo:
        local = frontend(x_full[lo:hi])
    else:
        # Fewer clips than ranks: this rank's slice is the empty batch.  leaf_pytorch_amd.Leaf returns (0, F, T') for it like the
        # reference does.  An arbitrary wrapped frontend (a BatchNorm in train mode, a reshape with -1) may refuse an empty
        # batch with a shape error; a rank that died here would leave the others waiting in the gather, so for THAT class of
        # error the empty block is shaped after a one-clip probe.  The probe runs in eval mode under no_grad (no running
        # statistics or counters move on this rank alone) and the mode is restored.  Device faults (out of memory, HIP errors)
        # are not shape errors and are re-raised.  The returned block carries no grad_fn: this rank takes no part in autograd
        # for this call (there is nothing to differentiate); gradient synchronisation across ranks is the caller's concern.
        try:
            local = frontend(x_full[lo:hi])
        except (RuntimeError, ValueError, IndexError, ZeroDivisionError) as e:
            msg = str(e)
            if isinstance(e, torch.OutOfMemoryError) or any(k in msg for k in ("HIP error", "CUDA error", "out of memory", "hipError")):
                raise
            was_training = bool(getattr(frontend, "training", False))
            try:
                if was_training:
                    frontend.eval()
                with torch.no_grad():
                    probe = frontend(x_full[:1])
            finally:
                if was_training:
                    frontend.train()
            local = probe.new_empty((0,) + tuple(probe.shape[1:]))
    return gather_features(local, x_full.shape[0], group=group) if gather else local


def map_peer_buffers(bufs, group=None):
    """Map every rank's tensors ``bufs`` into this process: returns ``peers[r][i]`` = rank r's ``bufs[i]`` (this rank's
    own tensors for r == rank).  One GPU per rank, one node.

    The alternative to a collective KERNEL for the feature gather: a rank writes its block straight into the peers'
    buffers with device-to-peer copies (``peers[r][i][lo:hi].copy_(local, non_blocking=True)``), which the runtime hands to
    the copy engines over xGMI -- no CU is needed, so the copies run beside compute kernels that occupy every CU (the
    default LEAF kernels keep one workgroup with ~all of a CU's LDS on each CU for the whole launch).  The mapping goes through
    torch's own CUDA-IPC tensor sharing (``torch.multiprocessing.reductions``; dmabuf handles: HSA_ENABLE_IPC_MODE_LEGACY=0 on
    this stack), exchanged with ``all_gather_object``.  The owner must keep ``bufs`` alive while peers use the mappings;
    ordering between a writer's copies and the owner's reads is the caller's (events + a barrier, as bench.py does)."""
    from torch.multiprocessing.reductions import reduce_tensor
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    # Every rank takes part in the one collective below whatever happens locally (a rank that raised before it would leave
    # the others waiting forever): a local failure travels as its message and is raised on ALL ranks afterwards.
    try:
        for t in bufs:
            if not (t.is_cuda and t.is_contiguous()):
                raise ValueError("map_peer_buffers needs contiguous device tensors")
        mine = [reduce_tensor(t) for t in bufs]               # (rebuild_fn, args): picklable description of the mapping
    except Exception as e:                                    # noqa: BLE001
        mine = f"rank {rank}: {type(e).__name__}: {e}"
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    failed = [m for m in everyone if isinstance(m, str)]
    if failed:
        raise RuntimeError("map_peer_buffers: " + "; ".join(failed))
    peers = []
    for r in range(world):
        if r == rank:
            peers.append(list(bufs))
        else:
            peers.append([fn(*args) for fn, args in everyone[r]])
    return peers
