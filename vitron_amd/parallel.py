"""Clip-parallel video encoding across the GPUs of one node (BASELINE config 4).

The reference has no multi-GPU path (vitron/ contains no torch.distributed call). The natural shard unit is the CLIP
(or image): the video tower's temporal attention couples the 8 frames of a clip in every layer
(reference modeling_video.py:105-127), so frames of one clip never leave a GPU. Each rank encodes its clips, then ONE
all-gather moves the projected visual tokens over xGMI (RCCL, `backend="nccl"`) so that every rank can prefill any
sequence; prefill itself is data parallel. No other collective is on the path.

Works with any torch.distributed backend: the tests run it under gloo on CPU tensors with a stub encoder.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [start, stop) of n_items for `rank` (first n_items % world ranks get one more)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_visual_tokens(local: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """local: [n_local, ...] tokens of this rank's shard (shard_range order) -> [n_items, ...] on every rank.
    One all_gather_into_tensor when the shards are even (the benchmark case); padded to the largest shard otherwise."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [shard_range(n_items, world, r)[1] - shard_range(n_items, world, r)[0] for r in range(world)]
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    mx = max(counts)
    if mx == 0:
        return local
    send = local.contiguous()
    if counts[rank] != mx:
        pad = torch.zeros((mx - counts[rank], *local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([send, pad], 0)
    out = torch.empty((world * mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if all(c == mx for c in counts):
        return out
    return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(world)], 0)


class PendingGather:
    """Handle of an all-gather started with start_all_gather_visual_tokens: `.wait()` orders the current stream behind the
    collective (it does not block the host) and returns the [n_items, ...] tensor."""

    def __init__(self, work, out: torch.Tensor):
        self._work, self._out = work, out

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._out


def start_all_gather_visual_tokens(local: torch.Tensor, group=None) -> PendingGather:
    """Even shards only (one clip per rank, the benchmark layout): start the all-gather and return at once, so that work that
    only needs THIS rank's tokens (its own prefill) overlaps the xGMI transfer. local [n, ...] -> wait() gives [world*n, ...]."""
    world = dist.get_world_size(group)
    send = local.contiguous()
    out = torch.empty((world * send.shape[0], *send.shape[1:]), dtype=send.dtype, device=send.device)
    work = dist.all_gather_into_tensor(out, send, group=group, async_op=True)
    return PendingGather(work, out)


def all_gather_direct_p2p(local: torch.Tensor, group=None, async_op: bool = False):
    """The same even-shard all-gather as ONE batch of point-to-point transfers: every rank sends its shard straight to each of
    the other world-1 ranks and receives theirs (dist.batch_isend_irecv; on RCCL one grouped launch whose sends leave over
    the world-1 xGMI links of the GPU at once). xGMI is a full mesh of point-to-point links (7 x ~153 GB/s per GPU), so the
    direct exchange moves each shard over exactly one link, once, while the ring all-gather forwards every shard over
    world-1 consecutive hops: ~0.25 ms vs ~1.7 ms for the 37.75 MB shards of BASELINE config 4.
    local [n, ...] -> [world * n, ...]; async_op=True returns a PendingGather instead."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    send = local.contiguous()
    n = send.shape[0]
    out = torch.empty((world * n, *send.shape[1:]), dtype=send.dtype, device=send.device)
    out[rank * n:(rank + 1) * n].copy_(send)
    ops = []
    for step in range(1, world):                       # peer order staggered by rank: every link pair is busy in every step
        dst, src = (rank + step) % world, (rank - step) % world
        ops.append(dist.P2POp(dist.isend, send, dst if group is None else dist.get_global_rank(group, dst), group))
        ops.append(dist.P2POp(dist.irecv, out[src * n:(src + 1) * n], src if group is None else dist.get_global_rank(group, src), group))
    works = dist.batch_isend_irecv(ops) if ops else []
    pending = PendingGather(_WorkList(works), out)
    return pending if async_op else pending.wait()


class _WorkList:
    def __init__(self, works):
        self._works = list(works)

    def wait(self):
        for w in self._works:
            w.wait()


def encode_clips_parallel(encode: Callable[[torch.Tensor], torch.Tensor], clips: Sequence[torch.Tensor], group=None) -> torch.Tensor:
    """clips: the GLOBAL list of clips [3,T,H,W] (every rank holds the list, or at least its own shard's entries).
    Each rank runs `encode` (tower + projector -> [n, T, P, H]) on its shard only; returns all clips' tokens."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    s, e = shard_range(len(clips), world, rank)
    if e > s:
        local = encode(torch.stack([clips[i] for i in range(s, e)]))
    else:
        probe = encode(torch.stack([clips[0]]))  # shape/dtype only (more ranks than clips)
        local = probe[:0]
    return all_gather_visual_tokens(local, len(clips), group)


# ---- prefill placement != encode placement: the case that NEEDS the gather ---------------------------------------------
# Clips are encoded where shard_range puts them (every clip costs the same: 8 frames through the tower), but prompts differ in length, and a
# prefill costs its sequence's tokens (linear layers) plus their square (attention): the rank that prefills a sequence is chosen by that
# cost, not by where its clip was encoded -- so a rank prefills sequences whose visual tokens arrived over the all-gather.
def prefill_cost(tokens: int, hidden: int = 4096, layers_linear_flop_per_token: float = 12.95e9) -> float:
    """Algorithmic prefill FLOP of one sequence: 12.95 GFLOP per token in the linear layers + 4 H S (S + 1) / 2 per layer x 32."""
    return layers_linear_flop_per_token * tokens + 32 * 4.0 * hidden * tokens * (tokens + 1) / 2.0


def plan_prefill_placement(seq_tokens: Sequence[int], world: int) -> List[int]:
    """sequence index -> rank: longest-processing-time greedy on prefill_cost (ties to the lower rank; deterministic, every rank computes the
    same plan from the same lengths). Balanced lengths reproduce shard_range's contiguous blocks only by accident -- callers must take their
    sequences' visual tokens from the GATHERED tensor."""
    order = sorted(range(len(seq_tokens)), key=lambda i: (-prefill_cost(seq_tokens[i]), i))
    load = [0.0] * world
    place = [0] * len(seq_tokens)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        place[i] = r
        load[r] += prefill_cost(seq_tokens[i])
    return place


def sequences_of_rank(placement: Sequence[int], rank: int) -> List[int]:
    return [i for i, r in enumerate(placement) if r == rank]


def visual_tokens_for_rank(gathered: torch.Tensor, placement: Sequence[int], rank: int) -> torch.Tensor:
    """gathered [n_items, ...] (clip order = sequence order) -> the visual tokens of the sequences `rank` prefills, in sequence order."""
    idx = sequences_of_rank(placement, rank)
    if not idx:
        return gathered[:0]
    return gathered[torch.tensor(idx, dtype=torch.long, device=gathered.device)]
