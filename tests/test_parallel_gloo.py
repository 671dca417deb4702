"""world_size-2 gloo tests (CPU) of the clip-parallel path: sharding, the visual-token all-gather (even and uneven
shards), and that every rank ends up with the same tokens a single process would compute."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vitron_amd.parallel import (all_gather_direct_p2p, all_gather_visual_tokens, encode_clips_parallel, plan_prefill_placement,
                                 sequences_of_rank, shard_range, start_all_gather_visual_tokens, visual_tokens_for_rank)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stub_encode(batch):  # [n,3,T,H,W] -> [n,T,P,Hd] deterministic function of the pixels
    n, _, T, H, W = batch.shape
    base = batch.float().mean(dim=(1, 3, 4))  # [n, T]
    return (base[:, :, None, None] + torch.arange(4)[None, None, :, None] * 10 + torch.arange(3)[None, None, None, :]).to(torch.bfloat16)


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clips = [torch.full((3, 2, 4, 4), float(i + 1)) for i in range(n_clips)]
        out = encode_clips_parallel(_stub_encode, clips)
        ref = _stub_encode(torch.stack(clips))
        ok = out.shape == ref.shape and torch.equal(out, ref)
        s, e = shard_range(n_clips, world, rank)
        loc = torch.arange(s, e, dtype=torch.float32).reshape(-1, 1)
        g = all_gather_visual_tokens(loc, n_clips)
        ok = ok and torch.equal(g.flatten(), torch.arange(n_clips, dtype=torch.float32))
        # asynchronous variant (even shards): started, other work happens, then waited for
        h = start_all_gather_visual_tokens(torch.full((1, 2, 3), float(rank)))
        busy = torch.ones(8).sum()
        ga = h.wait()
        ok = ok and busy.item() == 8 and ga.shape == (world, 2, 3) and all(bool((ga[r] == r).all()) for r in range(world))
        # direct full-mesh point-to-point variant: same result as the collective, blocking and asynchronous
        loc2 = torch.full((2, 3, 5), float(rank + 1)) + torch.arange(5)
        ref2 = torch.cat([torch.full((2, 3, 5), float(r + 1)) + torch.arange(5) for r in range(world)], 0)
        ok = ok and torch.equal(all_gather_direct_p2p(loc2), ref2)
        hp = all_gather_direct_p2p(loc2, async_op=True)
        ok = ok and torch.equal(hp.wait(), ref2) and torch.equal(hp.wait(), ref2)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [2, 3, 8, 1])
def test_clip_parallel_all_gather_gloo(n_clips):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_range_partitions():
    for n in range(0, 20):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker_placement(rank, world, port, q):
    """Uneven prompts: clips are encoded where shard_range puts them, sequences are prefilled where plan_prefill_placement puts them; the
    rank's "prefill" (a stub: a checksum of its sequences' visual tokens and lengths) must see exactly the tokens a single process would."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_clips = 8
        lens = [128, 896, 256, 768, 384, 640, 512, 512]                    # prompt tokens per sequence (mean 512)
        clips = [torch.full((3, 2, 4, 4), float(i + 1)) for i in range(n_clips)]
        ref = _stub_encode(torch.stack(clips))                              # what one process computes for all clips
        gathered = encode_clips_parallel(_stub_encode, clips)               # every rank encoded only its shard
        place = plan_prefill_placement([ref.shape[1] * ref.shape[2] + n for n in lens], world)
        mine = sequences_of_rank(place, rank)
        enc_s, enc_e = shard_range(n_clips, world, rank)
        foreign = [i for i in mine if not enc_s <= i < enc_e]               # sequences whose clip another rank encoded: the gather's consumer
        toks = visual_tokens_for_rank(gathered, place, rank)
        ok = toks.shape[0] == len(mine) and all(torch.equal(toks[k], ref[i]) for k, i in enumerate(mine))
        # every sequence is prefilled exactly once across the ranks, and the plan is the same on every rank
        owners = [torch.zeros(n_clips, dtype=torch.int64) for _ in range(world)]
        mine_t = torch.zeros(n_clips, dtype=torch.int64)
        mine_t[mine] = 1
        dist.all_gather(owners, mine_t)
        ok = ok and bool((torch.stack(owners).sum(0) == 1).all())
        q.put((rank, bool(ok), len(foreign)))
    finally:
        dist.destroy_process_group()


def test_prefill_placement_consumes_the_gathered_tokens_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_placement, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True)]
    assert sum(f for _, _, f in res) > 0          # at least one sequence was prefilled away from where its clip was encoded


def test_prefill_placement_balances_cost():
    from vitron_amd.parallel import prefill_cost
    lens = [4608 + n for n in (128, 896, 256, 768, 384, 640, 512, 512)]
    for world in (1, 2, 4, 8):
        place = plan_prefill_placement(lens, world)
        assert sorted(set(place)) == list(range(world)) and len(place) == 8
        load = [sum(prefill_cost(lens[i]) for i in sequences_of_rank(place, r)) for r in range(world)]
        assert max(load) <= 1.10 * (sum(load) / world) + (prefill_cost(max(lens)) if world == 8 else 0.0) * 0.2
    assert plan_prefill_placement([5, 5], 2) == [0, 1]
