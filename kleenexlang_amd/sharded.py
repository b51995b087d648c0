"""Input sharded over ranks (one contiguous shard per GPU) — host side of SURVEY.md §8e.

The data path needs no collective: every rank scans, sizes and emits its own shard.  What
crosses ranks is the chunk-boundary hand-off, a few tiny all-gathers per pipeline stage:

  forward   (end_known, end_state, head_len, fail_pos, n)  → state entering each shard
  backward  (constant, nleaves, start_leaf[≤256])          → leaf in which each shard must end
  sizes     out_len                                        → global output offset of each shard

On GPUs the all-gathers run over RCCL (torch.distributed backend "nccl"); the CPU tests run the
same protocol over gloo with a CPU stand-in for the shard kernels.  xGMI bandwidth is irrelevant
here (≤ 300 bytes per rank per exchange): the exchange is latency-bound.

`stage_protocol` is written as a coroutine that *yields* the array it wants all-gathered and is
*sent* the per-rank list, so the very same code is driven by torch.distributed (`run_stage_dist`)
or, for several shards inside one process, by `run_stage_local`.
"""
import numpy as np

from .host import NOFAIL, MatchError


def pack_fwd(s, n):
    fail = -1 if s.fail_pos == NOFAIL else int(s.fail_pos)
    return np.array([s.synced, s.end_state, s.head_len, fail, n], dtype=np.int64)


def pack_bwd(s):
    a = np.zeros(8 + 256, dtype=np.uint8)
    a[0] = 1 if s.constant else 0
    a[4:8] = np.frombuffer(np.uint32(s.nleaves).tobytes(), dtype=np.uint8)
    a[8:] = np.frombuffer(bytes(s.start_leaf), dtype=np.uint8)
    return a


def chain_end_leaves(bwd):
    """End leaf of every rank: rank r must end in the leaf rank r+1 starts in (the last rank's end
    leaf is fixed by the program's final state and is resolved by the shard itself)."""
    world = len(bwd)
    ends = [0] * world
    for r in range(world - 2, -1, -1):
        ends[r] = int(bwd[r + 1][8 + ends[r + 1]])
    return ends


def stage_protocol(shard, rank, world, n):
    """One rank's side of one pipeline stage.  Yields arrays to all-gather; finally returns
    ("ok", out_len, out_offset, total_out)  or  ("fail", global_fail_pos)."""
    fs = shard.forward()
    fwd = yield pack_fwd(fs, n)
    fixed = [False] * world
    while not all(fixed):
        known = [fixed[r] or bool(int(fwd[r][0])) for r in range(world)]
        todo = [r for r in range(world) if not fixed[r] and (r == 0 or known[r - 1])]
        if rank in todo:
            fs = shard.fix_head(int(fwd[rank - 1][1]) if rank else 0)
        for r in todo:
            fixed[r] = True
        fwd = yield pack_fwd(fs, n)
    offs = np.concatenate([[0], np.cumsum([int(f[4]) for f in fwd])])
    fails = [int(offs[r]) + int(f[3]) for r, f in enumerate(fwd) if int(f[3]) >= 0]
    if fails:
        return ("fail", fails[0])  # lowest rank = earliest position
    bs = shard.backward()
    bwd = yield pack_bwd(bs)
    ends = chain_end_leaves(bwd)
    out_len = shard.resolve(ends[rank])
    lens = yield np.array([out_len], dtype=np.int64)
    lens = [int(x[0]) for x in lens]
    return ("ok", out_len, sum(lens[:rank]), sum(lens))


def run_stage_dist(shard, n, device=None):
    """Drive this process's shard with torch.distributed collectives (RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = device if device is not None else "cpu"
    gen = stage_protocol(shard, rank, world, n)
    msg = next(gen)
    while True:
        t = torch.from_numpy(np.ascontiguousarray(msg)).to(dev)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        gathered = [o.cpu().numpy() for o in outs]
        try:
            msg = gen.send(gathered)
        except StopIteration as e:
            return e.value


def run_stage_local(shards, sizes):
    """Drive `len(shards)` shards inside one process in lock step (no communication library)."""
    world = len(shards)
    gens = [stage_protocol(s, r, world, sizes[r]) for r, s in enumerate(shards)]
    msgs = [next(g) for g in gens]
    results = [None] * world
    while any(r is None for r in results):
        gathered = [np.array(m, copy=True) for m in msgs]
        for r, g in enumerate(gens):
            if results[r] is not None:
                continue
            try:
                msgs[r] = g.send(gathered)
            except StopIteration as e:
                results[r] = e.value
    return results


def raise_on_fail(result, stage=0):
    if result[0] == "fail":
        raise MatchError(result[1], stage)
    return result
