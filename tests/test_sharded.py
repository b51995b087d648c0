"""The chunk-boundary hand-off protocol (kleenexlang_amd/sharded.py).

CPU: world_size-2 (and 3) gloo processes, each owning one shard evaluated by the CPU stand-in
(tests/kxp.py); the concatenated shard outputs must equal the oracle's output on the whole input.
GPU: the same protocol over real HIP shards, several shards in one process on one device."""
import os
import socket
import sys

import pytest
from conftest import blob_of

from kleenexlang_amd import sharded, workloads
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, prog, data_parts, q):
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    import kxp
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stage = kxp.parse(blob_of(prog))[0]
        sh = kxp.CpuShard(stage, data_parts[rank], rank == 0, rank == world - 1)
        res = sharded.run_stage_dist(sh, len(data_parts[rank]))
        q.put((rank, res, sh.emit() if res[0] == "ok" else None))
    finally:
        dist.destroy_process_group()


def _run_gloo(prog, parts):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, len(parts), port, prog, parts, q)) for r in range(len(parts))]
    for p in ps: p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps: p.join(30)
    return got


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_two_and_three_shards_match_oracle(world):
    prog = "apache_log"
    data = workloads.generate("apache_log", 6000, 31)
    cuts = [len(data) * i // world + 7 * i for i in range(world)] + [len(data)]   # mid-line cuts
    parts = [data[cuts[i]:cuts[i + 1]] for i in range(world)]
    got = _run_gloo(prog, parts)
    assert all(r[1][0] == "ok" for r in got)
    out = b"".join(r[2] for r in got)
    assert out == oracle.run(blob_of(prog), data)
    offs = [r[1][2] for r in got]
    assert offs == [sum(len(g[2]) for g in got[:i]) for i in range(world)] and got[0][1][3] == len(out)


def test_gloo_failure_position_is_global():
    data = workloads.generate("apache_log", 4000, 32)
    bad = data[:3000] + b"\x00" + data[3001:]
    parts = [bad[:2000], bad[2000:]]
    got = _run_gloo("apache_log", parts)
    with pytest.raises(oracle.OracleMatchError) as e:
        oracle.run(blob_of("apache_log"), bad)
    assert all(r[1] == ("fail", e.value.pos) for r in got)


def test_gloo_unsynchronised_shard_chains_states():
    """A program whose shard starts never synchronise: states must be chained rank by rank."""
    src = 'main := even\neven := ~/a/ odd | /b/ even | "E" /\\n/\nodd := ~/a/ even | ~/b/ odd | "O" /\\n/\n'
    data = b"abbab" * 40 + b"a\n"
    parts = [data[:70], data[70:150], data[150:]]
    got = _run_gloo(src, parts)
    assert b"".join(r[2] for r in got) == oracle.run(blob_of(src), data)


def test_local_driver_matches_oracle_on_cpu_standin():
    sys.path.insert(0, HERE)
    import kxp
    for prog, shape in [("csv2json", "csv"), ("iso_datetime_to_json", "datetime"), ("thousand_sep", "numbers")]:
        data = workloads.generate(shape, 5000, 33)
        stage = kxp.parse(blob_of(prog))[0]
        for world in (1, 2, 4):
            cuts = [len(data) * i // world for i in range(world)] + [len(data)]
            shards = [kxp.CpuShard(stage, data[cuts[i]:cuts[i + 1]], i == 0, i == world - 1) for i in range(world)]
            res = sharded.run_stage_local(shards, [cuts[i + 1] - cuts[i] for i in range(world)])
            assert all(r[0] == "ok" for r in res)
            assert b"".join(s.emit() for s in shards) == oracle.run(blob_of(prog), data), (prog, world)


@pytest.mark.gpu
@pytest.mark.parametrize("prog,shape", [("apache_log", "apache_log"), ("csv2json", "csv"), ("add_commas", None)])
def test_hip_shards_in_one_process(prog, shape):
    """k shards on one GPU through the real kx_shard_* entry points == one unsharded run."""
    import torch
    from kleenexlang_amd import Program
    blob = blob_of(prog)
    data = workloads.generate(shape, 3000000, 41) if shape else workloads.digits(500000)
    want = oracle.run(blob, data)
    for world in (2, 5):
        cuts = [((len(data) * i // world) // 16) * 16 + (0 if i == 0 else 0) for i in range(world)] + [len(data)]
        tens = [torch.frombuffer(bytearray(data[cuts[i]:cuts[i + 1]]), dtype=torch.uint8).to("cuda:0") for i in range(world)]
        progs = [Program(blob, segment_bytes=1024) for _ in range(world)]
        shards = [progs[i].shard_begin(0, tens[i].data_ptr(), tens[i].numel(), i == 0, i == world - 1) for i in range(world)]
        res = sharded.run_stage_local(shards, [t.numel() for t in tens])
        assert all(r[0] == "ok" for r in res), res
        outs = []
        for i, s in enumerate(shards):
            o = torch.empty(max(res[i][1], 1), dtype=torch.uint8, device="cuda:0")
            s.emit(o.data_ptr(), res[i][1])
            outs.append(bytes(o[:res[i][1]].cpu().numpy().tobytes()))
            s.end()
        assert b"".join(outs) == want, (prog, world)
        for p in progs: p.close()


def test_empty_last_shard_resolves_its_predecessor_with_the_final_leaf():
    """a last shard (or window) with n == 0 must hand back the FINAL state's leaf, not the identity map —
    otherwise the shard before it is resolved with end leaf 0 and silently emits another path's output."""
    sys.path.insert(0, HERE)
    import kxp
    for prog, shape in [("apache_log", "apache_log"), ("csv2json", "csv"), ("iso_datetime_to_json", "datetime")]:
        data = workloads.generate(shape, 4000, 35)
        stage = kxp.parse(blob_of(prog))[0]
        for parts in ([data, b""], [data[:1500], data[1500:], b""], [b"", data, b""]):
            shards = [kxp.CpuShard(stage, p, i == 0, i == len(parts) - 1) for i, p in enumerate(parts)]
            res = sharded.run_stage_local(shards, [len(p) for p in parts])
            assert all(r[0] == "ok" for r in res)
            assert b"".join(s.emit() for s in shards) == oracle.run(blob_of(prog), data), (prog, [len(p) for p in parts])


@pytest.mark.gpu
def test_hip_empty_last_and_empty_middle_shards():
    import torch
    from kleenexlang_amd import Program
    for prog, shape in [("apache_log", "apache_log"), ("csv2json", "csv"), ("iso_datetime_to_json", "datetime")]:
        blob = blob_of(prog)
        data = workloads.generate(shape, 300000, 36)
        want = oracle.run(blob, data)
        cut = (len(data) // 2) // 16 * 16
        for parts in ([data, b""], [data[:cut], b"", data[cut:]], [data[:cut], data[cut:], b""]):
            world = len(parts)
            tens = [torch.frombuffer(bytearray(p if p else b"\0" * 16), dtype=torch.uint8).to("cuda:0") for p in parts]
            progs = [Program(blob, segment_bytes=1024) for _ in range(world)]
            shards = [progs[i].shard_begin(0, tens[i].data_ptr(), len(parts[i]), i == 0, i == world - 1) for i in range(world)]
            res = sharded.run_stage_local(shards, [len(p) for p in parts])
            assert all(r[0] == "ok" for r in res), res
            outs = []
            for i, s in enumerate(shards):
                o = torch.empty(max(res[i][1], 16), dtype=torch.uint8, device="cuda:0")
                s.emit(o.data_ptr(), res[i][1])
                outs.append(bytes(o[:res[i][1]].cpu().numpy().tobytes()))
                s.end()
            assert b"".join(outs) == want, (prog, [len(p) for p in parts])
            for p in progs: p.close()


def test_thread_group_allgather_and_abort():
    """kx_group_* (the produced binary's `--gpus N`: ranks = threads, exchange through host memory): every member sees every
    member's record in rank order, round after round; a member that gives up wakes the others instead of leaving them in the
    barrier.  (Host code of libkxhip.so: runs without a GPU.)"""
    import ctypes
    import threading
    from kleenexlang_amd import host
    lib = host.load_engine()
    lib.kx_group_allgather.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    lib.kx_group_abort.argtypes = [ctypes.c_void_p]
    world = 5
    g = host.Group(world)
    seen = [None] * world

    def body(r):
        mb = g.member(r)
        got = []
        for rnd in range(50):
            rec = ctypes.create_string_buffer(bytes([r, rnd & 0xFF]) * 8, 16)
            out = ctypes.create_string_buffer(16 * world)
            assert lib.kx_group_allgather(mb._h, rec, out, 16) == 0
            got.append(out.raw)
        seen[r] = got
        mb.close()
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    for r in range(world):
        for rnd in range(50):
            assert seen[r][rnd] == b"".join(bytes([q, rnd]) * 8 for q in range(world))
    g.close()
    # abort: four members wait, the fifth gives up
    g = host.Group(world)
    rcs = [None] * world

    def waiter(r):
        mb = g.member(r)
        rec = ctypes.create_string_buffer(16); out = ctypes.create_string_buffer(16 * world)
        rcs[r] = lib.kx_group_allgather(mb._h, rec, out, 16)
        mb.close()
    th = [threading.Thread(target=waiter, args=(r,)) for r in range(world - 1)]
    [t.start() for t in th]
    lib.kx_group_abort(g._h)
    [t.join(timeout=20) for t in th]
    assert all(not t.is_alive() for t in th) and all(rc is not None and rc != 0 for rc in rcs[:world - 1])
    g.close()
