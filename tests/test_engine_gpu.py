"""GPU: parity of the HIP engine (through the C ABI) with the CPU oracle — bit-exact, always."""
import hashlib
import os
import random

import pytest
from conftest import GOLDEN, blob_of, dictionary_program, line_expected, line_input, same_modulo_trailing_newlines

from kleenexlang_amd import MatchError, Program, compile_source, workloads
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["delayed", "general"])
def engine_mode(request, monkeypatch):
    """Every test of this file runs twice: with the delayed form where a stage has one (round 5, the default), and with the general
    engine only (KX_DF=0: forward / backward / sweep) — the delayed form's fall-back, and the only engine of programs without one."""
    if request.param == "general":
        monkeypatch.setenv("KX_DF", "0")
    else:
        monkeypatch.delenv("KX_DF", raising=False)
    return request.param


def both(blob, data, **cfg):
    """(engine result, oracle result) where a result is bytes or ('fail', pos)."""
    try:
        want = oracle.run(blob, data)
    except oracle.OracleMatchError as e:
        want = ("fail", e.pos)
    p = Program(blob, **cfg)
    try:
        got = p.run_host(data)
    except MatchError as e:
        got = ("fail", e.pos)
    finally:
        p.close()
    return got, want


def test_loaded_library_is_the_in_tree_hip_engine():
    from kleenexlang_amd import host
    host.load_engine()
    maps = open("/proc/self/maps").read()
    assert os.path.join(host.BUILD_DIR, "libkxhip.so") in maps


@pytest.mark.parametrize("seg", [64, 4096])
def test_reference_vectors(vectors, seg):
    for t in vectors["line_tests"]:
        p = Program(blob_of(t["program"], 0), segment_bytes=seg)
        got = p.run_host(line_input(t["in"]))
        assert same_modulo_trailing_newlines(got, line_expected(t["out"])), t["name"]
        p.close()
    for t in vectors["exact_tests"]:
        p = Program(blob_of(t["program"], 3), segment_bytes=seg)
        for inp, out in t["cases"]:
            assert p.run_host(inp.encode("utf-8")) == out.encode("utf-8"), (t["name"], inp)
        p.close()


def test_reference_samples_and_seeded_goldens(expected):
    for prog, e in expected["samples"].items():
        data = open(os.path.join(GOLDEN, e["input"]), "rb").read()
        p = Program(blob_of(prog))
        got = p.run_host(data)
        assert hashlib.sha256(got).hexdigest() == e["expected"]["sha256"], prog
        p.close()
    for e in expected["synthetic"]:
        data = workloads.generate(workloads.PROGRAM_INPUT[e["program"]], e["nbytes"], e["seed"])
        p = Program(blob_of(e["program"]))
        assert hashlib.sha256(p.run_host(data)).hexdigest() == e["expected"]["sha256"], e
        p.close()


@pytest.mark.parametrize("prog", ["apache_log", "csv2json", "iso_datetime_to_json", "thousand_sep"])
@pytest.mark.parametrize("seg", [64, 192, 4096, 65536])
def test_workloads_vs_oracle_across_segment_sizes(prog, seg):
    blob = blob_of(prog)
    for n, seed in [(3000, 11), (200000, 12), (2500000, 13)]:
        data = workloads.generate(workloads.PROGRAM_INPUT[prog], n, seed)
        got, want = both(blob, data, segment_bytes=seg)
        assert got == want, (prog, seg, n)


def test_ragged_sizes_around_piece_and_segment_boundaries():
    blob = blob_of("flip_ab")
    base = (b"abba\nbab\n" * 2000)
    for n in [0, 1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 4095, 4096, 4097, 8191, 8192, 8193, 12345]:
        for seg in (64, 128, 4096):
            got, want = both(blob, base[:n], segment_bytes=seg)
            assert got == want, (n, seg)


def test_empty_input_and_initial_output():
    assert both(blob_of("apache_log"), b"")[0] == b"[]\n"
    assert both(blob_of("flip_ab"), b"")[0] == b""
    got, want = both(blob_of("main := /ab/"), b"")
    assert got == want == ("fail", 0)


def test_match_error_positions():
    blob = blob_of("apache_log")
    good = workloads.generate("apache_log", 300000, 21)
    for cut in [1, 50, 4096, 70000, 299000]:
        bad = bytearray(good)
        bad[cut] = 0          # NUL never matches inside a log line... unless inside a quoted string
        got, want = both(blob, bytes(bad), segment_bytes=4096)
        assert got == want, cut
    for cut in [0, 10, 5000, 131072, len(good) - 1]:   # truncated input: end of input in a non-final state
        got, want = both(blob, good[:cut], segment_bytes=4096)
        assert got == want, cut
    got, want = both(blob_of("flip_ab"), b"ab" * 100000 + b"x" + b"ab" * 50, segment_bytes=64)
    assert got == want == ("fail", 200000)


def test_config1_add_commas_unbounded_registers_on_gpu():
    """The register form parks MiBs here (SURVEY §7 'unbounded registers'); the path form does not care."""
    blob = blob_of("add_commas")
    for term in (True, False):
        d = workloads.digits(1 << 20, terminated=term)
        got, want = both(blob, d, segment_bytes=1024)
        assert got == want
    mixed = (b"x1234567y99\n" * 30000)
    assert both(blob, mixed, segment_bytes=256)[0] == oracle.run(blob, mixed)


def test_never_synchronising_program_falls_back_to_chaining():
    """Parity of the number of a's decides the output: segment starts never synchronise."""
    src = 'main := even\neven := ~/a/ odd | /b/ even | "E" /\\n/\nodd := ~/a/ even | ~/b/ odd | "O" /\\n/\n'
    blob = blob_of(src)
    data = b"ab" * 5000 + b"a\n"
    p = Program(blob, segment_bytes=64)
    got = p.run_host(data)
    assert got == oracle.run(blob, data)
    assert p.last_stats.unsynced_segments > 0
    p.close()


def test_many_simultaneous_paths():
    blob = blob_of("main := /(a?){32}a{32}\\n/")     # 34 leaves: exercises the wide candidate kernel
    data = b"a" * 40 + b"\n"
    assert both(blob, data)[0] == data
    assert both(blob, b"a" * 31 + b"\n")[0] == ("fail", 31)


def test_long_constants_use_the_wide_entry_form_and_oversize_pieces():
    """Constants of ≥127 bytes escape to the wide side tables; a piece whose output exceeds the staging
    buffer is written by its lane directly (both are rare paths with their own kernel instances)."""
    big = "x" * 300
    src = 'main := (~/a/ "%s" | /b/ | ~/c/ "<%s>")*\n' % (big, "y" * 130)
    blob = blob_of(src)
    for data in [b"a", b"abcab" * 50, b"b" * 1000 + b"a" + b"b" * 1000, b"c" * 64 + b"a" * 64, b"abc" * 3000]:
        for seg in (64, 4096):
            got, want = both(blob, data, segment_bytes=seg)
            assert got == want, (len(data), seg)


def test_constant_on_every_symbol_overflows_the_job_slots_and_the_staging_buffer():
    """k_emit notes constants in a fixed number of per-lane job slots and assembles output in a fixed
    staging buffer: a program that appends a constant on every symbol exhausts both (second sweep that
    copies constants in place; several flush rounds per wave; pieces written straight to global memory)."""
    rng = random.Random(11)
    text = bytes(rng.choice(b"abcdefgh") for _ in range(70000))
    for const in ("<->", "=" * 40, "#" * 100):   # 4x, 41x and 101x expansion (the last exceeds the staging buffer per piece)
        blob = blob_of('main := (/[a-d]/ "%s" | /[e-h]/)*\n' % const)
        for data in [text[:1], text[:63], text[:64], text[:4097], text]:
            for seg in (64, 4096):
                got, want = both(blob, data, segment_bytes=seg)
                assert got == want, (const[:3], len(data), seg)


def test_more_than_64_byte_classes_run_the_general_kernel_instances():
    """Up to 64 byte classes the class table holds class*4 in one byte; beyond that the GENERAL kernel
    instances shift the class index (DevTables::cshift)."""
    alts = " | ".join('/%s/ "%d;"' % (("\\x%02x" % b), b) for b in range(33, 127))
    blob = blob_of("main := (%s | ~/ /)*\n" % alts)
    assert oracle.info(blob)["nclasses"] > 64
    rng = random.Random(5)
    text = bytes(rng.choice(range(32, 127)) for _ in range(50000))
    for data in [b"", text[:5], text[:200], text]:
        for seg in (64, 4096):
            got, want = both(blob, data, segment_bytes=seg)
            assert got == want, (len(data), seg)


def test_large_table_program_one_workgroup_per_cu():
    """A keyword rewriter whose table image (≈ 50 KiB) no longer lets two 512-lane groups of k_backlen share a
    CU's LDS (the kernel then runs 1024-lane groups), and leaves k_emit less LDS per wave."""
    rng = random.Random(42)
    words = sorted({"".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(5, 8))) for _ in range(26)})
    alts = " | ".join('/%s/ "<%d>"' % (w, i) for i, w in enumerate(words))
    blob = blob_of("main := (%s | /[a-z ]/)*\n" % alts)
    info = oracle.info(blob)
    assert 256 + (info["nstates"] + 1) * info["nclasses"] * 4 > 24 * 1024
    text = " ".join(rng.choice(words + ["zz", "qq", "hello", "a"]) for _ in range(60000)).encode()
    for data in [text[:100], text[:70000], text]:
        for seg in (64, 4096):
            got, want = both(blob, data, segment_bytes=seg)
            assert got == want, (len(data), seg)


def test_multi_stage_pipeline_on_device():
    src = 'start: p >> a >> b\np := (~/abc/ "a")*\na := (/./ "b")*\nb := (/ab/ "c" | ~/[^ab]/ "lol")*\n'
    blob = blob_of(src)
    data = b"abc" * 10000
    got, want = both(blob, data, segment_bytes=128)
    assert got == want and len(got) == 30000


def test_device_resident_full_size_config2_tiled_property():
    """BASELINE config 2 at full size: 1 GiB apache_log, checked on the GPU against the tiled
    expectation derived from the oracle's output on one base chunk (no CPU pass over 1 GiB)."""
    import torch
    blob = blob_of("apache_log")
    t, base, k = workloads.device_input("apache_log", 1 << 30, "cuda:0", base_bytes=8 << 20)
    want = workloads.tiled_expected("apache_log", oracle.run(blob, base), k)
    p = Program(blob)
    out = p.run_tensor(t)
    torch.cuda.synchronize()
    assert out.numel() == len(want)
    exp = torch.frombuffer(bytearray(want), dtype=torch.uint8).to("cuda:0")
    assert bool(torch.equal(out, exp))
    p.close()


def test_device_resident_csv_and_datetime_tiled_property():
    import torch
    for prog in ("csv2json", "iso_datetime_to_json"):
        blob = blob_of(prog)
        t, base, k = workloads.device_input(prog, 256 << 20, "cuda:0", base_bytes=4 << 20)
        one = oracle.run(blob, base)
        p = Program(blob)
        out = p.run_tensor(t)
        torch.cuda.synchronize()
        assert out.numel() == len(one) * k
        exp = torch.frombuffer(bytearray(one), dtype=torch.uint8).to("cuda:0")
        assert bool(torch.equal(out.view(k, len(one)), exp.expand(k, len(one))))
        p.close()


def test_produced_binary_stdin_stdout_contract(tmp_path):
    """kexc compile --out BIN; BIN < in > out ; -t ; rejection message + exit code."""
    import subprocess
    from kleenexlang_amd import build, program_path
    exe = tmp_path / "apache"
    r = subprocess.run([os.path.join(build.OUT, "kexc"), "compile", "--quiet", program_path("apache_log"), "--out", str(exe)])
    assert r.returncode == 0
    data = workloads.generate("apache_log", 500000, 5)
    r = subprocess.run([str(exe), "-t"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == oracle.run(blob_of("apache_log"), data)
    assert b"time (ms): " in r.stderr
    r = subprocess.run([str(exe)], input=data[:1000], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and r.stdout == b"" and r.stderr.endswith(b"Match error at input symbol 1000!\n")


def test_produced_binary_writes_regular_files_in_place(tmp_path):
    """`BIN < in > out` on regular files (input read with positioned parallel reads, output written in order): output lands
    after whatever the descriptor already stood behind (`> out` opened by a shell that wrote a header first), is appended
    in `>>` mode, and the descriptor's position ends behind it."""
    import subprocess
    from kleenexlang_amd import build, program_path
    exe = tmp_path / "apache"
    assert subprocess.run([os.path.join(build.OUT, "kexc"), "compile", "--quiet", program_path("apache_log"), "--out", str(exe)]).returncode == 0
    data = workloads.generate("apache_log", 9 << 20, 77)
    want = oracle.run(blob_of("apache_log"), data)
    src = tmp_path / "in.log"
    src.write_bytes(data)
    for env in ({}, {"KX_WINDOW_BYTES": str(1 << 20), "KX_READ_THREADS": "3"}):
        e = {**os.environ, **env}
        out = tmp_path / "out.json"
        with open(src, "rb") as fi, open(out, "wb") as fo:
            fo.write(b"HEADER\n"); fo.flush()
            assert subprocess.run([str(exe)], stdin=fi, stdout=fo, env=e).returncode == 0
            assert os.lseek(fo.fileno(), 0, os.SEEK_CUR) == 7 + len(want)
            os.write(fo.fileno(), b"TRAILER\n")
        assert out.read_bytes() == b"HEADER\n" + want + b"TRAILER\n"
        with open(src, "rb") as fi, open(out, "ab") as fo:
            assert subprocess.run([str(exe)], stdin=fi, stdout=fo, env=e).returncode == 0
        assert out.read_bytes() == b"HEADER\n" + want + b"TRAILER\n" + want


def test_produced_binary_streams_inputs_in_windows(tmp_path):
    """kx_run_fd keeps a bounded window of the input in HBM: every window is one shard of the sharded
    protocol (start state from the previous window, end leaf from the next).  Windows far smaller than the
    input, through a pipe and from a regular file, single- and multi-stage, accepting and rejecting."""
    import subprocess
    from kleenexlang_amd import build, program_path
    kexc = os.path.join(build.OUT, "kexc")
    two_stage = tmp_path / "two.kex"
    two_stage.write_text('start: commas >> brackets\n'
                         'commas := (num /\\n/)*\nnum := digit{1,3} ("," digit{3})*\ndigit := /[0-9]/\n'
                         'brackets := (/[0-9]+/ "<" | /,/ ">" | /\\n/)*\n')
    cases = [("apache_log", program_path("apache_log"), workloads.generate("apache_log", 700000, 9)),
             ("thousand_sep", program_path("thousand_sep"), workloads.generate(workloads.PROGRAM_INPUT["thousand_sep"], 300000, 3)),
             ("two_stage", str(two_stage), workloads.generate("numbers", 200000, 4))]
    for name, path, data in cases:
        exe = tmp_path / name
        assert subprocess.run([kexc, "compile", "--quiet", path, "--out", str(exe)]).returncode == 0
        want = oracle.run(blob_of(open(path).read()), data)
        infile = tmp_path / (name + ".in")
        infile.write_bytes(data)
        for window in (4096, 65536, 262144, 1 << 30):
            env = dict(os.environ, KX_WINDOW_BYTES=str(window))
            r = subprocess.run([str(exe)], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)   # pipe
            assert r.returncode == 0 and r.stdout == want, (name, window, r.stderr[-200:])
            with open(infile, "rb") as f:                                                                          # regular file
                r = subprocess.run([str(exe)], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert r.returncode == 0 and r.stdout == want, (name, window, "file")
        # rejection in a later window: the symbol count is global, and no output beyond the accepted windows appears
        bad = data[:150000] + b"\x00" + data[150000:]
        try:
            oracle.run(blob_of(open(path).read()), bad)
            pos = None
        except oracle.OracleMatchError as e:
            pos = e.pos
        if pos is not None:
            r = subprocess.run([str(exe)], input=bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KX_WINDOW_BYTES="65536"))
            assert r.returncode == 1 and r.stderr.endswith(b"Match error at input symbol %d!\n" % pos), (name, r.stderr[-100:])
            # Contract (all programs, multi-stage included): what reaches stdout before a rejection is a PREFIX of the output on
            # the uncorrupted input.  A stage places a window's output only once the NEXT window has resolved its end leaf, and
            # a later stage only ever sees complete windows of the stage before it — so every byte written was produced from
            # input that the good and the bad run share, along a path that both runs resolve alike; a stage that has seen the
            # rejection upstream is never run to its end-of-input (that would emit the final state's output for a truncated
            # stream: the reference's later phase sees EOF there and fails or flushes, crt.c:414-454 — never more than a prefix).
            assert want.startswith(r.stdout), (name, len(r.stdout))


def test_windowed_multi_stage_last_window_emits_nothing(tmp_path):
    """when the last window of stage 1 produces no bytes, stage 2 receives an empty LAST window behind
    pending ones; the pending window must then be resolved with the final state's leaf."""
    import subprocess
    from kleenexlang_amd import build
    kexc = os.path.join(build.OUT, "kexc")
    src = ('start: strip >> lines\n'
           'strip := (~/x/ | /[a\\n]/)*\n'
           'lines := /a+\\n/ rest\nrest := "," lines | "."\n')
    path = tmp_path / "two.kex"
    path.write_text(src)
    exe = tmp_path / "two"
    assert subprocess.run([kexc, "compile", "--quiet", str(path), "--out", str(exe)]).returncode == 0
    data = (b"aaa\n" * 3000) + b"x" * 20000          # the trailing windows hold only suppressed bytes
    want = oracle.run(blob_of(src), data)
    assert want.endswith(b"aaa\n.") and want.count(b",") == 2999
    for window in (4096, 8192, 1 << 20):
        r = subprocess.run([str(exe)], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, KX_WINDOW_BYTES=str(window)))
        assert r.returncode == 0 and r.stdout == want, (window, r.stderr[-200:], r.stdout[-20:])


@pytest.mark.parametrize("prog", ["apache_log", "csv2json", "iso_datetime_to_json"])
def test_baseline_10gib_configs_every_output_byte_checked_on_device(prog):
    """BASELINE configs 3 and 5 (and the headline apache_log run) at their full 10 GiB, with the segment size the
    engine picks itself at that size: EVERY output byte is compared on the device with the tiled expectation built
    from the CPU oracle's output on one base chunk (the reference's equality check is on full outputs, bench/Makefile:73-127)."""
    import torch
    blob = blob_of(prog)
    t, base, k = workloads.device_input(prog, 10 << 30, "cuda:0", base_bytes=8 << 20)
    parts = workloads.tiled_parts(prog, oracle.run(blob, base), k)
    total = workloads.tiled_total(parts)
    out = torch.empty(total + 4096, dtype=torch.uint8, device="cuda:0")
    p = Program(blob)
    ol = p.run_device(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ol == total
    assert workloads.check_tiled_on_device(out[:ol], 0, parts)
    p.close()
    del out, t
    torch.cuda.empty_cache()


def test_rccl_boundary_handoff_two_and_four_ranks_on_one_device():
    """The sharded protocol over the real `nccl` (= RCCL) backend: bench.py launches itself with N ranks stacked on
    cuda:0 (the box has one GPU), every rank runs real HIP shards, the boundary tuples travel through RCCL
    all-gathers, and every output byte of every rank is verified on the device."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for world in (2, 4):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--single-device", "--gib", "0.25",
                            "--steps", "2", "--warmup", "1", "--no-cpu"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        line = json.loads([x for x in r.stdout.decode().splitlines() if x.startswith('{"metric"')][-1])
        assert line["n_gpus"] == world and line["config"]["boundary_backend"] == "nccl"
        assert line["config"]["boundary_driver"].startswith("kx_run_sharded")   # the library's own driver and RCCL communicator
        assert line["output_checked_bit_exact"] is True and line["output_bytes_checked"] == line["config"]["output_bytes_total"]


def test_layout_follows_the_constants_per_piece_of_earlier_shards(tmp_path, monkeypatch):
    """(General engine: the delayed form has one layout.)  Round 4: a stage that qualifies keeps two table images — vote-and-rank job noting and the job-stride layout with counted job
    slots — and every shard (a run of the program object, a window of a streamed input) picks the one that suits the constants per
    piece the shards before it held: iso_datetime_to_json (21 per piece) moves to the job-stride layout after its first shard,
    apache_log (4) stays.  Same bytes either way: consecutive runs on one Program, and a produced binary streaming its input in
    small windows (the switch happens in mid-stream)."""
    import subprocess
    from kleenexlang_amd import build, program_path
    monkeypatch.setenv("KX_DF", "0")
    for prog, shape in (("iso_datetime_to_json", "datetime"), ("apache_log", "apache_log"), ("csv2json", "csv")):
        blob = blob_of(prog)
        data = workloads.generate(shape, 3 << 20, 21)
        want = oracle.run(blob, data)
        p = Program(blob)
        try:
            for i in range(4):
                assert p.run_host(data) == want, (prog, i)
            assert p.run_host(data[:70000].rsplit(b"\n", 1)[0] + b"\n") == oracle.run(blob, data[:70000].rsplit(b"\n", 1)[0] + b"\n")
        finally:
            p.close()
    exe = tmp_path / "iso"
    assert subprocess.run([os.path.join(build.OUT, "kexc"), "compile", "--quiet", program_path("iso_datetime_to_json"), "--out", str(exe)]).returncode == 0
    data = workloads.generate("datetime", 2 << 20, 22)
    want = oracle.run(blob_of("iso_datetime_to_json"), data)
    r = subprocess.run([str(exe)], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KX_WINDOW_BYTES="262144", KX_DEBUG="1"))
    assert r.returncode == 0 and r.stdout == want
    assert b"layout=vote-and-rank" in r.stderr and b"layout=job-stride" in r.stderr    # both layouts ran inside this one stream


def test_sharded_run_reports_a_local_failure_on_every_rank():
    """kx_run_sharded's return code is collective.  One of three ranks has no room for its output (a local failure
    between two exchanges); every rank must come back with an error — the failing one with its own message, the others naming
    it — instead of waiting in the next all-gather for a peer that has already returned."""
    import threading
    import torch
    from kleenexlang_amd.host import EngineError, Group
    blob = blob_of("apache_log")
    data = workloads.generate("apache_log", 2 << 20, 12)
    world = 3
    grp = Group(world)
    L = (len(data) // world) // 4096 * 4096
    errs, done = [None] * world, [False] * world

    def body(r):
        lo, hi = r * L, (len(data) if r == world - 1 else (r + 1) * L)
        p = Program(blob)
        t = torch.frombuffer(bytearray(data[lo:hi]), dtype=torch.uint8).to("cuda:0")
        out = torch.empty(4 * (hi - lo) + 65536, dtype=torch.uint8, device="cuda:0")
        mb = grp.member(r)
        try:
            p.run_sharded(r, world, mb, t.data_ptr(), hi - lo, out.data_ptr(), 4096 if r == 1 else out.numel())   # rank 1: 4 KiB of room
        except Exception as e:   # noqa: BLE001
            errs[r] = e
        finally:
            done[r] = True
            mb.close(); p.close()
    th = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    [x.start() for x in th]
    [x.join(timeout=120) for x in th]
    grp.close()
    assert all(done), "a rank is still waiting for its peers"
    assert all(isinstance(e, EngineError) for e in errs), errs
    assert "too small" in str(errs[1]) and all("rank 1" in str(errs[r]) for r in (0, 2)), errs


def test_c_driver_shards_inside_one_process_and_in_the_binary(tmp_path):
    """kx_run_sharded (include/kxhip.h): the boundary hand-off is run by the library itself.  (1) Three and five ranks as
    threads of this process, exchange through kx_group_*: shards cut mid-line, a two-stage pipeline, a rejected input
    (every rank reports the same global position).  (2) `BIN --gpus N < file > out` (KX_SHARD_SAME_DEVICE=1: the box has
    one GPU): regular file and pipe on stdout, match error text and exit code."""
    import subprocess
    import threading
    import torch
    from kleenexlang_amd import build, program_path
    from kleenexlang_amd.host import Group

    def run_ranks(blob, data, world):
        grp = Group(world)
        L = (len(data) // world) // 4096 * 4096
        outs, errs = [None] * world, [None] * world

        def body(r):
            try:
                lo, hi = r * L, (len(data) if r == world - 1 else (r + 1) * L)
                p = Program(blob)
                t = torch.frombuffer(bytearray(data[lo:hi] or b"\0"), dtype=torch.uint8).to("cuda:0")
                out = torch.empty(4 * (hi - lo) + 65536, dtype=torch.uint8, device="cuda:0")
                mb = grp.member(r)
                try:
                    res = p.run_sharded(r, world, mb, t.data_ptr(), hi - lo, out.data_ptr(), out.numel())
                    torch.cuda.synchronize()
                    outs[r] = (int(res.out_offset), bytes(out[:int(res.out_len)].cpu().numpy().tobytes()), int(res.total_out))
                finally:
                    mb.close(); p.close()
            except Exception as e:   # noqa: BLE001
                errs[r] = e
        th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        [x.start() for x in th]; [x.join() for x in th]
        grp.close()
        return outs, errs

    blob = blob_of("apache_log")
    data = workloads.generate("apache_log", 3 << 20, 11)
    want = oracle.run(blob, data)
    for world in (3, 5):
        outs, errs = run_ranks(blob, data, world)
        assert errs == [None] * world, errs
        assert b"".join(o[1] for o in outs) == want and all(o[2] == len(want) for o in outs)
        assert [o[0] for o in outs] == [sum(len(x[1]) for x in outs[:r]) for r in range(world)]
    # the inline-constant layout (thousand_sep) and a coder's symbol tables (one table: translated per piece) across shards
    from kleenexlang_amd import host
    for blob_x, data_x in ((blob_of("thousand_sep"), workloads.generate("numbers", 2 << 20, 5)),
                           (host.compile_regex("(([^,\\n]*),([^,\\n]*)\\n)*"), b"".join(b"ab%d,x%dyz\n" % (i, i * 7) for i in range(150000)))):
        want_x = oracle.run(blob_x, data_x)
        outs, errs = run_ranks(blob_x, data_x, 4)
        assert errs == [None] * 4, errs
        assert b"".join(o[1] for o in outs) == want_x
    # two stages: the second stage's shards are the first stage's output slices
    src2 = 'start: a >> b\na := (~/x/ "yy" | /[^x]/)*\nb := (~/yy/ "z" | /./)*\n'
    blob2 = blob_of(src2)
    d2 = bytes(random.Random(3).choice(b"xyab\n") for _ in range(200000))
    outs, errs = run_ranks(blob2, d2, 3)
    assert errs == [None] * 3 and b"".join(o[1] for o in outs) == oracle.run(blob2, d2)
    # rejection: the global position, on every rank
    cut = data.index(b"\n", 1 << 20) + 1
    bad = data[:cut] + b"not a log line\n" + data[cut:]
    outs, errs = run_ranks(blob, bad, 3)
    assert all(isinstance(e, MatchError) for e in errs) and {e.pos for e in errs} == {oracle_fail_pos(blob, bad)}
    # the binary
    exe = tmp_path / "apache"
    assert subprocess.run([os.path.join(build.OUT, "kexc"), "compile", "--quiet", program_path("apache_log"), "--out", str(exe)]).returncode == 0
    src = tmp_path / "in.log"; src.write_bytes(data)
    env = dict(os.environ, KX_SHARD_SAME_DEVICE="1")
    for world in (2, 4):
        out = tmp_path / ("out%d.json" % world)
        with open(src, "rb") as fi, open(out, "wb") as fo:
            fo.write(b"HDR\n"); fo.flush()
            assert subprocess.run([str(exe), "--gpus", str(world)], stdin=fi, stdout=fo, env=env).returncode == 0
            assert os.lseek(fo.fileno(), 0, os.SEEK_CUR) == 4 + len(want)
        assert out.read_bytes() == b"HDR\n" + want
        with open(src, "rb") as fi:
            r = subprocess.run([str(exe), "--gpus", str(world)], stdin=fi, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0 and r.stdout == want
    (tmp_path / "bad.log").write_bytes(bad)
    with open(tmp_path / "bad.log", "rb") as fi:
        r = subprocess.run([str(exe), "--gpus", "3"], stdin=fi, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 1 and r.stdout == b"" and r.stderr.endswith(b"Match error at input symbol %d!\n" % oracle_fail_pos(blob, bad))
    r = subprocess.run([str(exe), "--gpus", "2"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)   # a pipe on stdin
    assert r.returncode == 1 and b"regular file" in r.stderr


def oracle_fail_pos(blob, data):
    try:
        oracle.run(blob, data)
    except Exception as e:   # noqa: BLE001
        return e.pos
    raise AssertionError("input was accepted")


def test_never_merging_program_resolves_in_linear_work():
    """a program whose surviving path is only decided by the LAST byte never lets the blocks'
    candidates merge; the end leaves are then resolved by one backward pass over the blocks (k_resolve_leaf), not by
    a walk per block.  16 Ki blocks here; both outcomes."""
    src = 'main := /[ab]*/ "!" | /[ab]*/ ~/c/ "?"\n'
    blob = blob_of(src)
    rnd = random.Random(5)
    body = bytes(rnd.choice(b"ab") for _ in range(1 << 16)) * 256          # 16 MiB
    for tail, mark in ((b"", b"!"), (b"c", b"?")):
        data = body + tail
        p = Program(blob, segment_bytes=1024)
        got = p.run_host(data)
        assert got == body + mark
        p.close()
    small = body[:5000] + b"c"
    got, want = both(blob, small, segment_bytes=64)
    assert got == want == body[:5000] + b"?"


def test_phase_option_runs_one_stage_of_a_pipeline(tmp_path):
    """`BIN --phase K` runs only phase K, stdin -> stdout (crt/crt.c:390-393,408-411): piping phase 1 into phase 2 by hand
    equals the whole pipeline; an invalid phase gives the reference's message and exit 1."""
    import subprocess
    from kleenexlang_amd import build
    kexc = os.path.join(build.OUT, "kexc")
    src = ('start: commas >> brackets\n'
           'commas := (num /\\n/)*\nnum := digit{1,3} ("," digit{3})*\ndigit := /[0-9]/\n'
           'brackets := (/[0-9]+/ "<" | /,/ ">" | /\\n/)*\n')
    path = tmp_path / "two.kex"; path.write_text(src)
    exe = tmp_path / "two"
    assert subprocess.run([kexc, "compile", "--quiet", str(path), "--out", str(exe)]).returncode == 0
    data = workloads.generate("numbers", 50000, 4)
    whole = subprocess.run([str(exe)], input=data, stdout=subprocess.PIPE).stdout
    assert whole == oracle.run(blob_of(src), data)
    p1 = subprocess.run([str(exe), "--phase", "1"], input=data, stdout=subprocess.PIPE).stdout
    assert p1 != whole and b"," in p1 and b"<" not in p1
    p2 = subprocess.run([str(exe), "-p", "2"], input=p1, stdout=subprocess.PIPE).stdout
    assert p2 == whole
    r = subprocess.run([str(exe), "--phase", "3"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Invalid phase: 3 given" in r.stderr


def test_pipeline_handed_over_as_tables_runs_on_the_engine(tmp_path):
    """The compile-side seam (kexc_emit_pipeline = compileProgram): a hand-marshalled flip_ab program and a workload
    program taken apart into tables both run on the GPU like their source-compiled twins."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import kxp
    from test_pipeline_seam import flip_ab_program, marshal
    from kleenexlang_amd import emit_pipeline
    out = tmp_path / "flip.kxp"
    emit_pipeline([flip_ab_program()], srcout=out)
    p = Program(out.read_bytes())
    data = b"abba\nbb\n" * 5000
    assert p.run_host(data) == b"baab\naa\n" * 5000
    with pytest.raises(MatchError) as e:
        p.run_host(b"ab" * 40 + b"x")
    assert e.value.pos == 80
    p.close()
    emit_pipeline([marshal(s) for s in kxp.parse(blob_of("csv2json"))], srcout=out)
    data = workloads.generate("csv", 300000, 8)
    p = Program(out.read_bytes())
    assert p.run_host(data) == oracle.run(blob_of("csv2json"), data)
    p.close()


def test_register_actions_on_the_engine(tmp_path):
    """programs with register actions run on the GPU — transducer (tokens in band) + action post-pass —
    and give the reference's vectors (actionbug, makeDanish), the oracle's output on larger inputs (arena growth: frames
    and registers of megabytes), and the same through the produced binary in small windows (tokens cut by window ends,
    frames and registers carried from window to window)."""
    import json
    import subprocess
    from kleenexlang_amd import build
    from test_register_actions import PROGRAMS
    av = json.load(open(os.path.join(GOLDEN, "action_vectors.json"), encoding="utf-8"))
    for t in av["line_tests"]:      # (makeDanish, 658 states x 47 classes, runs in the BIG table form: DESIGN.md §3)
        p = Program(blob_of(t["program"], 0))
        assert p.stage_has_actions(0)
        got = p.run_host(line_input(t["in"]))
        assert same_modulo_trailing_newlines(got, line_expected(t["out"])), t["name"]
        p.close()
    rnd = random.Random(3)
    words = lambda k: b"".join(bytes(rnd.choice(b"abcxyz") for _ in range(rnd.randint(1, 9))) + b"," + str(rnd.randint(0, 10 ** rnd.randint(1, 8))).encode() + b"\n" for _ in range(k))
    cases = [("swap_fields", words(60000)), ("accumulate", b"".join(bytes(rnd.choice(b"abc") for _ in range(rnd.randint(1, 6))) + b" " for _ in range(6000))),   # (quadratic: acc is copied per word)
             ("nested", b"a" * 3000000 + b"b" * 70000), ("byte_ff", bytes(rnd.choice(b"ab\xff\xfe\n") for _ in range(500000)) + b"\n"),
             ("two_stage", bytes(rnd.choice(b"abcz") for _ in range(400000)))]
    kexc = os.path.join(build.OUT, "kexc")
    for name, data in cases:
        src = PROGRAMS[name]
        want = oracle.run(blob_of(src), data)
        p = Program(blob_of(src), segment_bytes=4096)
        assert p.run_host(data) == want, name
        p.close()
        path = tmp_path / (name + ".kex"); path.write_text(src)
        exe = tmp_path / name
        assert subprocess.run([kexc, "compile", "--quiet", str(path), "--out", str(exe)]).returncode == 0
        for window in (4096, 50000, 1 << 30):
            r = subprocess.run([str(exe)], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KX_WINDOW_BYTES=str(window)))
            assert r.returncode == 0 and r.stdout == want, (name, window, r.stderr[-300:])
    # sharding such a stage is refused
    p = Program(blob_of(PROGRAMS["swap_fields"]))
    with pytest.raises(Exception, match="register actions"):
        p.shard_begin(0, 0, 0, True, True)
    p.close()


def test_action_post_pass_replays_independent_chunks_in_parallel(tmp_path, monkeypatch, capfd):
    """the post-pass cuts the token stream at safe points (stack at the bottom buffer, every register
    empty) and replays the chunks by thousands of waves (k_actions_chunks).  Tiny chunks here, so that a few hundred KB make
    hundreds of them: line programs (cut after every line), the escaped byte FF next to cuts, two stages; programs without
    safe points (one frame around everything, a register alive to the end) and with more than 32 registers fall back to
    the one-wave replay; windows carry the state into a parallel replay; an 8 MiB stream with the default sizes."""
    import subprocess
    from kleenexlang_amd import build
    from test_register_actions import PROGRAMS
    monkeypatch.setenv("KX_ACT_PAR_MIN", "8192")
    monkeypatch.setenv("KX_ACT_CHUNK", "2048")
    monkeypatch.setenv("KX_DEBUG", "1")
    rnd = random.Random(9)
    words = lambda k: b"".join(bytes(rnd.choice(b"abcxyz") for _ in range(rnd.randint(0, 9))) + b"," + str(rnd.randint(0, 10 ** rnd.randint(1, 8))).encode() + b"\n" for _ in range(k))
    many = 'main := (' + " ".join('r%d@/%s/' % (i, "ab"[i & 1]) for i in range(40)) + " " + " ".join("!r%d" % i for i in reversed(range(40))) + ' /\n/)*\n'
    cases = [("swap_fields", PROGRAMS["swap_fields"], words(40000), True),
             ("byte_ff", PROGRAMS["byte_ff"], bytes(rnd.choice(b"ab\xff\xff\xfe\n") for _ in range(300000)) + b"\n", True),
             ("two_stage", PROGRAMS["two_stage"], bytes(rnd.choice(b"abcz") for _ in range(200000)), True),
             ("long_lines", 'main := (l@/[^\n]*/ ~/\n/ "<" !l ">\n")*\n', b"".join(bytes(rnd.choice(b"abcdefgh \xff") for _ in range(rnd.randrange(0, 700))) + b"\n" for _ in range(800)), True),
             ("nested", PROGRAMS["nested"], b"a" * 300000 + b"b" * 7000, False),
             ("accumulate", PROGRAMS["accumulate"], b"".join(bytes(rnd.choice(b"abc") for _ in range(rnd.randint(1, 6))) + b" " for _ in range(4000)), False),
             ("forty_registers", many, (b"ab" * 20 + b"\n") * 3000, False)]
    kexc = os.path.join(build.OUT, "kexc")
    for name, src, data, parallel in cases:
        blob = blob_of(src)
        want = oracle.run(blob, data)
        p = Program(blob)
        capfd.readouterr()
        assert p.run_host(data) == want, name
        err = capfd.readouterr().err
        assert ("chunks of about" in err) == parallel, (name, err[-300:])
        p.close()
        if name in ("swap_fields", "byte_ff", "accumulate"):
            path = tmp_path / (name + ".kex"); path.write_text(src)
            exe = tmp_path / name
            assert subprocess.run([kexc, "compile", "--quiet", str(path), "--out", str(exe)]).returncode == 0
            for window in (30000, 1 << 30):
                r = subprocess.run([str(exe)], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KX_WINDOW_BYTES=str(window)))
                assert r.returncode == 0 and r.stdout == want, (name, window, r.stderr[-300:])
    monkeypatch.delenv("KX_ACT_PAR_MIN"); monkeypatch.delenv("KX_ACT_CHUNK")
    data = words(700000)
    blob = blob_of(PROGRAMS["swap_fields"])
    p = Program(blob)
    capfd.readouterr()
    assert p.run_host(data) == oracle.run(blob, data)
    assert "chunks of about" in capfd.readouterr().err
    p.close()


def test_tables_beyond_16_bit_addressing_run_from_global_memory(monkeypatch):
    """a program whose image exceeds 64 KiB (make_danish-sized: ~1000 states x 28 classes) is not
    refused any more — its GENERAL kernel instances read the image from global memory with scaled handles
    (DevTables::big).  `KX_FORCE_BIG` sends the ordinary workloads down the same instances."""
    src, words = dictionary_program()
    blob = blob_of(src, opt=0)
    info = oracle.info(blob)
    assert 256 + (info["nstates"] + 1) * info["nclasses"] * 4 > 100 * 1024
    rnd = random.Random(9)
    other = ["zz", "hello", "a", "x1", "0", words[3][:-1], words[7] + "s", words[11][1:]]
    seps = [" ", ", ", "\n", " - ", ".  ", "\t", " (", ") "]
    text = "".join(rnd.choice(words if rnd.random() < 0.4 else other) + rnd.choice(seps) for _ in range(60000)).encode()
    for data in [b"", b"a", text[:57], text[:4097], text[:100001], text]:
        for seg in (64, 4096, 0):
            got, want = both(blob, data, segment_bytes=seg)
            assert got == want, (len(data), seg)
    got, want = both(blob, b" " + text[:5000], segment_bytes=64)     # a leading separator is the one way to fail
    assert got == want == ("fail", 0)
    monkeypatch.setenv("KX_FORCE_BIG", "1")
    for prog in ("apache_log", "csv2json", "iso_datetime_to_json", "thousand_sep"):
        for seg in (64, 4096, 0):
            for n, seed in [(3000, 31), (300000, 32)]:
                data = workloads.generate(workloads.PROGRAM_INPUT[prog], n, seed)
                got, want = both(blob_of(prog), data, segment_bytes=seg)
                assert got == want, (prog, seg, n)
    big = "x" * 300     # wide entries + oversize pieces on the BIG instances
    blob = blob_of('main := (~/a/ "%s" | /b/ | ~/c/ "<%s>")*\n' % (big, "y" * 130))
    for data in [b"a", b"abcab" * 50, b"c" * 64 + b"a" * 64, b"abc" * 3000]:
        got, want = both(blob, data, segment_bytes=64)
        assert got == want, len(data)


def test_forward_walk_variants_agree(monkeypatch):
    """k_forward's instances: two-symbol pair table + cooperative line loads (default where the tables fit LDS),
    cooperative loads alone (`KX_NO_PAIR`), pair table with per-lane loads and the plain walk (`KX_NO_COOP`, + `KX_NO_PAIR`) —
    same bytes and the same failure positions, on ragged sizes (trips of eight pieces are wave-uniform: lanes with short or
    unsynchronised segments only help loading)."""
    good = workloads.generate("apache_log", 500000, 41)
    cut = good.index(b"\n", 300000) + 1
    cases = [("apache_log", good), ("apache_log", good[:cut] + b"\n" + good[cut:]), ("apache_log", good[:cut + 33]),
             ("csv2json", workloads.generate("csv", 300000, 42)), ("thousand_sep", workloads.generate("numbers", 200001, 43)),
             ("flip_ab", b"ab" * 150000 + b"x" + b"ab" * 50), ("flip_ab", b"ab" * 70000)]
    for env in ({}, {"KX_NO_PAIR": "1"}, {"KX_NO_COOP": "1"}, {"KX_NO_COOP": "1", "KX_NO_PAIR": "1"}):
        for k in ("KX_NO_PAIR", "KX_NO_COOP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for prog, data in cases:
            for seg in (512, 4096, 0):
                got, want = both(blob_of(prog), data, segment_bytes=seg)
                assert got == want, (env, prog, len(data), seg)


def test_kexc_simulate_runs_the_program_on_the_engine(tmp_path):
    """`kexc simulate` / `interpret` (stdin → pipeline → stdout, Commands.hs:277-323).  `--sim sst` is the
    compiled program on the HIP engine; `lockstep` (the default) and `backtrack` are the reference's FST simulators on the CPU
    (csrc/kexc/simulate.cpp, tests/test_simulators.py) — three routes, one output.  Rejections use the reference simulators'
    words ("Reject" for the FST simulations, Commands.hs:285; SymbolicSST.hs:425-427 for `--sim sst`), exit code 1, nothing on
    stdout."""
    import subprocess
    from kleenexlang_amd import build, program_path
    kexc = os.path.join(build.OUT, "kexc")
    data = workloads.generate("csv", 200000, 51)
    want = oracle.run(blob_of("csv2json"), data)
    for args in (["simulate"], ["simulate", "--sim", "backtrack"], ["interpret", "--sim=sst"], ["simulate", "--opt", "0", "--sim", "sst"]):
        r = subprocess.run([kexc, *args, program_path("csv2json")], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == want, (args, r.stderr[-300:])
    src = tmp_path / "two.kex"
    two = 'start: a >> b\na := (/x/ "yy" | /y/ | /\\n/)*\nb := (~/y/ "z" | ~/x/ | /\\n/)*\n'
    src.write_text(two)
    r = subprocess.run([kexc, "simulate", str(src)], input=b"xyx\n" * 1000, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == b"zzzzz\n" * 1000 == oracle.run(blob_of(two), b"xyx\n" * 1000), r.stderr
    flip = program_path("flip_ab")
    for sim, inp, msg in (("lockstep", b"abx", b"Reject\n"), ("backtrack", b"abx", b"Reject\n"), ("sst", b"abxab", b"No match\n"),
                          ("sst", b"ab\n" * 3 + b"a", None)):
        r = subprocess.run([kexc, "simulate", "--sim", sim, flip], input=inp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if msg is None:     # flip_ab accepts every prefix over {a, b, \\n}: accepted
            assert r.returncode == 0
        else:
            assert r.returncode == 1 and r.stdout == b"" and r.stderr.endswith(msg), (sim, r.stderr)
    eof = tmp_path / "eof.kex"
    eof.write_text('main := /a/ /b/\n')
    r = subprocess.run([kexc, "simulate", "--sim", "sst", str(eof)], input=b"a", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and r.stderr.endswith(b"End of input reached, but final state is not accepting.\n"), r.stderr


@pytest.mark.gpu
def test_action_mirrors_in_lds_fall_back_to_global_memory():
    """k_actions mirrors the top of the frame stack and small registers in LDS (16 KiB each).  Frames and registers that
    outgrow the mirrors, registers written long after they were popped (the register arena has been reset in between),
    deep nesting and many registers must all give the oracle's bytes."""
    rnd = random.Random(99)
    cases = []
    # one frame far larger than the mirror, then written twice (the second write finds it empty)
    cases.append(('main := big@/[a-z]*/ ~/;/ "<" !big "|" !big ">"\n', [b"abc;", bytes(rnd.choice(b"abcdefgh") for _ in range(70000)) + b";"]))
    # 26 registers filled one after the other with lines of 0..3000 bytes (the register arena wraps), written in reverse order
    regs = "abcdefghijklmnopqrstuvwxyz"
    src = "main := " + " ".join("r%s@line" % c for c in regs) + " " + " ".join("!r%s" % c for c in reversed(regs)) + "\nline := /[^\\n]*/ ~/\\n/\n"
    lines = [bytes(rnd.choice(b"0123456789") for _ in range(rnd.choice([0, 1, 17, 900, 3000]))) for _ in regs]
    cases.append((src, [b"".join(l + b"\n" for l in lines)]))
    # nesting: inner registers popped inside an outer frame that keeps growing past the mirror
    cases.append(('main := o@(item*) "[" !o "]"\nitem := i@/[a-z]+/ ~/,/ "(" !i ")"\n',
                  [b"ab,c,", b"".join(bytes(rnd.choice(b"xyz") for _ in range(rnd.randrange(1, 400))) + b"," for _ in range(400))]))
    for src, inputs in cases:
        blob = compile_source(src)
        prog = Program(blob)
        for data in inputs:
            assert prog.run_host(data) == oracle.run(blob, data), (src[:40], len(data))
        prog.close()


def test_inline_constant_layout(monkeypatch):
    """One-byte constants ride in their path entries (DevTables::inl, piece_sweep2i): a program whose constants are mostly
    one byte takes the layout by itself; KX_INL=1 forces it onto the workloads (few one-byte constants beside long ones, the
    jobs still run); bytes >= 0x80 as inline constants (they look like jobs to the sweep); a constant on EVERY symbol with
    longer ones between (staging rounds, oversize pieces); KX_INL=0 is the ordinary layout.  Engine = oracle throughout."""
    rng = random.Random(23)
    text = bytes(rng.choice(b"abcdefgh,\n") for _ in range(300000))
    progs = ['main := (/[a-d]/ "1" | /[e-h]/ "\\xfe" | ~/,/ ";" | /\\n/ "<end of line>\\n")*\n',
             'main := (/[a-h]/ "." | ~/,/ | ~/\\n/ "%s")*\n' % ("=" * 90),
             'main := (~/[a-d]/ "x" | /[e-h,]/ | /\\n/ "\\x80\\x81")*\n']
    for src in progs:
        blob = blob_of(src)
        for inl in ("", "0", "1"):
            if inl:
                monkeypatch.setenv("KX_INL", inl)
            for data in [text[:1], text[:64], text[:4097], text]:
                for seg in (64, 4096, 0):
                    got, want = both(blob, data, segment_bytes=seg)
                    assert got == want, (src[:30], inl, len(data), seg)
            bad = text[:100000] + b"Z" + text[:50]
            got, want = both(blob, bad)
            assert got == want == ("fail", 100000)
            monkeypatch.delenv("KX_INL", raising=False)
    monkeypatch.setenv("KX_INL", "1")
    for prog in ["apache_log", "csv2json", "iso_datetime_to_json", "thousand_sep"]:
        data = workloads.generate(workloads.PROGRAM_INPUT[prog], 6 << 20, 31)
        for seg in (0, 4096):
            got, want = both(blob_of(prog), data, segment_bytes=seg)
            assert got == want, (prog, seg)
    monkeypatch.setenv("KX_EMIT_STG", "1024")   # tiny staging: rounds and pieces written straight to global memory
    got, want = both(blob_of(progs[1]), text)
    assert got == want


def test_lane_replay_keeps_short_strings_in_registers(monkeypatch, capfd):
    """k_actions_lanes (one lane per chunk) keeps frames and registers of at most 8 bytes in VGPRs and everything longer in its
    arenas: fields of 0..20 bytes cross that line in both directions — a short register written into a long frame, a long one
    into a short frame, frames pushed over a cached frame (nesting inside a line), empty fields, the byte FF."""
    monkeypatch.setenv("KX_DEBUG", "1")
    monkeypatch.setenv("KX_ACT_PAR_MIN", "20000")
    monkeypatch.setenv("KX_ACT_CHUNK", "2048")
    monkeypatch.setenv("KX_ACT_LANES", "1")
    monkeypatch.setenv("KX_ACT_PREFIX3_MIN", "64")     # (the three-step prefix of the block summaries, as on streams of GiBs)
    rnd = random.Random(77)
    word = lambda lo, hi, abc: bytes(rnd.choice(abc) for _ in range(rnd.randrange(lo, hi)))
    progs = [
        # swap: both fields 0..20 bytes
        ('main := (a@/[a-z\\xff]*/ ~/,/ b@/[0-9]*/ !b "," !a /\\n/)*\n', lambda: word(0, 21, b"abcxyz\xff") + b"," + word(0, 21, b"0123456789") + b"\n"),
        # nesting inside the line: c is captured inside the frame of a, written back into it, a doubled
        ('main := (a@(/[a-z]*/ c@/[0-9]*/ "<" !c !c ">") ~/;/ "[" !a "|" !a "]" /\\n/)*\n', lambda: word(0, 12, b"abc") + word(0, 12, b"0123") + b";\n"),
        # a register appended to piecewise ([r += ...]) and written once per line
        ('main := (line)*\nline := [r <- ""] (w@/[a-z]+/ [r += w "."] | ~/,/)* ~/;/ !r /\\n/\n', lambda: b"".join(word(1, 4, b"abc") + b"," for _ in range(rnd.randrange(0, 8))) + b";\n"),
    ]
    for src, line in progs:
        data = b"".join(line() for _ in range(30000))
        blob = blob_of(src)
        want = oracle.run(blob, data)
        p = Program(blob)
        capfd.readouterr()
        assert p.run_host(data) == want, src[:40]
        err = capfd.readouterr().err
        assert "one lane each" in err, err[-400:]
        p.close()
