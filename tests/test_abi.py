"""CPU: the C-ABI libraries load and export every symbol their headers declare (no compute)."""
import ctypes
import os
import re
import subprocess

from kleenexlang_amd import build

INC = os.path.join(build.ROOT, "include")


def _declared(header):
    txt = open(os.path.join(INC, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(k(?:x|exc)_[a-z_0-9]+)\s*\(", txt)))


def test_kxhip_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(build.OUT, "libkxhip.so"))
    names = _declared("kxhip.h")
    assert len(names) >= 17, names
    for n in names:
        assert hasattr(lib, n), "libkxhip.so lacks %s" % n


def test_kexc_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(build.OUT, "libkexc.so"))
    names = _declared("kexc_api.h")
    assert set(names) == {"kexc_compile", "kexc_emit_c", "kexc_emit_pipeline_v2", "kexc_dump_fst", "kexc_compile_regex", "kexc_dump_regex_fst",
                          "kexc_compile_flags", "kexc_dump_words",
                          "kexc_last_error", "kexc_free"}
    for n in names:
        assert hasattr(lib, n)


def test_engine_is_gfx950_code_object():
    so = os.path.join(build.OUT, "libkxhip.so")
    out = subprocess.run(["strings", "-a", so], stdout=subprocess.PIPE, check=True).stdout
    assert b"gfx950" in out


def test_engine_refuses_to_run_without_a_gpu():
    """No CPU fallback: without a HIP device kx_load fails loudly (skipped on the GPU box)."""
    import torch
    import pytest
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from kleenexlang_amd import EngineError, Program, compile_file
    with pytest.raises(EngineError, match="no HIP device|HIP"):
        Program(compile_file("flip_ab"))


def test_corrupt_and_truncated_blobs_are_refused_not_read_out_of_bounds():
    """every section and index of a blob is validated before use (kx_validate = the checks of kx_load,
    without a device).  Truncations at every length and a few thousand single-field corruptions must give an error
    or, where the change is harmless, a clean pass — never a crash."""
    import random
    import struct
    import pytest
    from kleenexlang_amd import EngineError, compile_file, host
    for name in ("flip_ab", "csv2json", "apache_log", "coder"):
        blob = host.compile_regex("([a-c]*|[0-9]+)(,|;)?x*") if name == "coder" else compile_file(name)   # (coder: a symbol-table section)
        host.validate_blob(blob)                         # the real thing passes
        step = max(1, len(blob) // 400)
        for cut in list(range(0, 200)) + list(range(200, len(blob) - 1, step)):
            with pytest.raises(EngineError):
                host.validate_blob(blob[:cut])
        r = random.Random(7)
        hdr = 20 + ((struct.unpack_from("<I", blob, 16)[0] + 3) & ~3)
        refused = 0
        for _ in range(1500):
            b = bytearray(blob)
            off = hdr + (r.randrange(0, 16) * 4 if r.random() < 0.3 else r.randrange(0, len(blob) - hdr - 4) & ~3)
            struct.pack_into("<I", b, off, r.choice([0xFFFFFFFF, 0x7FFFFFFF, 0x10000, r.getrandbits(32)]))
            try:
                host.validate_blob(bytes(b))
            except EngineError:
                refused += 1
        assert refused > 100, refused
        # single bytes the 32-bit mutations do not reach: a byte class beyond the class count, a state without
        # leaves, a final leaf beyond the state's leaves — each would index past a table on the device
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import kxp
        st = kxp.parse(blob)[0]
        cls_off = hdr + 64
        b = bytearray(blob); b[cls_off + ord("z")] = st.nclasses
        with pytest.raises(EngineError, match="byte class"):
            host.validate_blob(bytes(b))
        sc = st.nstates * st.nclasses
        pad4 = lambda n: (n + 3) & ~3
        nact = struct.unpack_from("<I", blob, hdr + 5 * 4)[0]; nops = struct.unpack_from("<I", blob, hdr + 6 * 4)[0]
        nconsts = struct.unpack_from("<I", blob, hdr + 7 * 4)[0]; cpl = struct.unpack_from("<I", blob, hdr + 8 * 4)[0]
        nl_off = cls_off + 256 + pad4(sc * 2) + sc * 4 + st.nstates * 4 + (nact + 1) * 4 + nops * 8 + (nconsts + 1) * 4 + pad4(cpl) + sc * 4
        assert bytes(blob[nl_off:nl_off + st.nstates]) == bytes(st.nleaves)      # (the offsets above are the format's)
        b = bytearray(blob); b[nl_off + st.q0] = 0
        with pytest.raises(EngineError, match="leaf count"):
            host.validate_blob(bytes(b))
        if st.tables is not None:   # table field of a back entry: beyond the tables / on an entry that does not copy
            nback = st.back.shape[0]
            back_off = nl_off + 2 * pad4(st.nstates)
            assert bytes(blob[back_off:back_off + 4 * nback * st.maxleaves]) == st.back.tobytes()
            live = [i for i, e in enumerate(st.back.reshape(-1)) if e != 0xFFFFFFFF]
            with_tb = [i for i in live if int(st.back.reshape(-1)[i]) >> 24]
            assert with_tb
            b = bytearray(blob); struct.pack_into("<I", b, back_off + 4 * with_tb[0], (int(st.back.reshape(-1)[with_tb[0]]) & 0xFFFFFF) | (len(st.tables) + 1) << 24)
            with pytest.raises(EngineError, match="backward entry"):
                host.validate_blob(bytes(b))
            b = bytearray(blob); struct.pack_into("<I", b, back_off + 4 * with_tb[0], int(st.back.reshape(-1)[with_tb[0]]) & ~0x100)
            with pytest.raises(EngineError, match="backward entry"):
                host.validate_blob(bytes(b))
            with pytest.raises(EngineError, match="symbol tables"):
                host.validate_blob(blob[:len(blob) - 100])
        fin = [q for q in range(st.nstates) if st.fin_leaf[q] != 0xFF]
        if fin:
            b = bytearray(blob); b[nl_off + pad4(st.nstates) + fin[0]] = st.nleaves[fin[0]]
            with pytest.raises(EngineError, match="leaf count"):
                host.validate_blob(bytes(b))


def test_produced_binary_refuses_a_corrupt_trailer(tmp_path):
    import subprocess as sp
    from kleenexlang_amd import program_path
    exe = tmp_path / "flip"
    assert sp.run([os.path.join(build.OUT, "kexc"), "compile", "--quiet", program_path("flip_ab"), "--out", str(exe)]).returncode == 0
    data = bytearray(exe.read_bytes())
    data[-24:-16] = (1 << 40).to_bytes(8, "little")      # blob length far beyond the file
    bad = tmp_path / "bad"
    bad.write_bytes(bytes(data)); bad.chmod(0o755)
    r = sp.run([str(bad), "-i"], stdout=sp.PIPE, stderr=sp.PIPE)
    assert r.returncode == 1 and b"corrupt" in r.stderr


def test_big_table_programs_pass_validation():
    """A make_danish-sized program (state table > 100 KiB) is inside the engine's limits since round 2 (image in global
    memory, DevTables::big); one beyond 256 KiB of state table is still refused with a message."""
    import pytest
    from conftest import blob_of, dictionary_program
    from kleenexlang_amd import host
    from oracle import oracle
    src, _ = dictionary_program()
    blob = blob_of(src, opt=0)
    info = oracle.info(blob)
    assert 256 + (info["nstates"] + 1) * info["nclasses"] * 4 > 100 * 1024
    host.validate_blob(blob)
    src, _ = dictionary_program(nwords=100, lo=25, hi=32)
    # round 4: the compiler itself gives up as soon as the states and classes found so far prove the table too large …
    with pytest.raises(host.CompileError, match="outside engine limits: more than [0-9]+ SST states x [0-9]+ byte classes"):
        host.compile_source(src, opt=0)
    # … and a blob that reaches the engine anyway (a front end without that check) is refused at load
    os.environ["KEXC_NO_TABLE_CAP"] = "1"
    try:
        blob = host.compile_source(src, opt=0)
    finally:
        del os.environ["KEXC_NO_TABLE_CAP"]
    info = oracle.info(blob)
    assert 256 + (info["nstates"] + 1) * info["nclasses"] * 4 > 256 * 1024
    with pytest.raises(host.EngineError, match="outside engine limits"):
        host.validate_blob(blob)


def test_kx_config_mirrors_the_header_and_the_library_reads_no_switch_from_the_environment():
    """Round 6: the engine's switches are fields of kx_config.  (1) The ctypes mirror has the header's fields in the
    header's order; (2) the library's sources call getenv only for KX_DEBUG / KX_FD_TRACE / KX_WINDOW_BYTES / KX_READ_THREADS;
    (3) the old variable names map onto the struct (host.config_from_env, the test binding's twin of kxrun.cpp's configFromEnv);
    (4) load-time fields act on the host-only table builder: the delayed form can be switched off, its delay and window pinned."""
    from kleenexlang_amd import compile_file, host
    txt = open(os.path.join(INC, "kxhip.h")).read()
    body = re.search(r"typedef struct kx_config \{(.*?)\} kx_config;", txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:uint32_t|uint64_t)\s+([a-z_0-9]+)(?:\[\d+\])?\s*;", body)
    assert fields == [f[0] for f in host.KxConfig._fields_], (fields, [f[0] for f in host.KxConfig._fields_])
    eng = os.path.join(build.CSRC, "engine")
    seen = set()
    for name in ("kx_engine.hip", "kx_dfkernels.inc", "kx_delayed.h", "kx_sharded.cpp"):
        seen |= set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', open(os.path.join(eng, name)).read()))
    assert seen <= {"KX_DEBUG", "KX_FD_TRACE", "KX_WINDOW_BYTES", "KX_READ_THREADS"}, seen
    c = host.config_from_env({"KX_DF": "0", "KX_DF_K": "2", "KX_DF_J": "3", "KX_INL": "1", "KX_JL": "0", "KX_NO_PAIR": "1", "KX_FORCE_BIG": "1",
                              "KX_EMIT_WAVES": "8", "KX_EMIT_HALF": "0", "KX_DEBUG_FLAGS": "64", "KX_ACT_CHUNK": "2048", "KX_NO_SLOW": "1"})
    assert (c.delayed_form, c.delay, c.merge_window, c.inline_consts, c.job_stride, c.disable, c.force, c.emit_waves, c.emit_half, c.debug_flags,
            c.act_chunk) == (1, 2, 4, 2, 1, host.KX_OFF_PAIR | host.KX_OFF_SLOW, host.KX_FORCE_BIG, 8, 1, 64, 2048)
    blob = compile_file("apache_log")
    info, _ = host.df_describe(blob, with_image=False, cfg=host.KxConfig())
    assert info.available == 1 and info.delay == 2 and info.merge_window >= 1            # defaults: the engine picks delay and window
    info, _ = host.df_describe(blob, with_image=False, cfg=host.config_from_env({}, delayed_form=1))
    assert info.available == 0 and b"switched off" in info.reason
    info, _ = host.df_describe(blob, with_image=False, cfg=host.config_from_env({}, delay=2, merge_window=1))
    assert info.available == 1 and info.merge_window == 0
    info, _ = host.df_describe(blob, with_image=False, cfg=host.config_from_env({}, delay=2, merge_window=7))
    assert info.merge_window == 6
