"""ctypes bindings over the two C ABIs (include/kexc_api.h, include/kxhip.h).

Python here is plumbing only: it loads the in-tree shared libraries, moves
pointers around and mirrors the error behaviour of the reference's produced
binaries (``Match error at input symbol N!``, exit code 1 — Backends/C.hs:79-81).
There is no CPU fallback: if ``libkxhip.so`` is missing or no HIP device is
present, constructing a :class:`Program` raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
BUILD_DIR = os.path.join(_HERE, "_build")
PROGRAM_DIR = os.path.join(_HERE, "programs")

KX_NKERNELS = 6
KERNEL_NAMES = ("sync", "forward", "head", "backlen", "resolve", "emit")
KX_MAX_LEAVES = 256
NOFAIL = 0xFFFFFFFFFFFFFFFF


class KleenexError(RuntimeError):
    pass


class CompileError(KleenexError):
    pass


class EngineError(KleenexError):
    pass


class MatchError(KleenexError):
    """The input is not in the program's language (reference: exit 1 + stderr message)."""

    def __init__(self, pos, stage=0):
        super().__init__("Match error at input symbol %d!" % pos)
        self.pos = pos
        self.stage = stage


class KxStats(ctypes.Structure):
    _fields_ = [("fail_pos", ctypes.c_uint64), ("fail_stage", ctypes.c_uint32),
                ("unsynced_segments", ctypes.c_uint32), ("in_bytes", ctypes.c_uint64),
                ("out_bytes", ctypes.c_uint64), ("kernel_ms", ctypes.c_float * KX_NKERNELS),
                ("total_ms", ctypes.c_float), ("emit_overflow_pieces", ctypes.c_uint32)]

    def as_dict(self):
        return {"unsynced_segments": self.unsynced_segments, "in_bytes": self.in_bytes,
                "out_bytes": self.out_bytes, "total_ms": self.total_ms,
                "emit_overflow_pieces": self.emit_overflow_pieces,
                "kernel_ms": {k: self.kernel_ms[i] for i, k in enumerate(KERNEL_NAMES)}}


class KxShardedResult(ctypes.Structure):
    _fields_ = [("out_len", ctypes.c_uint64), ("out_offset", ctypes.c_uint64), ("total_out", ctypes.c_uint64),
                ("boundary_ms", ctypes.c_float), ("stats", KxStats)]


ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


class KxConfig(ctypes.Structure):
    """include/kxhip.h::kx_config (0 = default in every field)."""
    _fields_ = [("segment_bytes", ctypes.c_uint32), ("block_threads", ctypes.c_uint32),
                ("collect_timing", ctypes.c_uint32), ("phase", ctypes.c_uint32), ("window_bytes", ctypes.c_uint64)] + \
               [(k, ctypes.c_uint32) for k in ("delayed_form", "delay", "merge_window", "inline_consts", "job_stride", "disable", "force",
                                               "emit_waves", "emit_half", "emit_inplace", "emit_staging", "df_backoff", "debug_flags",
                                               "act_par_min", "act_prefix3_min", "act_lanes", "act_chunk")] + \
               [("reserved", ctypes.c_uint32 * 4)]


KX_OFF_DIRECT, KX_OFF_PAIR, KX_OFF_CMPX, KX_OFF_COOP, KX_OFF_SLOW = 1, 2, 4, 8, 16
KX_FORCE_BIG, KX_FORCE_TBLMODE, KX_FORCE_ACT_SEQ, KX_FORCE_SAME_DEVICE = 1, 2, 4, 8


def config_from_env(env=None, **fields):
    """The engine's switches are fields of kx_config; the library reads no environment variable for them.  Tests and profiling
    scripts keep using the variable names of earlier rounds: they are read HERE (and in kxrun.cpp for produced binaries) and mapped
    onto the struct.  `fields` override."""
    env = os.environ if env is None else env
    c = KxConfig()

    def tri(name):      # unset: auto, 0: off, anything else: on
        return 0 if name not in env else (2 if int(env[name]) else 1)

    def num(name):
        return int(env.get(name, "0") or 0)

    if "KX_DF" in env:
        v = int(env["KX_DF"])
        c.delayed_form = 1 if v == 0 else 2 if v == 2 else 0
    c.delay = num("KX_DF_K")
    if "KX_DF_J" in env:
        c.merge_window = int(env["KX_DF_J"]) + 1
    c.inline_consts = tri("KX_INL")
    c.job_stride = tri("KX_JL") if "KX_JL" in env else (1 if "KX_JL_AUTO_OFF" in env else 0)
    for name, bit in (("KX_NO_DIRECT", KX_OFF_DIRECT), ("KX_NO_PAIR", KX_OFF_PAIR), ("KX_NO_CMPX", KX_OFF_CMPX), ("KX_NO_COOP", KX_OFF_COOP),
                      ("KX_NO_SLOW", KX_OFF_SLOW)):
        if name in env:
            c.disable |= bit
    for name, bit in (("KX_FORCE_BIG", KX_FORCE_BIG), ("KX_FORCE_TBLMODE", KX_FORCE_TBLMODE), ("KX_ACT_SEQ", KX_FORCE_ACT_SEQ),
                      ("KX_SHARD_SAME_DEVICE", KX_FORCE_SAME_DEVICE)):
        if name in env:
            c.force |= bit
    c.emit_waves = num("KX_EMIT_WAVES")
    c.emit_half = tri("KX_EMIT_HALF")
    c.emit_inplace = tri("KX_EMIT_INPLACE")
    c.emit_staging = num("KX_EMIT_STG")
    c.df_backoff = 1 if num("KX_DF_BACKOFF_OFF") else 0
    c.debug_flags = num("KX_DEBUG_FLAGS")
    c.act_par_min = num("KX_ACT_PAR_MIN")
    c.act_prefix3_min = num("KX_ACT_PREFIX3_MIN")
    c.act_lanes = tri("KX_ACT_LANES")
    c.act_chunk = num("KX_ACT_CHUNK")
    for k, v in fields.items():
        setattr(c, k, v)
    return c


class KxDfInfo(ctypes.Structure):
    """include/kxhip.h::kx_df_info — the delayed form of one stage."""
    _fields_ = [(k, ctypes.c_uint32) for k in ("available", "delay", "nstates", "nclasses", "image_bytes", "off_pool", "start_handle",
                                               "dead_handle", "escape_handle", "transitions", "escapes", "transitions_start",
                                               "escapes_start", "merge_window")] + [("reason", ctypes.c_char * 96)]


class KxFwdSummary(ctypes.Structure):
    _fields_ = [("synced", ctypes.c_uint32), ("end_state", ctypes.c_uint32),
                ("head_len", ctypes.c_uint64), ("fail_pos", ctypes.c_uint64)]


class KxBwdSummary(ctypes.Structure):
    _fields_ = [("constant", ctypes.c_uint32), ("nleaves", ctypes.c_uint32),
                ("start_leaf", ctypes.c_uint8 * KX_MAX_LEAVES)]


_kexc = None
_kxhip = None


def _lib_path(name):
    return os.path.join(BUILD_DIR, name)


def load_compiler():
    """dlopen libkexc.so (built by ``kleenexlang_amd.build`` / ``__graft_entry__.build``)."""
    global _kexc
    if _kexc is None:
        path = _lib_path("libkexc.so")
        if not os.path.exists(path):
            raise CompileError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        lib = ctypes.CDLL(path)
        lib.kexc_compile.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        lib.kexc_emit_c.argtypes = lib.kexc_compile.argtypes
        lib.kexc_last_error.restype = ctypes.c_char_p
        lib.kexc_free.argtypes = [ctypes.c_void_p]
        _kexc = lib
    return _kexc


def load_engine():
    """dlopen libkxhip.so; raises if it has not been built (no fallback)."""
    global _kxhip
    if _kxhip is None:
        path = _lib_path("libkxhip.so")
        if not os.path.exists(path):
            raise EngineError("HIP engine %s is missing: build it with __graft_entry__.build(); "
                              "there is no CPU fallback" % path)
        try:
            # torch wheels bundle their own libamdhip64.so.7; /opt/rocm ships one with the same soname.
            # Whichever is mapped first serves the whole process, and torch cannot initialise on the
            # other one — so inside a torch process let torch's runtime load first.  (C consumers such
            # as kxrun simply get the system runtime.)
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(path)
        vp, sz, u32, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint64
        lib.kx_load.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(vp)]
        lib.kx_load_config.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(KxConfig), ctypes.POINTER(vp)]
        lib.kx_stage_reset_delayed_form.argtypes = [vp, u32]
        lib.kx_stage_reset_delayed_form.restype = None
        lib.kx_validate.argtypes = [ctypes.c_char_p, sz]
        lib.kx_free.argtypes = [vp]
        lib.kx_last_error.restype = ctypes.c_char_p
        lib.kx_set_config.argtypes = [vp, ctypes.POINTER(KxConfig)]
        lib.kx_num_stages.argtypes = [vp]
        lib.kx_num_stages.restype = u32
        lib.kx_stage_has_actions.argtypes = [vp, u32]
        lib.kx_stage_delayed_form.argtypes = [vp, u32]
        lib.kx_df_describe.argtypes = [ctypes.c_char_p, sz, u32, ctypes.POINTER(KxDfInfo), vp, sz]
        lib.kx_df_pending.argtypes = [ctypes.c_char_p, sz, u32, u32, u32, ctypes.POINTER(u32), ctypes.POINTER(u32), ctypes.POINTER(u32)]
        cfgp = ctypes.POINTER(KxConfig)
        lib.kx_df_describe_cfg.argtypes = [ctypes.c_char_p, sz, u32, cfgp, ctypes.POINTER(KxDfInfo), vp, sz]
        lib.kx_df_pending_cfg.argtypes = [ctypes.c_char_p, sz, u32, cfgp, u32, u32, ctypes.POINTER(u32), ctypes.POINTER(u32), ctypes.POINTER(u32)]
        lib.kx_df_deferred.argtypes = [ctypes.c_char_p, sz, u32, cfgp, u32, ctypes.c_void_p, sz, ctypes.POINTER(sz)]
        lib.kx_df_start_of_state.argtypes = [ctypes.c_char_p, sz, u32, cfgp, u32, ctypes.POINTER(u32)]
        lib.kx_run_device.argtypes = [vp, vp, sz, vp, sz, ctypes.POINTER(sz), ctypes.POINTER(KxStats), vp]
        lib.kx_run_host.argtypes = [vp, ctypes.c_char_p, sz, ctypes.POINTER(vp), ctypes.POINTER(sz),
                                    ctypes.POINTER(KxStats)]
        lib.kx_host_free.argtypes = [vp]
        lib.kx_run_fd.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(KxStats)]
        lib.kx_shard_begin.argtypes = [vp, u32, vp, sz, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]
        lib.kx_shard_forward.argtypes = [vp, ctypes.POINTER(KxFwdSummary)]
        lib.kx_shard_fix_head.argtypes = [vp, u32, ctypes.POINTER(KxFwdSummary)]
        lib.kx_shard_backward.argtypes = [vp, ctypes.POINTER(KxBwdSummary)]
        lib.kx_shard_resolve.argtypes = [vp, u32, ctypes.POINTER(u64)]
        lib.kx_shard_emit.argtypes = [vp, vp, sz]
        lib.kx_shard_stats.argtypes = [vp, ctypes.POINTER(KxStats)]
        lib.kx_shard_end.argtypes = [vp]
        lib.kx_run_sharded.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, sz, vp, sz, ctypes.POINTER(KxShardedResult), vp]
        lib.kx_comm_unique_id.argtypes = [ctypes.c_char_p]
        lib.kx_comm_init.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
        lib.kx_comm_free.argtypes = [vp]
        lib.kx_group_create.argtypes = [ctypes.c_int]
        lib.kx_group_create.restype = vp
        lib.kx_group_free.argtypes = [vp]
        lib.kx_group_join.argtypes = [vp, ctypes.c_int]
        lib.kx_group_join.restype = vp
        lib.kx_group_leave.argtypes = [vp]
        _kxhip = lib
    return _kxhip


# ------------------------------------------------------------------ compile
def compile_source(source, name="<memory>", opt=3):
    """Kleenex source text → KXP blob (bytes).  Mirrors ``kexc compile --opt N`` (direct mode)."""
    lib = load_compiler()
    if isinstance(source, str):
        source = source.encode("utf-8")
    blob = ctypes.c_void_p()
    n = ctypes.c_size_t()
    rc = lib.kexc_compile(source, len(source), name.encode(), opt, ctypes.byref(blob), ctypes.byref(n))
    if rc:
        raise CompileError(lib.kexc_last_error().decode("utf-8", "replace"))
    try:
        return ctypes.string_at(blob, n.value)
    finally:
        lib.kexc_free(blob)


def compile_flags(source, name="<memory>", opt=3, la=False, regex=False):
    """``kexc compile --opt N --la=BOOL`` (regex=True: ``--re``) → KXP blob.  la=True builds the reference's lookahead machine
    (word tests) first and the tables from its path form (include/kexc_api.h::kexc_compile_flags)."""
    lib = load_compiler()
    if isinstance(source, str):
        source = source.encode("utf-8")
    blob = ctypes.c_void_p()
    n = ctypes.c_size_t()
    lib.kexc_compile_flags.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    rc = lib.kexc_compile_flags(source, len(source), name.encode(), opt, 1 if la else 0, 1 if regex else 0, ctypes.byref(blob), ctypes.byref(n))
    if rc:
        raise CompileError(lib.kexc_last_error().decode("utf-8", "replace"))
    try:
        return ctypes.string_at(blob, n.value)
    finally:
        lib.kexc_free(blob)


def dump_words(source, name="<memory>", regex=False):
    """JSON-decoded lookahead machines (``--la=true``) of a program's stages in path form (test support)."""
    import json
    lib = load_compiler()
    if isinstance(source, str):
        source = source.encode("utf-8")
    txt = ctypes.c_void_p()
    n = ctypes.c_size_t()
    lib.kexc_dump_words.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                    ctypes.POINTER(ctypes.c_size_t)]
    rc = lib.kexc_dump_words(source, len(source), name.encode(), 1 if regex else 0, ctypes.byref(txt), ctypes.byref(n))
    if rc:
        raise CompileError(lib.kexc_last_error().decode("utf-8", "replace"))
    try:
        return json.loads(ctypes.string_at(txt, n.value).decode("utf-8"))
    finally:
        lib.kexc_free(txt)


def emit_c(source, name="<memory>", opt=3):
    """``--backend=c``: reference-shaped C text for the CPU baseline."""
    lib = load_compiler()
    if isinstance(source, str):
        source = source.encode("utf-8")
    txt = ctypes.c_void_p()
    n = ctypes.c_size_t()
    rc = lib.kexc_emit_c(source, len(source), name.encode(), opt, ctypes.byref(txt), ctypes.byref(n))
    if rc:
        raise CompileError(lib.kexc_last_error().decode("utf-8", "replace"))
    try:
        return ctypes.string_at(txt, n.value).decode("utf-8")
    finally:
        lib.kexc_free(txt)


def dump_fst(source, name="<memory>"):
    """JSON-decoded nondeterministic transducers of a program (test support)."""
    import json
    lib = load_compiler()
    if isinstance(source, str):
        source = source.encode("utf-8")
    txt = ctypes.c_void_p()
    n = ctypes.c_size_t()
    lib.kexc_dump_fst.argtypes = lib.kexc_compile.argtypes[:3] + lib.kexc_compile.argtypes[4:]
    rc = lib.kexc_dump_fst(source, len(source), name.encode(), ctypes.byref(txt), ctypes.byref(n))
    if rc:
        raise CompileError(lib.kexc_last_error().decode("utf-8", "replace"))
    try:
        return json.loads(ctypes.string_at(txt, n.value).decode("utf-8"))
    finally:
        lib.kexc_free(txt)


def compile_regex(regex, name="<command line>", opt=3):
    """Regular expression → KXP blob of its bit-coder (``kexc compile --re EXPR`` / ``FILE.re``; Commands.hs:246-275)."""
    lib = load_compiler()
    if isinstance(regex, str):
        regex = regex.encode("utf-8")
    blob = ctypes.c_void_p()
    n = ctypes.c_size_t()
    lib.kexc_compile_regex.argtypes = lib.kexc_compile.argtypes
    rc = lib.kexc_compile_regex(regex, len(regex), name.encode(), opt, ctypes.byref(blob), ctypes.byref(n))
    if rc:
        raise CompileError(lib.kexc_last_error().decode("utf-8", "replace"))
    try:
        return ctypes.string_at(blob, n.value)
    finally:
        lib.kexc_free(blob)


def dump_regex_fst(regex, oracle=True):
    """JSON-decoded transducer of a regex program: its oracle machine, or (oracle=False) the transducer before `oracle`."""
    import json
    lib = load_compiler()
    if isinstance(regex, str):
        regex = regex.encode("utf-8")
    txt = ctypes.c_void_p()
    n = ctypes.c_size_t()
    lib.kexc_dump_regex_fst.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.POINTER(ctypes.c_size_t)]
    rc = lib.kexc_dump_regex_fst(regex, len(regex), 1 if oracle else 0, ctypes.byref(txt), ctypes.byref(n))
    if rc:
        raise CompileError(lib.kexc_last_error().decode("utf-8", "replace"))
    try:
        return json.loads(ctypes.string_at(txt, n.value).decode("utf-8"))
    finally:
        lib.kexc_free(txt)


def validate_blob(blob):
    """Structural check of a KXP blob (no device needed).  Raises EngineError with the engine's message."""
    lib = load_engine()
    blob = bytes(blob)
    if lib.kx_validate(blob, len(blob)):
        raise EngineError(lib.kx_last_error().decode("utf-8", "replace"))


def df_describe(blob, stage=0, with_image=True, cfg=None):
    """The delayed form of a stage, built on the host (no device needed): (KxDfInfo, table image bytes or None).
    cfg: a KxConfig (its load-time fields); None = what the environment's KX_DF* variables say (config_from_env)."""
    lib = load_engine()
    blob = bytes(blob)
    info = KxDfInfo()
    cfg = config_from_env() if cfg is None else cfg
    if lib.kx_df_describe_cfg(blob, len(blob), stage, ctypes.byref(cfg), ctypes.byref(info), None, 0):
        raise EngineError(lib.kx_last_error().decode("utf-8", "replace"))
    img = None
    if with_image and info.image_bytes:
        buf = ctypes.create_string_buffer(info.image_bytes)
        lib.kx_df_describe_cfg(blob, len(blob), stage, ctypes.byref(cfg), ctypes.byref(info), buf, info.image_bytes)
        img = buf.raw
    return info, img


def df_pending(blob, stage, state, slot, cfg=None):
    """(SST state, [copy | path-constant id << 1 per leaf, or one value]) of a product state's pending slot."""
    lib = load_engine()
    blob = bytes(blob)
    kinds = (ctypes.c_uint32 * KX_MAX_LEAVES)()
    n = ctypes.c_uint32()
    q = ctypes.c_uint32()
    cfg = config_from_env() if cfg is None else cfg
    if lib.kx_df_pending_cfg(blob, len(blob), stage, ctypes.byref(cfg), state, slot, ctypes.byref(q), kinds, ctypes.byref(n)):
        raise EngineError(lib.kx_last_error().decode("utf-8", "replace"))
    return q.value, list(kinds[:n.value])


def df_deferred(blob, stage, state, cfg=None):
    """The constants a product state still has due (merged constants, kx_delayed.h): their bytes, oldest first."""
    lib = load_engine()
    blob = bytes(blob)
    buf = ctypes.create_string_buffer(1024)
    n = ctypes.c_size_t()
    cfg = config_from_env() if cfg is None else cfg
    if lib.kx_df_deferred(blob, len(blob), stage, ctypes.byref(cfg), state, buf, 1024, ctypes.byref(n)):
        raise EngineError(lib.kx_last_error().decode("utf-8", "replace"))
    return buf.raw[:n.value]


def df_start_of_state(blob, stage, sst_state, cfg=None):
    """Handle of (state, nothing pending, nothing due) in the delayed form's table, or 0xFFFF."""
    lib = load_engine()
    blob = bytes(blob)
    h = ctypes.c_uint32()
    cfg = config_from_env() if cfg is None else cfg
    if lib.kx_df_start_of_state(blob, len(blob), stage, ctypes.byref(cfg), sst_state, ctypes.byref(h)):
        raise EngineError(lib.kx_last_error().decode("utf-8", "replace"))
    return h.value


class KexcIlProgram(ctypes.Structure):
    """include/kexc_api.h::kexc_il_program — one IL Program in table form + the path-tree annotation."""
    _u8p, _u16p, _u32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint32)
    _fields_ = [("nstates", ctypes.c_uint32), ("nclasses", ctypes.c_uint32), ("init_state", ctypes.c_uint32), ("nregs", ctypes.c_uint32),
                ("class_of", _u8p), ("delta", _u16p), ("action", _u32p), ("final_action", _u32p),
                ("nactions", ctypes.c_uint32), ("action_off", _u32p), ("ops", _u32p),
                ("nconsts", ctypes.c_uint32), ("const_off", _u32p), ("const_pool", _u8p),
                ("maxleaves", ctypes.c_uint32), ("nback", ctypes.c_uint32), ("back_row", _u32p),
                ("nleaves", _u8p), ("final_leaf", _u8p), ("back", _u32p),
                ("npconsts", ctypes.c_uint32), ("pconst_off", _u32p), ("pconst_pool", _u8p), ("init_const", _u32p),
                ("has_actions", ctypes.c_uint32), ("action_regs", ctypes.c_uint32),
                ("ntables", ctypes.c_uint32), ("tbl_width", _u32p), ("tbl_data", _u8p), ("back_table", _u32p),
                ("ntests", ctypes.c_uint32), ("test_block", _u32p), ("test_target", _u32p), ("test_len", _u32p),
                ("test_preds", _u8p), ("test_back", _u32p)]


class KexcPipeline(ctypes.Structure):
    _fields_ = [("is_oracle_action", ctypes.c_int), ("nprograms", ctypes.c_uint32), ("programs", ctypes.POINTER(KexcIlProgram)),
                ("program_size", ctypes.c_uint32)]


def emit_pipeline(programs, env_info=None, out=None, srcout=None, buffer_unit_bits=8, copt=3, cc="cc", word_alignment=True,
                  oracle_action=False, info=None, program_size=None):
    """compileProgram's seam (include/kexc_api.h::kexc_emit_pipeline): `programs` is a list of dicts of numpy arrays
    (keys = the fields of kexc_il_program).  Returns the exit code; raises CompileError with the message on failure."""
    import numpy as np
    lib = load_compiler()
    keep, structs = [], (KexcIlProgram * len(programs))()
    kinds = {"class_of": np.uint8, "delta": np.uint16, "action": np.uint32, "final_action": np.uint32, "action_off": np.uint32, "ops": np.uint32,
             "const_off": np.uint32, "const_pool": np.uint8, "back_row": np.uint32, "nleaves": np.uint8, "final_leaf": np.uint8,
             "back": np.uint32, "pconst_off": np.uint32, "pconst_pool": np.uint8, "init_const": np.uint32}
    def ptr(a, dt):
        return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8 if dt == np.uint8 else ctypes.c_uint16 if dt == np.uint16 else ctypes.c_uint32))
    for st, P in zip(structs, programs):
        st.ntests = int(P.get("ntests", 0))
        if st.ntests:      # block form (--la=true): only the annotation travels, the class tables stay NULL
            for k in ("nstates", "init_state", "maxleaves", "npconsts"):
                setattr(st, k, int(P[k]))
            st.has_actions, st.action_regs = int(P.get("has_actions", 0)), int(P.get("action_regs", 0))
            for k, dt in (("nleaves", np.uint8), ("final_leaf", np.uint8), ("pconst_off", np.uint32), ("pconst_pool", np.uint8), ("init_const", np.uint32),
                          ("test_block", np.uint32), ("test_target", np.uint32), ("test_len", np.uint32), ("test_preds", np.uint8), ("test_back", np.uint32)):
                a = np.ascontiguousarray(np.asarray(P[k], dtype=dt).ravel())
                if a.size == 0:
                    a = np.zeros(1, dtype=dt)
                keep.append(a)
                setattr(st, k, ptr(a, dt))
            continue
        for k in ("nstates", "nclasses", "init_state", "nregs", "nactions", "nconsts", "maxleaves", "nback", "npconsts"):
            setattr(st, k, int(P[k]))
        st.has_actions, st.action_regs = int(P.get("has_actions", 0)), int(P.get("action_regs", 0))
        st.ntables = int(P.get("ntables", 0))
        for k, dt in (("tbl_width", np.uint32), ("tbl_data", np.uint8), ("back_table", np.uint32)) if st.ntables else ():
            a = np.ascontiguousarray(np.asarray(P[k], dtype=dt).ravel())
            keep.append(a)
            setattr(st, k, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8 if dt == np.uint8 else ctypes.c_uint32)))
        for k, dt in kinds.items():
            a = np.ascontiguousarray(np.asarray(P[k], dtype=dt).ravel())
            if a.size == 0:
                a = np.zeros(1, dtype=dt)
            keep.append(a)
            setattr(st, k, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8 if dt == np.uint8 else ctypes.c_uint16 if dt == np.uint16 else ctypes.c_uint32)))
    pl = KexcPipeline(1 if oracle_action else 0, len(programs), structs, ctypes.sizeof(KexcIlProgram) if program_size is None else program_size)
    CB = ctypes.CFUNCTYPE(None, ctypes.c_char_p, ctypes.c_void_p)
    cb = CB((lambda line, ctx: info(line.decode())) if info else (lambda line, ctx: None))
    lib.kexc_emit_pipeline_v2.argtypes = [ctypes.c_int, ctypes.c_int, CB, ctypes.c_void_p, ctypes.POINTER(KexcPipeline), ctypes.c_char_p,
                                       ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    enc = lambda x: None if x is None else str(x).encode()
    rc = lib.kexc_emit_pipeline_v2(buffer_unit_bits, copt, cb, None, ctypes.byref(pl), enc(env_info), enc(cc), enc(out), enc(srcout), 1 if word_alignment else 0)
    if rc:
        raise CompileError(lib.kexc_last_error().decode("utf-8", "replace"))
    return rc


def program_path(name):
    p = os.path.join(PROGRAM_DIR, name if name.endswith(".kex") else name + ".kex")
    if not os.path.exists(p):
        raise FileNotFoundError(p)
    return p


def compile_file(path, opt=3):
    if not os.path.exists(path):
        path = program_path(path)
    with open(path, "rb") as f:
        return compile_source(f.read(), os.path.basename(path), opt)


# ------------------------------------------------------------------- engine
class Program:
    """A compiled Kleenex program loaded on the current HIP device."""

    def __init__(self, blob, segment_bytes=0, block_threads=0, collect_timing=False, window_bytes=0, config=None):
        """config: a KxConfig; None = the environment's KX_* variables mapped onto one (config_from_env) — the library itself reads
        none of them."""
        self._lib = load_engine()
        self._h = ctypes.c_void_p()
        self._blob = bytes(blob)
        self._cfg = config_from_env() if config is None else config
        self._cfg.segment_bytes, self._cfg.block_threads = segment_bytes, block_threads
        self._cfg.collect_timing, self._cfg.window_bytes = 1 if collect_timing else 0, window_bytes
        rc = self._lib.kx_load_config(self._blob, len(self._blob), ctypes.byref(self._cfg), ctypes.byref(self._h))
        if rc:
            raise EngineError(self._err())
        self.last_stats = None

    @classmethod
    def from_file(cls, path, opt=3, **kw):
        return cls(compile_file(path, opt), **kw)

    @classmethod
    def from_source(cls, source, opt=3, **kw):
        return cls(compile_source(source, opt=opt), **kw)

    def _err(self):
        return self._lib.kx_last_error().decode("utf-8", "replace")

    def configure(self, segment_bytes=0, block_threads=0, collect_timing=False, window_bytes=0, **fields):
        """Run-time fields (kx_set_config); a load-time field in `fields` that differs from the loaded program's is refused."""
        cfg = self._cfg
        cfg.segment_bytes, cfg.block_threads = segment_bytes, block_threads
        cfg.collect_timing, cfg.window_bytes = 1 if collect_timing else 0, window_bytes
        for k, v in fields.items():
            setattr(cfg, k, v)
        if self._lib.kx_set_config(self._h, ctypes.byref(cfg)):
            raise EngineError(self._err())

    def reset_delayed_form(self, stage=0):
        self._lib.kx_stage_reset_delayed_form(self._h, stage)

    @property
    def num_stages(self):
        return self._lib.kx_num_stages(self._h)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.kx_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, stats):
        self.last_stats = stats
        if rc == 1:
            raise MatchError(stats.fail_pos, stats.fail_stage)
        if rc:
            raise EngineError(self._err())

    def run_host(self, data):
        """bytes → bytes through H2D / engine / D2H."""
        data = bytes(data)
        out = ctypes.c_void_p()
        n = ctypes.c_size_t()
        stats = KxStats()
        rc = self._lib.kx_run_host(self._h, data, len(data), ctypes.byref(out), ctypes.byref(n), ctypes.byref(stats))
        try:
            self._check(rc, stats)
            return ctypes.string_at(out, n.value)
        finally:
            if out.value:
                self._lib.kx_host_free(out)

    def run_device(self, d_in, n, d_out, cap, stream=None):
        """Raw device pointers (ints).  Returns the output length; raises MatchError on rejection."""
        ol = ctypes.c_size_t()
        stats = KxStats()
        rc = self._lib.kx_run_device(self._h, ctypes.c_void_p(d_in), n, ctypes.c_void_p(d_out), cap,
                                     ctypes.byref(ol), ctypes.byref(stats), ctypes.c_void_p(stream or 0))
        if rc == -3:
            self.last_stats = stats
            raise EngineError("output buffer too small: need %d bytes" % ol.value)
        self._check(rc, stats)
        return ol.value

    def run_tensor(self, t, out=None):
        """torch uint8 CUDA tensor → torch uint8 CUDA tensor (device-resident both ends)."""
        import torch
        assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()
        n = t.numel()
        if out is None:
            out = torch.empty(self.out_capacity(n), dtype=torch.uint8, device=t.device)
        stream = torch.cuda.current_stream(t.device).cuda_stream
        ol = self.run_device(t.data_ptr(), n, out.data_ptr(), out.numel(), stream)
        return out[:ol]

    def out_capacity(self, n, factor=None):
        """A generous output allocation for n input bytes (callers may also size exactly via shards)."""
        f = factor if factor is not None else 8
        return int(n * f) + 65536

    # sharded protocol (one shard per rank); thin wrappers, see include/kxhip.h
    def stage_has_actions(self, stage):
        return bool(self._lib.kx_stage_has_actions(self._h, stage))

    def stage_delayed_form(self, stage=0):
        """0: the stage has no delayed form; 1: it runs on it; 2: a shard gave it up (an undecided context), later shards run the general engine."""
        return int(self._lib.kx_stage_delayed_form(self._h, stage))

    def run_sharded(self, rank, world, gather, d_in, n, d_out, cap, stream=None):
        """kx_run_sharded: this rank's shard through every stage, the boundary hand-off done by the library through `gather`
        (a Comm, a GroupMember, or None when world == 1).  Returns the KxShardedResult; raises MatchError (global position)."""
        res = KxShardedResult()
        fn, ctx = (None, None) if gather is None else gather.callback()
        rc = self._lib.kx_run_sharded(self._h, rank, world, fn, ctx, ctypes.c_void_p(d_in), n, ctypes.c_void_p(d_out), cap,
                                      ctypes.byref(res), ctypes.c_void_p(stream or 0))
        if rc == -3:
            self.last_stats = res.stats
            raise EngineError("output buffer too small: need %d bytes" % res.out_len)
        self._check(rc, res.stats)
        return res

    def shard_begin(self, stage, d_in, n, is_first, is_last, stream=None):
        if self.stage_has_actions(stage):
            raise EngineError("stage %d uses register actions: their replay is sequential over the whole stream, "
                              "run the program unsharded (kx_run_device / kx_run_fd)" % stage)
        return Shard(self, stage, d_in, n, is_first, is_last, stream)


class Comm:
    """The library's own RCCL communicator (kx_comm_*): rank 0 makes the 128-byte id, the launcher distributes it."""

    def __init__(self, rank, world, unique_id=None):
        self._lib = load_engine()
        self._h = ctypes.c_void_p()
        rc = self._lib.kx_comm_init(ctypes.byref(self._h), rank, world, unique_id)
        if rc:
            raise EngineError(self._lib.kx_last_error().decode("utf-8", "replace"))

    @staticmethod
    def unique_id():
        lib = load_engine()
        buf = ctypes.create_string_buffer(128)
        if lib.kx_comm_unique_id(buf):
            raise EngineError(lib.kx_last_error().decode("utf-8", "replace"))
        return buf.raw

    def callback(self):
        return ctypes.cast(self._lib.kx_comm_allgather, ctypes.c_void_p), self._h

    def close(self):
        if self._h:
            self._lib.kx_comm_free(self._h)
            self._h = ctypes.c_void_p()


class Group:
    """Ranks = threads of this process (kx_group_*): what the produced binary's `--gpus N` uses."""

    def __init__(self, world):
        self._lib = load_engine()
        self._h = ctypes.c_void_p(self._lib.kx_group_create(world))
        self.world = world

    def member(self, rank):
        return GroupMember(self, rank)

    def close(self):
        if self._h:
            self._lib.kx_group_free(self._h)
            self._h = ctypes.c_void_p()


class GroupMember:
    def __init__(self, group, rank):
        self._lib = group._lib
        self._h = ctypes.c_void_p(self._lib.kx_group_join(group._h, rank))

    def callback(self):
        return ctypes.cast(self._lib.kx_group_allgather, ctypes.c_void_p), self._h

    def close(self):
        if self._h:
            self._lib.kx_group_leave(self._h)
            self._h = ctypes.c_void_p()


class Shard:
    def __init__(self, prog, stage, d_in, n, is_first, is_last, stream=None):
        self._p = prog
        self._lib = prog._lib
        self._h = ctypes.c_void_p()
        rc = self._lib.kx_shard_begin(prog._h, stage, ctypes.c_void_p(d_in), n, int(is_first), int(is_last),
                                      ctypes.c_void_p(stream or 0), ctypes.byref(self._h))
        if rc:
            raise EngineError(prog._err())

    def _ck(self, rc):
        if rc:
            raise EngineError(self._p._err())

    def forward(self):
        s = KxFwdSummary()
        self._ck(self._lib.kx_shard_forward(self._h, ctypes.byref(s)))
        return s

    def fix_head(self, incoming_state):
        s = KxFwdSummary()
        self._ck(self._lib.kx_shard_fix_head(self._h, incoming_state, ctypes.byref(s)))
        return s

    def backward(self):
        s = KxBwdSummary()
        self._ck(self._lib.kx_shard_backward(self._h, ctypes.byref(s)))
        return s

    def resolve(self, end_leaf):
        n = ctypes.c_uint64()
        self._ck(self._lib.kx_shard_resolve(self._h, end_leaf, ctypes.byref(n)))
        return n.value

    def emit(self, d_out, cap):
        self._ck(self._lib.kx_shard_emit(self._h, ctypes.c_void_p(d_out), cap))

    def stats(self):
        s = KxStats()
        self._lib.kx_shard_stats(self._h, ctypes.byref(s))
        return s

    def end(self):
        if self._h.value:
            self._lib.kx_shard_end(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.end()
        except Exception:
            pass
