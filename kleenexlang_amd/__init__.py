"""kleenexlang_amd — MI355X-native execution engine for compiled Kleenex programs.

Only the hot path of the reference (diku-kmc/kleenexlang) lives here: the compiled
streaming-string-transducer state loop and its output runtime, as hand-written HIP
for gfx950 behind a C ABI (include/kxhip.h), plus the compiler restatement needed to
obtain transducers at all (include/kexc_api.h).  See DESIGN.md.
"""
from .host import (CompileError, EngineError, KleenexError, MatchError, Program, compile_file,  # noqa: F401
                   compile_source, emit_c, emit_pipeline, program_path)

__all__ = ["Program", "compile_source", "compile_file", "emit_c", "emit_pipeline", "program_path",
           "KleenexError", "CompileError", "EngineError", "MatchError"]
