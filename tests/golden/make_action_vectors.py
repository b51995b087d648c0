#!/usr/bin/env python3
"""Writes tests/golden/action_vectors.json: the reference's known-answer vectors for programs with REGISTER ACTIONS
(`// IN:` / `// OUT:` headers of test/test_compiled/src/actionbug.kex and test/test_simulated/src/makeDanish.kex;
runtest.sh:17-37 joins the IN lines with newlines, appends one, and compares modulo trailing newlines).
Runs only where /root/reference is mounted; the vectors (program text + lines) are data, as in reference_vectors.json."""
import json
import os

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
out = []
for name, rel in (("actionbug", "test/test_compiled/src/actionbug.kex"), ("makeDanish", "test/test_simulated/src/makeDanish.kex")):
    ins, outs, prog = [], [], []
    for line in open(os.path.join(REF, rel), encoding="utf-8").read().split("\n"):
        if line.startswith("// IN:"):
            ins.append(line[6:])
        elif line.startswith("// OUT:"):
            outs.append(line[7:])
        else:
            prog.append(line)
    out.append({"name": name, "source": rel, "program": "\n".join(prog), "in": ins, "out": outs})
json.dump({"_comment": "Register-action vectors of the reference (see make_action_vectors.py)", "line_tests": out},
          open(os.path.join(HERE, "action_vectors.json"), "w"), ensure_ascii=False, indent=1)
print([(t["name"], t["in"], t["out"]) for t in out])
