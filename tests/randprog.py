"""Seeded random Kleenex programs and inputs (test support)."""
import random

ALPHA = "abc"


def _regex(r, depth):
    k = r.random()
    if depth <= 0 or k < 0.30:
        c = r.random()
        if c < 0.5:
            return r.choice(ALPHA)
        if c < 0.7:
            return "[%s]" % "".join(sorted(set(r.choice(ALPHA) for _ in range(2))))
        if c < 0.8:
            return "[^%s]" % r.choice(ALPHA)
        return "."
    if k < 0.50:
        return "%s%s" % (_regex(r, depth - 1), _regex(r, depth - 1))
    if k < 0.65:
        return "(%s|%s)" % (_regex(r, depth - 1), _regex(r, depth - 1))
    if k < 0.80:
        return "(%s)%s" % (_regex(r, depth - 1), r.choice(["*", "+", "?", "*?", "??"]))
    if k < 0.90:
        lo = r.randint(0, 2)
        return "(%s){%d,%d}" % (_regex(r, depth - 1), lo, lo + r.randint(0, 2))
    return "(%s){%d}" % (_regex(r, depth - 1), r.randint(1, 3))


def _term(r, depth):
    k = r.random()
    if depth <= 0 or k < 0.35:
        c = r.random()
        if c < 0.45:
            return "/%s/" % _regex(r, 2)
        if c < 0.65:
            return "~/%s/" % _regex(r, 2)
        if c < 0.95:
            return '"%s"' % "".join(r.choice("XYZ,") for _ in range(r.randint(1, 3)))
        return "1"
    if k < 0.60:
        return "%s %s" % (_term(r, depth - 1), _term(r, depth - 1))
    if k < 0.80:
        return "(%s | %s)" % (_term(r, depth - 1), _term(r, depth - 1))
    if k < 0.92:
        return "(%s)%s" % (_term(r, depth - 1), r.choice(["*", "+", "?"]))
    return "(%s){%d,%d}" % (_term(r, depth - 1), r.randint(0, 1), r.randint(1, 3))


def program(seed):
    r = random.Random(seed)
    body = _term(r, 3)
    if r.random() < 0.5:
        return "main := (%s)*\n" % body
    if r.random() < 0.5:
        return "main := (part /\\n/)*\npart := %s\n" % body
    return "main := %s\n" % body


def inputs(seed, count, maxlen):
    r = random.Random(seed * 7919 + 1)
    out = []
    for _ in range(count):
        n = r.randint(0, maxlen)
        out.append("".join(r.choice(ALPHA + "\n") if r.random() < 0.9 else r.choice("abcx") for _ in range(n)).encode())
    return out
