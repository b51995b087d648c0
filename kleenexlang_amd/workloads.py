"""Seeded synthetic inputs of the shapes BASELINE.json names.

Shapes follow the reference's own data:
  apache_log   lines like test/data/apache_log/example.log (combined log format, single
               spaces, numeric size, no quote/backslash inside quoted fields; line lengths
               spread like the sample: min≈81, median≈194, mean≈228, max≈980)
  csv          rows like test/data/csv/gen_csv.pl -f 1 (gen_csv.pl:33-77)
  datetime     lines like test/data/datetime/gen_datetime.pl:22-45
  numbers      digit lines like test/data/numbers/gen_numbers.pl (never an empty first line)
  digits       BASELINE config 1: random ASCII digits, optionally newline-terminated
Large inputs are a seeded base chunk replicated (the reference's own method for its big log,
test/data/apache_log/generate_big_log.sh:4-5); `tiled_expected` states what the output of a
replicated input must be in terms of the base chunk's output, which lets full-size runs be
checked bit-exactly on the GPU without a CPU pass.
"""
import random

_ALPHA = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
_MONTHS = ["Jan", "Feb", "Mar", "Apr", "May", "Jun", "Jul", "Aug", "Sep", "Oct", "Nov", "Dec"]
_METHODS = ["GET", "GET", "GET", "GET", "POST", "HEAD"]
_AGENTS = [
    "Mozilla/5.0 (compatible; MSIE 10.0; Windows NT 6.2; Win64; x64; Trident/6.0; Touch)",
    "Mozilla/5.0 (Windows NT 5.1) AppleWebKit/537.36 (KHTML, like Gecko) Chrome/32.0.1700.107 Safari/537.36",
    "Mozilla/5.0 (X11; Linux x86_64; rv:36.0) Gecko/20100101 Firefox/36.0",
    "Mozilla/5.0 (compatible; Googlebot/2.1; +http://www.google.com/bot.html)",
    "Mozilla/5.0 (Macintosh; Intel Mac OS X 10_10_2) AppleWebKit/600.4.10 (KHTML, like Gecko) Version/8.0.4 Safari/600.4.10",
    "-",
    "curl/7.35.0",
]
_URLCH = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789-_.~%&=+"


def _word(r, lo, hi, alphabet=_ALPHA):
    return "".join(r.choice(alphabet) for _ in range(r.randint(lo, hi)))


def apache_log_line(r):
    ip = "%d.%d.%d.%d" % (r.randint(1, 255), r.randint(0, 255), r.randint(0, 255), r.randint(0, 255))
    ts = "%02d/%s/%d:%02d:%02d:%02d %s%02d00" % (r.randint(1, 28), r.choice(_MONTHS), r.randint(2010, 2016),
                                                r.randint(0, 23), r.randint(0, 59), r.randint(0, 59),
                                                r.choice("+-"), r.randint(0, 12))
    path = "/" + "/".join(_word(r, 2, 12, _URLCH) for _ in range(r.randint(1, 4)))
    u = r.random()
    if u < 0.25:
        ref = "-"
    elif u < 0.9:
        ref = "http://%s.com/%s" % (_word(r, 4, 14), _word(r, 0, 40, _URLCH))
    else:  # the long tail of the sample: search-engine referers with long query strings
        ref = "http://www.%s.com/url?%s" % (_word(r, 4, 10), _word(r, 150, 700, _URLCH))
    return '%s - - [%s] "%s %s HTTP/1.%d" %d %d "%s" "%s"\n' % (
        ip, ts, r.choice(_METHODS), path, r.randint(0, 1), r.choice([200, 200, 200, 304, 404, 301, 500]),
        r.randint(0, 99999), ref, r.choice(_AGENTS))


def csv_row(r):
    name = lambda lo, hi: _word(r, hi - lo + 1, hi - lo + 1)  # gen_name(min,max): max-min+1 letters
    return "%d,%s,%s,%s@%s.%s,%s,%d.%d.%d.%d\n" % (
        r.randrange(0, 10000000), name(6, 25), name(6, 25), name(5, 25), name(10, 30), name(2, 4), name(6, 25),
        r.randrange(0, 255), r.randrange(0, 255), r.randrange(0, 255), r.randrange(0, 255))


def datetime_line(r):
    tz = "Z" if r.random() < 0.3 else "%s%02d:%02d" % (r.choice("+-"), r.randint(0, 23), r.randint(0, 59))
    return "%d-%02d-%02dT%02d:%02d:%02d%s\n" % (r.randint(1000, 9999), r.randint(1, 12), r.randint(1, 31),
                                                r.randint(0, 23), r.randint(0, 59), r.randint(0, 59), tz)


def numbers_line(r):
    return "".join(r.choice("0123456789") for _ in range(r.randint(1, 1000))) + "\n"


_LINE = {"apache_log": apache_log_line, "csv": csv_row, "datetime": datetime_line, "numbers": numbers_line}
# which input shape each workload program consumes
PROGRAM_INPUT = {"apache_log": "apache_log", "csv2json": "csv", "iso_datetime_to_json": "datetime",
                 "thousand_sep": "numbers"}


def generate(shape, nbytes, seed=0x4B4C4558):
    """About `nbytes` (never more) of whole lines of the given shape, as bytes."""
    r = random.Random(seed)
    gen = _LINE[shape]
    parts, total = [], 0
    while True:
        line = gen(r)
        if total + len(line) > nbytes:
            break
        parts.append(line)
        total += len(line)
    return "".join(parts).encode("ascii")


def digits(n, seed=0x4B4C4558, terminated=True):
    """BASELINE config 1: n random ASCII digits (+ '\\n')."""
    r = random.Random(seed)
    s = "".join(r.choice("0123456789") for _ in range(n))
    return (s + ("\n" if terminated else "")).encode("ascii")


def tiled_expected(program, base_out, k):
    """Output of `program` on (base chunk × k), given its output on one base chunk."""
    if k == 1:
        return base_out
    if program == "apache_log":  # "[" R "\n]\n"  →  "[" (R ",\n")^(k-1) R "\n]\n"
        assert base_out[:1] == b"[" and base_out[-3:] == b"\n]\n"
        body = base_out[1:-3]
        return b"[" + (body + b",\n") * (k - 1) + body + b"\n]\n"
    if program in ("csv2json", "iso_datetime_to_json", "thousand_sep"):  # records are independent
        return base_out * k
    raise KeyError(program)


def device_input(program, nbytes, device, base_bytes=32 << 20, seed=0x4B4C4558):
    """(tensor, base, k): a uint8 CUDA tensor of ≈nbytes made of k copies of a seeded base chunk."""
    import torch
    shape = PROGRAM_INPUT[program]
    base = generate(shape, min(base_bytes, nbytes), seed)
    k = max(1, nbytes // len(base))
    tb = torch.frombuffer(bytearray(base), dtype=torch.uint8).to(device)
    t = tb.repeat(k) if k > 1 else tb
    return t, base, k


def tiled_parts(program, base_out, k):
    """The expected output of `program` on (base chunk × k) as (prefix, unit, reps, suffix):
    prefix ++ unit × reps ++ suffix.  The same statement as `tiled_expected`, in a form that can be
    checked piecewise at any size (and at any offset: see `check_tiled_on_device`)."""
    if program == "apache_log":   # "[" (R ",\n")^(k-1) R "\n]\n"
        assert base_out[:1] == b"[" and base_out[-3:] == b"\n]\n"
        body = base_out[1:-3]
        return b"[", body + b",\n", k - 1, body + b"\n]\n"
    if program in ("csv2json", "iso_datetime_to_json", "thousand_sep"):
        return b"", base_out, k, b""
    raise KeyError(program)


def tiled_total(parts):
    prefix, unit, reps, suffix = parts
    return len(prefix) + len(unit) * reps + len(suffix)


def check_tiled_on_device(out, offset, parts, chunk_bytes=256 << 20):
    """True iff the CUDA uint8 tensor `out` equals bytes [offset, offset + out.numel()) of the stream
    described by `parts` (see `tiled_parts`).  Every byte is compared, on the device, in bounded
    chunks — this is how 10 GiB runs are verified bit-exactly without a CPU pass (the unit's content
    comes from the CPU oracle run on ONE base chunk)."""
    import torch
    prefix, unit, reps, suffix = parts
    dev = out.device
    n = out.numel()
    total = tiled_total(parts)
    if offset < 0 or offset + n > total:
        return False
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if len(b) else torch.empty(0, dtype=torch.uint8, device=dev)
    pos, end = offset, offset + n          # positions in the expected stream
    def take(lo, hi):                       # the slice of `out` holding stream positions [lo, hi)
        return out[lo - offset:hi - offset]
    # prefix
    if pos < len(prefix):
        hi = min(end, len(prefix))
        if not torch.equal(take(pos, hi), to_dev(prefix[pos:hi])):
            return False
        pos = hi
    # repeated unit
    u0, u1 = len(prefix), len(prefix) + len(unit) * reps
    if pos < min(end, u1) and len(unit):
        U = len(unit)
        ud = to_dev(unit)
        hi = min(end, u1)
        # ragged head up to the next unit boundary
        ph = (pos - u0) % U
        if ph:
            h = min(hi, pos + (U - ph))
            if not torch.equal(take(pos, h), ud[ph:ph + (h - pos)]):
                return False
            pos = h
        per = max(1, chunk_bytes // U)
        while pos + U <= hi:
            c = min(per, (hi - pos) // U)
            if not bool((take(pos, pos + c * U).view(c, U) == ud).all()):
                return False
            pos += c * U
        if pos < hi:                        # ragged tail
            if not torch.equal(take(pos, hi), ud[:hi - pos]):
                return False
            pos = hi
    # suffix
    if pos < end:
        s0 = u1
        if not torch.equal(take(pos, end), to_dev(suffix[pos - s0:end - s0])):
            return False
    return True
