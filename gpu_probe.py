# quick GPU bring-up probe (dev only)
import sys, time, torch
sys.path.insert(0, '.')
from kleenexlang_amd import Program, compile_file, compile_source, workloads, MatchError
from oracle import oracle
def check(name, blob, data, **kw):
    try: want = oracle.run(blob, data); werr=None
    except oracle.OracleMatchError as e: want=None; werr=e.pos
    p = Program(blob, **kw)
    try: got = p.run_host(data); gerr=None
    except MatchError as e: got=None; gerr=e.pos
    ok = (got == want) and (gerr == werr)
    print(name, kw, len(data), 'OK' if ok else 'MISMATCH', (len(got) if got is not None else gerr), (len(want) if want is not None else werr), p.last_stats.unsynced_segments, flush=True)
    if not ok and got is not None and want is not None:
        for i,(a,b) in enumerate(zip(got,want)):
            if a!=b: print('  first diff at', i, got[max(0,i-40):i+40], want[max(0,i-40):i+40]); break
    return ok
blob = compile_file('flip_ab')
check('flip', blob, b'abba\nbb\n')
check('flip', blob, b'abba\nbb\n'*1000, segment_bytes=64)
check('flipfail', blob, b'abxa\n')
for prog, shape in [('apache_log','apache_log'),('csv2json','csv'),('iso_datetime_to_json','datetime'),('thousand_sep','numbers')]:
    blob = compile_file(prog)
    for n in (2000, 70000, 3000000):
        data = workloads.generate(shape, n, seed=n)
        for seg in (64, 256, 4096):
            check(prog, blob, data, segment_bytes=seg)
blob = compile_file('add_commas')
check('add_commas', blob, workloads.digits(100000), segment_bytes=256)
check('add_commas', blob, workloads.digits(100000, terminated=False), segment_bytes=256)
check('empty', blob, b'')
# timing
blob = compile_file('apache_log')
t, base, k = workloads.device_input('apache_log', 1<<30, 'cuda:0')
p = Program(blob, collect_timing=True)
out = torch.empty(int(t.numel()*1.5), dtype=torch.uint8, device='cuda:0')
for it in range(3):
    torch.cuda.synchronize(); t0=time.time()
    o = p.run_tensor(t, out); torch.cuda.synchronize(); dt=time.time()-t0
    print('apache 1GiB: %.1f ms  %.1f GB/s' % (dt*1e3, t.numel()/dt/1e9), p.last_stats.as_dict(), flush=True)
want1 = oracle.run(blob, base)
exp = workloads.tiled_expected('apache_log', want1, k)
print('len match', len(exp)==o.numel())
import hashlib
print('full-size exact:', hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()==hashlib.sha256(exp).hexdigest())
