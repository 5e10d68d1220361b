import faulthandler, sys, os, time
faulthandler.dump_traceback_later(45, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
use_torch = len(sys.argv) > 1 and sys.argv[1] == "torch"
t0 = time.time()
def log(*a):
    print(f"[{time.time()-t0:6.1f}s]", *a, flush=True)
if use_torch:
    import torch
    log("torch imported", torch.__version__)
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
    log("torch cuda ready")
from metamdbg_amd import capi, synth
ctx = capi.Context(0)
log("ctx", ctx.device_info())
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
log("reads", reads.info())
ctx.timing(True)
mins = ctx.scan(reads, K=15, density=0.005, hpc=True)
log("scan", mins.info(), ctx.timing_get("scan"))
corr = ctx.purge_palindromes(mins, 4, 100)
log("purge", corr.info(), ctx.timing_get("purge_palindromes"))
t = ctx.kminmer_count_first(corr, 4, 0)
log("count", t.info(), {k: ctx.timing_get(k) for k in ("kminmer_insert", "kminmer_rescue", "kminmer_emit")})
