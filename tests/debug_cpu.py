import sys, os, time, subprocess, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth, formats
print("cores", os.cpu_count(), flush=True)
ctx = capi.Context(0)
n = 20000
spec = synth.hifi_spec(100000, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec, 0, n)
t0 = time.time()
bases, offs = reads.export_ascii(0, n)
print("export", time.time() - t0, flush=True)
work = tempfile.mkdtemp(prefix="mdbg_cpu_")
fasta = os.path.join(work, "s.fasta")
with open(fasta, "wb") as f:
    for r in range(n):
        f.write(b">r%d\n" % r); f.write(bases[int(offs[r]):int(offs[r+1])].tobytes()); f.write(b"\n")
print("fasta written", time.time() - t0, flush=True)
refdrv = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "refdrv")
for threads in (8, 32, os.cpu_count()):
    tmp = os.path.join(work, f"t{threads}", "tmp")
    for d in ("", "filter", "smallContigs", "checkpoints"):
        os.makedirs(os.path.join(tmp, d), exist_ok=True)
    formats.Parameters().save(os.path.join(tmp, "parameters.gz"))
    open(os.path.join(tmp, "input.txt"), "w").write(fasta + "\n")
    t1 = time.time()
    try:
        subprocess.run([refdrv, "readSelection", tmp, tmp + "/read_data_init.txt", tmp + "/input.txt", "--threads", str(threads), "--min-read-quality", "0.000000"], check=True, timeout=60, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t2 = time.time()
        subprocess.run([refdrv, "graph", tmp, "--threads", str(threads), "--min-abundance", "0", "--firstpass"], check=True, timeout=60, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t3 = time.time()
        print("threads", threads, "readSelection", t2 - t1, "graph", t3 - t2, flush=True)
    except Exception as e:
        print("threads", threads, "failed", e, flush=True)
shutil.rmtree(work)
