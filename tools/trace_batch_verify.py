"""Phase times of kzg_verify_blob_kzg_proof_batch (BASELINE config 5, 4 096 rows) with KZG_VB_TRACE=1: the bench's workload (256 distinct blobs of
35 .. 50 000 raw bytes, each used 16 times), TRACE_CALLS calls (default 60) with the cgroup's CPU-throttling counters beside every call.  Usage (GPU box): [KZG_VB_TRACE=1] [KZG_HOST_THREADS=16] python tools/trace_batch_verify.py"""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
from rust_kzg_bn254_amd.helpers import pad_payload
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 12, ctx=ctx)
u8p = C.POINTER(C.c_uint8)
rng5 = np.random.default_rng(5)
rows5 = []
for n_raw in rng5.integers(35, 50000, size=256):
    data = pad_payload(rng5.integers(32, 127, size=int(n_raw), dtype=np.uint8).tobytes())
    npad = 1
    while npad < len(data) // 32:
        npad <<= 1
    buf = np.frombuffer(data, dtype=np.uint8)
    c5 = np.zeros(8, np.uint64); p5 = np.zeros(8, np.uint64); ci5 = C.c_uint8(0); pi5 = C.c_uint8(0)
    assert lib.kzg_commit_and_prove_blob(ctx.handle, srs.handle, buf.ctypes.data_as(u8p), len(data), npad, _lib.ptr(c5), C.byref(ci5), _lib.ptr(p5), C.byref(pi5), None, None) == 0
    rows5.append((data, c5, p5))
nb = 4096
sel5 = [rows5[i % 256] for i in range(nb)]
ptrs5, lens5, _keep5 = _lib.blob_args([r[0] for r in sel5])
cm5 = np.ascontiguousarray(np.stack([r[1] for r in sel5])); pf5 = np.ascontiguousarray(np.stack([r[2] for r in sel5]))
tau_g2 = np.zeros(16, np.uint64)
lib.kzg_g2_mul_generator(_lib.ptr(k.fr.fr_from_int(tau)), _lib.ptr(tau_g2))
ok5 = C.c_int32(0)
def cg(name):
    """one file of this process's cgroup (v2), or None"""
    for base in ("/sys/fs/cgroup", "/sys/fs/cgroup/cpu"):
        try:
            return open(os.path.join(base, name)).read()
        except OSError:
            pass
    return None
def throttled():
    s = cg("cpu.stat") or ""
    d = dict(l.split() for l in s.splitlines() if len(l.split()) == 2)
    return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", d.get("throttled_time", 0)))
calls = int(os.environ.get("TRACE_CALLS", "60"))
gap_ms = float(os.environ.get("TRACE_GAP_MS", "0"))
print("cpu.max:", (cg("cpu.max") or cg("cpu.cfs_quota_us") or "n/a").strip(), "| hardware threads", os.cpu_count(), "| affinity", len(os.sched_getaffinity(0)),
      "| KZG_HOST_THREADS", os.environ.get("KZG_HOST_THREADS"), "KZG_HOST_THREADS_MAX", os.environ.get("KZG_HOST_THREADS_MAX"), flush=True)
if os.environ.get("TRACE_GC", "1") == "0":
    import gc
    gc.disable()
tracing = os.environ.get("KZG_VB_TRACE", "0") != "0"
def sched_of_threads():
    """/proc/self/task/<tid>/schedstat of every thread: tid -> (ns on a CPU, ns RUNNABLE BUT WAITING for one, time slices); {} where the kernel does not keep it"""
    out = {}
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                f = open("/proc/self/task/%s/schedstat" % tid).read().split()
                out[tid] = (int(f[0]), int(f[1]), int(f[2]))
            except (OSError, ValueError, IndexError):
                pass
    except OSError:
        pass
    return out
ts, thr, cpu, waits, waiters = [], [], [], [], []
for i in range(calls):
    s0 = sched_of_threads()
    a = throttled()
    c0 = time.process_time()
    t0 = time.perf_counter()
    rc_ = lib.kzg_verify_blob_kzg_proof_batch(ctx.handle, ptrs5, lens5, _lib.ptr(cm5), _lib.ptr(pf5), nb, _lib.ptr(tau_g2), C.byref(ok5))
    assert os.environ.get("TRACE_NOASSERT") or (rc_ == 0 and ok5.value == 1)
    ts.append((time.perf_counter() - t0) * 1e3)
    cpu.append((time.process_time() - c0) * 1e3)
    if tracing:
        print("  call %d seen from Python: %.3f ms" % (i, ts[-1]), file=sys.stderr, flush=True)
    b = throttled()
    thr.append((b[0] - a[0], (b[1] - a[1]) / 1e3))
    s1 = sched_of_threads()
    dw = [(s1[t][1] - s0[t][1]) / 1e6 for t in s1 if t in s0]
    waits.append((max(dw) if dw else float("nan"), sum(dw), len(dw)))
    if dw and max(dw) > 1.0:                                          # who waited: the thread's name (the pool's are "kzg-pool", the caller is the process's main thread)
        tid = max((t for t in s1 if t in s0), key=lambda t: s1[t][1] - s0[t][1])
        try:
            who = open("/proc/self/task/%s/comm" % tid).read().strip() + ("(main)" if int(tid) == os.getpid() else "")
        except OSError:
            who = "?"
        waiters.append((i, who, max(dw), ts[-1]))
    if gap_ms: time.sleep(gap_ms / 1e3)
print("per call ms:", " ".join("%.2f" % t for t in ts), "| blob MiB", sum(len(r[0]) for r in sel5) / 2 ** 20, flush=True)
print("cgroup throttling per call (periods, ms):", " ".join("%d/%.1f" % t for t in thr), flush=True)
print("runqueue wait per call, ms (the longest any one thread of the process sat runnable without a CPU / summed over its %d threads):" % waits[-1][2],
      " ".join("%.2f/%.2f" % (w[0], w[1]) for w in waits), flush=True)
print("threads that waited > 1 ms for a CPU (call: thread name, its wait ms, the call's ms):", " ".join("%d:%s,%.1f,%.1f" % w for w in waiters) or "none", flush=True)
print("process CPU ms per call: median %.1f  mean %.1f (all threads of the process; quota = cpu.max)" % (sorted(cpu)[len(cpu) // 2], sum(cpu) / len(cpu)), flush=True)
ts = ts[int(os.environ.get("TRACE_SKIP", "2")):]                     # the first calls size the staging buffers and start the pool
srt = sorted(ts)
print("median %.2f  mean %.2f  p90 %.2f  p99 %.2f  max %.2f over %d calls" % (srt[len(srt) // 2], sum(ts) / len(ts), srt[int(len(srt) * 0.9)], srt[min(len(srt) - 1, int(len(srt) * 0.99))], srt[-1], len(ts)), flush=True)
