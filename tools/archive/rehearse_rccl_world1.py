"""One-rank rehearsal of the collective calls bench.py makes at N > 1 (RCCL through torch.distributed 'nccl'):
init with device_id, barrier, all_gather of the 16 x int64 partial (list form), all_reduce MAX, broadcast — while the two-slot
MSM pipeline is running on the library's own streams.  The driver's multi-GPU run is the first time these run with N > 1."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib, sharding
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
lib = _lib.load(); ctx = k.Context(0)
n = 1 << 18
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
sc = bench.blob_like_scalars(n, 9)
d = torch.from_numpy(sc.view(np.int64)).cuda(); torch.cuda.synchronize()
want = np.zeros(8, np.uint64); inf = C.c_uint8(0)
assert lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(want), C.byref(inf)) == 0
dist.barrier()
t0 = time.perf_counter()
prev = None; res = None
for i in range(20):
    assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, i & 1) == 0
    if prev is not None:
        part = np.zeros(16, np.uint64)
        assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, None, None, _lib.ptr(part)) == 0
        res = sharding.fold_partials(sharding.gather_partials(part, 1, "cuda", force_collective=True))
        assert np.array_equal(res, want)
    prev = i & 1
part = np.zeros(16, np.uint64)
assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, None, None, _lib.ptr(part)) == 0
res = sharding.fold_partials(sharding.gather_partials(part, 1, "cuda", force_collective=True))
torch.cuda.synchronize(); dist.barrier()
el = time.perf_counter() - t0
t = torch.tensor([el], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
chk = torch.from_numpy(res.view(np.int64).copy()).cuda(); ref = chk.clone(); dist.broadcast(ref, src=0)
assert torch.equal(ref, chk) and np.array_equal(res, want)
# where does the host time go?  (one more pass, each piece timed on the host)
import collections
acc = collections.defaultdict(float)
pg = sharding.PartialGatherer(1, "cuda")
def tick(name, t): acc[name] += time.perf_counter() - t
prev = None
for i in range(20):
    t = time.perf_counter(); assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, i & 1) == 0; tick("begin", t)
    if prev is not None:
        part = np.zeros(16, np.uint64)
        t = time.perf_counter(); assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, None, None, _lib.ptr(part)) == 0; tick("end(wait+epilogue)", t)
        t = time.perf_counter(); tt = torch.from_numpy(part.view(np.int64).copy()).to("cuda"); tick("h2d", t)
        t = time.perf_counter(); outs = [torch.empty(16, dtype=torch.int64, device="cuda")]; dist.all_gather(outs, tt); tick("all_gather call", t)
        t = time.perf_counter(); g = torch.stack(outs).cpu().numpy().view(np.uint64); tick("stack+d2h", t)
        t = time.perf_counter(); sharding.fold_partials(g); tick("fold", t)
        t = time.perf_counter(); g2 = pg.gather(part); tick("PartialGatherer.gather (preallocated, pinned)", t)
        assert np.array_equal(g2, g)
    prev = i & 1
assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, None, None, _lib.ptr(part)) == 0
# pipelined form: exchange of step k-1 started after its MSM, finished one step later
t0 = time.perf_counter(); prev = None; exch = False
for i in range(40):
    assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, i & 1) == 0
    if prev is not None:
        assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, None, None, _lib.ptr(part)) == 0
        if exch:
            assert np.array_equal(sharding.fold_partials(pg.finish()), want)
        pg.start(part); exch = True
    prev = i & 1
assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, None, None, _lib.ptr(part)) == 0
assert np.array_equal(sharding.fold_partials(pg.finish()), want)
print("  pipelined with asynchronous exchange: %.3f ms per 2^18 step" % ((time.perf_counter() - t0) / 40 * 1e3))
for k_, v in acc.items(): print("  %-22s %.1f us per step" % (k_, v / 20 * 1e6))
t = torch.tensor([el], dtype=torch.float64, device="cuda")
print("rccl world-1 rehearsal ok: %.3f ms per 2^18 step incl. all_gather" % (float(t.item()) / 20 * 1e3), flush=True)
dist.destroy_process_group()
