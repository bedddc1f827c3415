"""-m gpu: the RCCL exchange BEHIND the C-ABI (kzg_rccl_allgather_fold, kzg_commit_coeff_form_rccl; BASELINE north_star: "a single RCCL
all-reduce of the per-GPU partial G1 sums ... behind a thin C-ABI").  The communicator belongs to the host: here a child process WITHOUT
torch (the library and /opt/rocm's librccl share one HIP runtime; PyTorch bundles its own) creates a communicator with ctypes -- one
rank on this one-GPU box (RCCL refuses two ranks on one device) -- and checks

* partial -> all-gather over the communicator -> fold == the plain commitment of the same coefficients (and == sum_i c_i tau^i G1);
* the one-call form kzg_commit_coeff_form_rccl on resident coefficients, an empty shard (identity partial), and the error path of a bad
  communicator argument;
* BASELINE config 4's shape: kzg_commit_eval_form_partial / kzg_compute_proof_partial + the exchange == the single-GPU commitment, proof and y.
With N ranks the same two calls run once per process; the N-rank exchange itself is exercised through torch.distributed by bench.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, hashlib, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import pyref
from pyref import R_
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
assert "torch" not in sys.modules
lib = _lib.load()
ctx = k.Context(0)
TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") %% R_
n = 1 << 15
srs = k.SRS.generate(TAU, n, ctx=ctx)
rng = np.random.default_rng(3)
vals = [int.from_bytes(rng.bytes(40), "little") %% R_ for _ in range(n)]
wire = pyref.frs_to_mont(vals)

rccl = C.CDLL(os.environ.get("KZG_RCCL_LIB", "/opt/rocm/lib/librccl.so"))
class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = UniqueId()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0, "ncclCommInitRank"

'''

HEALTHY = r'''
want = np.zeros(8, np.uint64); inf = C.c_uint8(0)
assert lib.kzg_commit_coeff_form(ctx.handle, srs.handle, _lib.ptr(wire), n, _lib.ptr(want), C.byref(inf)) == 0
acc, tp = 0, 1
for v in vals:
    acc = (acc + v * tp) %% R_; tp = tp * TAU %% R_
assert pyref.point_from_wire(want) == pyref.ec_mul(acc, (1, 2))

part = np.zeros(16, np.uint64)
assert lib.kzg_msm_g1_srs_partial(ctx.handle, srs.handle, 0, _lib.ptr(wire), n, _lib.ptr(part)) == 0
got = np.zeros(8, np.uint64)
rc = lib.kzg_rccl_allgather_fold(ctx.handle, comm, 1, _lib.ptr(part), _lib.ptr(got), C.byref(inf))
assert rc == 0, (rc, lib.kzg_ctx_last_error(ctx.handle))
assert np.array_equal(got, want), "fold of the gathered partial differs from the commitment"

# the one-call form on resident coefficients (hipMalloc + hipMemcpy through the runtime the library links)
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), C.c_size_t(n * 32)) == 0
assert hip.hipMemcpy(d, wire.ctypes.data_as(C.c_void_p), C.c_size_t(n * 32), 1) == 0
got2 = np.zeros(8, np.uint64)
rc = lib.kzg_commit_coeff_form_rccl(ctx.handle, srs.handle, d, n, comm, 1, _lib.ptr(got2), C.byref(inf))
assert rc == 0, (rc, lib.kzg_ctx_last_error(ctx.handle))
assert np.array_equal(got2, want)
# an empty shard contributes the identity
got3 = np.ones(8, np.uint64)
assert lib.kzg_commit_coeff_form_rccl(ctx.handle, srs.handle, None, 0, comm, 1, _lib.ptr(got3), C.byref(inf)) == 0
assert inf.value == 1 and not got3.any()
# BASELINE config 4 through the same exchange: eval-form commitment and proof of one polynomial, this rank's shard = the whole SRS here
ev = pyref.frs_to_mont([int.from_bytes(rng.bytes(40), "little") %% R_ for _ in range(n)])
z = pyref.fr_to_mont(0x1234567)
want_c = np.zeros(8, np.uint64); want_p = np.zeros(8, np.uint64); want_y = np.zeros(4, np.uint64)
assert lib.kzg_commit_eval_form(ctx.handle, srs.handle, _lib.ptr(ev), n, _lib.ptr(want_c), C.byref(inf)) == 0
assert lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(ev), n, None, n, _lib.ptr(z), _lib.ptr(want_p), C.byref(inf), _lib.ptr(want_y)) == 0
pc = np.zeros(16, np.uint64); pp = np.zeros(16, np.uint64); y = np.zeros(4, np.uint64)
assert lib.kzg_commit_eval_form_partial(ctx.handle, srs.handle, 0, _lib.ptr(ev), n, _lib.ptr(pc)) == 0
assert lib.kzg_compute_proof_partial(ctx.handle, srs.handle, 0, _lib.ptr(ev), n, None, n, _lib.ptr(z), _lib.ptr(pp), _lib.ptr(y)) == 0
gc = np.zeros(8, np.uint64); gp = np.zeros(8, np.uint64)
assert lib.kzg_rccl_allgather_fold(ctx.handle, comm, 1, _lib.ptr(pc), _lib.ptr(gc), C.byref(inf)) == 0
assert lib.kzg_rccl_allgather_fold(ctx.handle, comm, 1, _lib.ptr(pp), _lib.ptr(gp), C.byref(inf)) == 0
assert np.array_equal(gc, want_c) and np.array_equal(gp, want_p) and np.array_equal(y, want_y)
# argument errors
assert lib.kzg_rccl_allgather_fold(ctx.handle, None, 1, _lib.ptr(part), _lib.ptr(got), C.byref(inf)) == -1
assert lib.kzg_rccl_allgather_fold(ctx.handle, comm, 0, _lib.ptr(part), _lib.ptr(got), C.byref(inf)) == -1
# `world` must be the communicator's size (ADVICE r4: the staging buffer is sized by it)
assert lib.kzg_rccl_allgather_fold(ctx.handle, comm, 2, _lib.ptr(part), _lib.ptr(got), C.byref(inf)) == -1
assert b"communicator has 1 rank" in lib.kzg_ctx_last_error(ctx.handle)

# ---- round 5: config 4 in ONE call per rank over the Lagrange shard (kzg_commit_eval_form_rccl / kzg_compute_proof_rccl) ----------------
m = 1 << 12
ev4 = np.ascontiguousarray(ev[:m])
lag = srs.lagrange_shard(m, 0, m)                    # one rank: its shard is the whole basis
want_c4 = np.zeros(8, np.uint64); want_p4 = np.zeros(8, np.uint64); want_y4 = np.zeros(4, np.uint64)
assert lib.kzg_commit_eval_form(ctx.handle, srs.handle, _lib.ptr(ev4), m, _lib.ptr(want_c4), C.byref(inf)) == 0
g4 = np.zeros(8, np.uint64)
rc = lib.kzg_commit_eval_form_rccl(ctx.handle, lag.handle, _lib.ptr(ev4), m, comm, 1, _lib.ptr(g4), C.byref(inf))
assert rc == 0 and np.array_equal(g4, want_c4), (rc, lib.kzg_ctx_last_error(ctx.handle))
roots = np.zeros((m, 4), np.uint64); nr = C.c_size_t(0)
assert lib.kzg_calculate_roots_of_unity(ctx.handle, m * 32, _lib.ptr(roots), m, C.byref(nr)) == 0
for zz in (z, np.ascontiguousarray(roots[1234])):    # off the domain, and the domain point w^1234 (kzg.rs:237-260)
    assert lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(ev4), m, None, m, _lib.ptr(zz), _lib.ptr(want_p4), C.byref(inf), _lib.ptr(want_y4)) == 0
    gp4 = np.zeros(8, np.uint64); gy4 = np.zeros(4, np.uint64)
    rc = lib.kzg_compute_proof_rccl(ctx.handle, lag.handle, 0, _lib.ptr(ev4), m, m, _lib.ptr(zz), comm, 1, _lib.ptr(gp4), C.byref(inf), _lib.ptr(gy4))
    assert rc == 0, (rc, lib.kzg_ctx_last_error(ctx.handle))
    assert np.array_equal(gp4, want_p4) and np.array_equal(gy4, want_y4)
    # the resident form reads the caller's device buffer in place
    assert hip.hipMemcpy(d, ev4.ctypes.data_as(C.c_void_p), C.c_size_t(m * 32), 1) == 0
    gp5 = np.zeros(8, np.uint64)
    assert lib.kzg_compute_proof_rccl_device(ctx.handle, lag.handle, 0, d, m, m, _lib.ptr(zz), comm, 1, _lib.ptr(gp5), C.byref(inf), None) == 0
    assert np.array_equal(gp5, want_p4)
    g5 = np.zeros(8, np.uint64)
    assert lib.kzg_commit_eval_form_rccl_device(ctx.handle, lag.handle, d, m, comm, 1, _lib.ptr(g5), C.byref(inf)) == 0 and np.array_equal(g5, want_c4)
# a LOCAL failure (this rank's slice is longer than its shard) still goes through the collectives and comes back as the rank's own status
big = np.zeros((m + 1, 4), np.uint64)
assert lib.kzg_commit_eval_form_rccl(ctx.handle, lag.handle, _lib.ptr(big), m + 1, comm, 1, _lib.ptr(g4), C.byref(inf)) == _lib.ERR_SRS_CAPACITY_EXCEEDED
assert lib.kzg_compute_proof_rccl(ctx.handle, lag.handle, 0, _lib.ptr(big), m + 1, 2 * m, _lib.ptr(z), comm, 1, _lib.ptr(g4), C.byref(inf), None) == _lib.ERR_SRS_CAPACITY_EXCEEDED
assert lib.kzg_commit_coeff_form_rccl(ctx.handle, srs.handle, d, n + 1, comm, 1, _lib.ptr(got2), C.byref(inf)) == _lib.ERR_MSM_LENGTH_MISMATCH
# a local failure that comes back as KZG_ERR_INVALID_ARG (slot 0 still holds an MSM of the caller's) must NOT be taken for "no collective issued":
# the rank joins BOTH exchanges of the call with poisoned rows (ADVICE r5: it used to skip the second one and leave its peers in it for 60 s)
assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, _lib.ptr(wire), n, 0) == 0
rc = lib.kzg_compute_proof_rccl(ctx.handle, lag.handle, 0, _lib.ptr(ev4), m, m, _lib.ptr(z), comm, 1, _lib.ptr(g4), C.byref(inf), None)
assert rc == _lib.ERR_INVALID_ARG, rc
assert b"joined both collectives" in lib.kzg_ctx_last_error(ctx.handle), lib.kzg_ctx_last_error(ctx.handle)
assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, _lib.ptr(got2), C.byref(inf), None) == 0 and np.array_equal(got2, want)
# ... and the context is usable afterwards (no slot left in flight)
assert lib.kzg_compute_proof_rccl(ctx.handle, lag.handle, 0, _lib.ptr(ev4), m, m, _lib.ptr(z), comm, 1, _lib.ptr(gp4), C.byref(inf), None) == 0
lag.close()
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
print("rccl c-abi ok")
'''


POISONED = r'''
# KZG_RCCL_TEST_POISON=1 (test hook of csrc/multi.hip, compiled into libkzg_bn254_mi355x_hooks.so only): this healthy rank marks every row it sends as failed -> every rank of a call returns
# KZG_ERR_PEER after the SAME collectives a successful call issues, and nothing stays in flight
inf = C.c_uint8(0)
part = np.zeros(16, np.uint64)
assert lib.kzg_msm_g1_srs_partial(ctx.handle, srs.handle, 0, _lib.ptr(wire), n, _lib.ptr(part)) == 0
got = np.zeros(8, np.uint64)
assert lib.kzg_rccl_allgather_fold(ctx.handle, comm, 1, _lib.ptr(part), _lib.ptr(got), C.byref(inf)) == _lib.ERR_PEER
assert b"rank(s) 0" in lib.kzg_ctx_last_error(ctx.handle)
m = 1 << 12
ev4 = np.ascontiguousarray(wire[:m]); z = pyref.fr_to_mont(0x1234567)
lag = srs.lagrange_shard(m, 0, m)
assert lib.kzg_commit_eval_form_rccl(ctx.handle, lag.handle, _lib.ptr(ev4), m, comm, 1, _lib.ptr(got), C.byref(inf)) == _lib.ERR_PEER
assert lib.kzg_compute_proof_rccl(ctx.handle, lag.handle, 0, _lib.ptr(ev4), m, m, _lib.ptr(z), comm, 1, _lib.ptr(got), C.byref(inf), None) == _lib.ERR_PEER
roots = np.zeros((m, 4), np.uint64); nr = C.c_size_t(0)
assert lib.kzg_calculate_roots_of_unity(ctx.handle, m * 32, _lib.ptr(roots), m, C.byref(nr)) == 0
zon = np.ascontiguousarray(roots[5])
assert lib.kzg_compute_proof_rccl(ctx.handle, lag.handle, 0, _lib.ptr(ev4), m, m, _lib.ptr(zon), comm, 1, _lib.ptr(got), C.byref(inf), None) == _lib.ERR_PEER
p1 = np.zeros(8, np.uint64)
assert lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(ev4), m, None, m, _lib.ptr(z), _lib.ptr(p1), C.byref(inf), None) == 0   # slot 0 is free again
lag.close()
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
print("poison path ok")
'''


def run_child(body, **env_extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    res = subprocess.run([sys.executable, "-c", (CHILD + body) % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-800:], res.stderr[-2500:])
    return res.stdout


def test_rccl_exchange_behind_the_c_abi_one_rank():
    assert "rccl c-abi ok" in run_child(HEALTHY)


def test_rccl_failure_protocol_poisoned_rank():
    hooks = os.path.join(ROOT, "rust-kzg-bn254_amd", "libkzg_bn254_mi355x_hooks.so")
    assert os.path.exists(hooks), "make -C rust-kzg-bn254_amd/csrc hooks (__graft_entry__.build() does it)"
    assert "poison path ok" in run_child(POISONED, KZG_RCCL_TEST_POISON="1", KZG_LIB_PATH=hooks)
    # the shipped library has no such switch: the same child, same variable, healthy results
    assert "rccl c-abi ok" in run_child(HEALTHY, KZG_RCCL_TEST_POISON="1")
