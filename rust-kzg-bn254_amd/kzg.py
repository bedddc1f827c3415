"""Mirror of prover/src/kzg.rs `KZG`: same method names, argument meaning and error behaviour; the
G1 MSM and the Fr NTT/IFFT behind them run as HIP kernels through the C-ABI."""
import ctypes as C

import numpy as np

from . import _lib, helpers
from .errors import CommitError, FFTError, GenericError, NotOnCurveError, SerializationError, SrsCapacityExceeded
from .fr import g1_is_identity
from .polynomial import PolynomialCoeffForm, PolynomialEvalForm


def _drain_slot(ctx, slot, proof=False):
    """Wait for an asynchronous call left in flight on `slot` and drop its result (generator clean-up paths)."""
    lib = _lib.load()
    out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0); y = np.zeros(4, dtype=np.uint64)
    if proof:
        lib.kzg_compute_proof_end(ctx.handle, slot, _lib.ptr(out), C.byref(inf), _lib.ptr(y))
    else:
        lib.kzg_msm_g1_srs_end(ctx.handle, slot, _lib.ptr(out), C.byref(inf), None)


class KZG:
    def __init__(self, ctx=None):
        self.ctx = ctx
        self.expanded_roots_of_unity = np.zeros((0, 4), dtype=np.uint64)

    @classmethod
    def new(cls, ctx=None):
        return cls(ctx)

    def _ctx(self):
        if self.ctx is None:
            self.ctx = _lib.default_context()
        return self.ctx

    # kzg.rs:65-72
    def calculate_and_store_roots_of_unity(self, length_of_data_after_padding: int):
        self.expanded_roots_of_unity = helpers.calculate_roots_of_unity(length_of_data_after_padding, self._ctx())

    def get_roots_of_unities(self):
        return self.expanded_roots_of_unity.copy()

    def get_nth_root_of_unity(self, i):
        return self.expanded_roots_of_unity[i] if 0 <= i < len(self.expanded_roots_of_unity) else None

    # kzg.rs:84-104
    def commit_eval_form(self, polynomial: PolynomialEvalForm, srs):
        if len(polynomial) > len(srs):
            raise SrsCapacityExceeded(len(polynomial), len(srs))
        ctx = self._ctx()
        evals = _lib.as_u64(polynomial.evaluations(), 4)
        out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0)
        rc = _lib.load().kzg_commit_eval_form(ctx.handle, srs.handle, _lib.ptr(evals), len(evals), _lib.ptr(out), C.byref(inf))
        if rc == _lib.ERR_NOT_POWER_OF_TWO:
            raise FFTError("length provided is not a power of 2")
        if rc == _lib.ERR_DOMAIN:
            raise FFTError("Could not perform IFFT due to domain consturction error")
        ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(_lib.status_message(rc))
        return out

    # kzg.rs:107-125
    def commit_coeff_form(self, polynomial: PolynomialCoeffForm, srs):
        if len(polynomial) > len(srs):
            raise SerializationError("polynomial length is not correct")
        ctx = self._ctx()
        coeffs = _lib.as_u64(polynomial.coeffs(), 4)
        out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0)
        rc = _lib.load().kzg_commit_coeff_form(ctx.handle, srs.handle, _lib.ptr(coeffs), len(coeffs), _lib.ptr(out), C.byref(inf))
        if rc == _lib.ERR_MSM_LENGTH_MISMATCH:
            raise CommitError(str(min(len(coeffs), len(srs))))
        ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(_lib.status_message(rc))
        return out

    # ---- batched commitments: many polynomials of one length against one SRS in one kernel sequence (no counterpart in the
    # reference, which commits one polynomial per call; same values as that many calls) -----------------------------------------
    def _commit_batch(self, rows, srs, eval_form):
        ctx = self._ctx()
        rows = [_lib.as_u64(r, 4).reshape(-1, 4) for r in rows]
        if not rows:
            return np.zeros((0, 8), dtype=np.uint64)
        n = len(rows[0])
        if any(len(r) != n for r in rows):
            raise GenericError("batched commitments need polynomials of one length")
        if n > len(srs):
            if eval_form:
                raise SrsCapacityExceeded(n, len(srs))
            raise SerializationError("polynomial length is not correct")
        data = np.ascontiguousarray(np.concatenate(rows, axis=0)) if n else np.zeros((0, 4), dtype=np.uint64)
        out = np.zeros((len(rows), 8), dtype=np.uint64)
        fn = _lib.load().kzg_commit_eval_form_batch if eval_form else _lib.load().kzg_commit_coeff_form_batch
        rc = fn(ctx.handle, srs.handle, _lib.ptr(data) if n else None, n, len(rows), _lib.ptr(out), None)
        ctx.check_device(rc)
        if rc == _lib.ERR_NOT_POWER_OF_TWO:
            raise FFTError("length provided is not a power of 2")
        if rc != _lib.OK:
            raise GenericError(_lib.status_message(rc))
        return out

    def commit_coeff_form_batch(self, polynomials, srs):
        """commit_coeff_form of every polynomial (all of one length) in ONE call: (count, 8) wire points."""
        return self._commit_batch([p.coeffs() for p in polynomials], srs, False)

    def commit_eval_form_batch(self, polynomials, srs):
        """commit_eval_form of every polynomial (all of one padded length) in ONE call; the Lagrange basis of that length stays cached with the SRS."""
        return self._commit_batch([p.evaluations() for p in polynomials], srs, True)

    def commit_blob_batch(self, blobs, srs):
        """commit_blob (kzg.rs:182-185) of every blob; blobs whose polynomials differ in length are grouped by length."""
        polys = [b.to_polynomial_eval_form() for b in blobs]
        out = np.zeros((len(polys), 8), dtype=np.uint64)
        by_len = {}
        for i, p in enumerate(polys):
            by_len.setdefault(len(p), []).append(i)
        for _, idx in sorted(by_len.items()):
            res = self.commit_eval_form_batch([polys[i] for i in idx], srs)
            for i, r in zip(idx, res):
                out[i] = r
        return out

    # ---- streams of commitments, two in flight (kzg_*_begin(slot) / kzg_msm_g1_srs_end(slot)) --------------------------
    def _pipelined(self, items, begin):
        """begin(item, slot) -> status, or None when the item needs no device work (yields the identity)."""
        ctx = self._ctx()
        lib = _lib.load()

        def end(slot):
            out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0)
            rc = lib.kzg_msm_g1_srs_end(ctx.handle, slot, _lib.ptr(out), C.byref(inf), None)
            ctx.check_device(rc)
            if rc != _lib.OK:
                raise CommitError(_lib.status_message(rc))
            return out

        prev = None
        k = 0
        try:
            for item in items:
                slot = k & 1
                rc = begin(item, slot)
                if rc is None:
                    if prev is not None:
                        p, prev = prev, None
                        yield end(p)
                    yield np.zeros(8, dtype=np.uint64)
                    continue
                k += 1
                if rc == _lib.ERR_SRS_CAPACITY_EXCEEDED:
                    raise SrsCapacityExceeded(len(item), 0)
                if rc == _lib.ERR_NOT_POWER_OF_TWO:
                    raise FFTError("length provided is not a power of 2")
                ctx.check_device(rc)
                if rc != _lib.OK:
                    raise CommitError(_lib.status_message(rc))
                if prev is not None:
                    p, prev = prev, slot
                    yield end(p)
                else:
                    prev = slot
            if prev is not None:
                p, prev = prev, None
                yield end(p)
        finally:
            # an exception in begin() for item k+1, or a consumer that stops early, must not leave a slot in flight: a pending
            # slot can never be begun again and slot 0 blocks every synchronous call of the context
            if prev is not None:
                _drain_slot(ctx, prev)

    def commit_coeff_form_stream(self, polynomials, srs):
        """`commit_coeff_form` over a stream of polynomials: the H2D copy and the sort of polynomial k+1 run beside the bucket
        accumulation of polynomial k.  Yields the commitments in order."""
        ctx = self._ctx()
        lib = _lib.load()

        def begin(polynomial, slot):
            if len(polynomial) > len(srs):
                raise SerializationError("polynomial length is not correct")
            coeffs = _lib.as_u64(polynomial.coeffs(), 4)
            if len(coeffs) == 0:
                return None
            return lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, _lib.ptr(coeffs), len(coeffs), slot)
        return self._pipelined(polynomials, begin)

    def commit_eval_form_stream(self, polynomials, srs):
        """`commit_eval_form` over a stream of evaluation-form polynomials (H2D, IFFT and MSM of each on its slot's stream)."""
        ctx = self._ctx()
        lib = _lib.load()

        def begin(polynomial, slot):
            ev = _lib.as_u64(polynomial.evaluations(), 4)
            if len(ev) > len(srs):
                raise SrsCapacityExceeded(len(ev), len(srs))
            return lib.kzg_commit_eval_form_begin(ctx.handle, srs.handle, _lib.ptr(ev), len(ev), slot)
        return self._pipelined(polynomials, begin)

    def commit_blob_stream(self, blobs, srs):
        """`commit_blob` over a stream of blobs: bytes in, points out, nothing but the 32-byte-per-element blob crosses PCIe."""
        ctx = self._ctx()
        lib = _lib.load()

        def begin(blob, slot):
            data = np.frombuffer(blob.data(), dtype=np.uint8)
            n = 1
            while n < -(-len(data) // 32):
                n <<= 1
            if n > len(srs):
                raise SrsCapacityExceeded(n, len(srs))
            return lib.kzg_commit_blob_begin(ctx.handle, srs.handle, data.ctypes.data_as(_lib.u8p) if len(data) else None, len(data), slot)
        return self._pipelined(blobs, begin)

    def compute_proof_stream(self, items, srs, want_y=False):
        """`compute_proof` over a stream of (polynomial, z_fr) pairs, two in flight (`kzg_compute_proof_begin` / `_end`): upload,
        batch inversion and quotient of proof k+1 run beside the MSM of proof k.  Yields proofs (or (proof, y)) in order."""
        ctx = self._ctx()
        lib = _lib.load()

        def end(slot):
            out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0); y = np.zeros(4, dtype=np.uint64)
            rc = lib.kzg_compute_proof_end(ctx.handle, slot, _lib.ptr(out), C.byref(inf), _lib.ptr(y))
            ctx.check_device(rc)
            if rc != _lib.OK:
                raise GenericError(_lib.status_message(rc))
            return (out, y) if want_y else out

        prev = None
        try:
            for k, (polynomial, z_fr) in enumerate(items):
                ev = _lib.as_u64(polynomial.evaluations(), 4)
                z = np.ascontiguousarray(_lib.as_u64(z_fr, 0).reshape(4))
                slot = k & 1
                rc = lib.kzg_compute_proof_begin(ctx.handle, srs.handle, _lib.ptr(ev), len(ev), None, len(ev), _lib.ptr(z), slot)
                if rc == _lib.ERR_SRS_CAPACITY_EXCEEDED:
                    raise SrsCapacityExceeded(len(ev), len(srs))
                ctx.check_device(rc)
                if rc != _lib.OK:
                    raise GenericError(_lib.status_message(rc))
                if prev is not None:
                    p, prev = prev, slot
                    yield end(p)
                else:
                    prev = slot
            if prev is not None:
                p, prev = prev, None
                yield end(p)
        finally:
            if prev is not None:
                _drain_slot(ctx, prev, proof=True)

    # kzg.rs:182-185
    def commit_blob(self, blob, srs):
        """Bytes in, point out: bytes -> Fr, IFFT and MSM all on the device (`kzg_commit_blob`)."""
        ctx = self._ctx()
        data = blob.data()
        n_elems = -(-len(data) // 32)
        n = 1
        while n < n_elems:
            n <<= 1
        if n > len(srs):
            raise SrsCapacityExceeded(n, len(srs))
        buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
        out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0)
        rc = _lib.load().kzg_commit_blob(ctx.handle, srs.handle, buf.ctypes.data_as(_lib.u8p), len(data), _lib.ptr(out), C.byref(inf))
        if rc == _lib.ERR_SRS_CAPACITY_EXCEEDED:
            raise SrsCapacityExceeded(n, len(srs))
        if rc == _lib.ERR_TOO_LARGE:
            raise GenericError("Input size exceeds maximum polynomial size")
        ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(_lib.status_message(rc))
        return out

    # kzg.rs:128-178
    def _compute_proof_impl(self, polynomial, z_fr, srs, want_y=False):
        if len(polynomial) != len(self.expanded_roots_of_unity):
            raise GenericError("inconsistent length between blob and root of unities")
        if len(polynomial) > len(srs):
            raise SrsCapacityExceeded(len(polynomial), len(srs))
        ctx = self._ctx()
        evals = _lib.as_u64(polynomial.evaluations(), 4)
        roots = _lib.as_u64(self.expanded_roots_of_unity, 4)
        z = _lib.as_u64(z_fr, 0).reshape(4)
        out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0); y = np.zeros(4, dtype=np.uint64)
        rc = _lib.load().kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(evals), len(evals), _lib.ptr(roots), len(roots),
                                           _lib.ptr(z), _lib.ptr(out), C.byref(inf), _lib.ptr(y))
        if rc == _lib.ERR_ROOTS_LENGTH:
            raise GenericError("inconsistent length between blob and root of unities")
        ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(_lib.status_message(rc))
        return (out, y) if want_y else out

    # kzg.rs:215-234
    def compute_proof(self, polynomial, z_fr, srs):
        if len(polynomial) != len(self.expanded_roots_of_unity):
            raise GenericError("inconsistent length between blob and root of unities")
        return self._compute_proof_impl(polynomial, z_fr, srs)

    # kzg.rs:187-207
    def compute_proof_with_known_z_fr_index(self, polynomial, index: int, srs):
        if index < 0 or index >= 1 << 64:
            raise GenericError("Index conversion to usize failed")
        z = self.get_nth_root_of_unity(index)
        if z is None:
            raise GenericError("Root of unity not found")
        return self.compute_proof(polynomial, z, srs)

    # kzg.rs:237-260
    def compute_quotient_eval_on_domain(self, z_fr, eval_fr, value_fr):
        """sum over the stored roots w^i != z of (f_i - value) w^i / ((z - w^i) z): the quotient's evaluation at the domain point z, on the GPU
        (`kzg_compute_quotient_eval_on_domain`).  Like the reference it reads one evaluation per stored root."""
        n = len(self.expanded_roots_of_unity)
        ev = _lib.as_u64(eval_fr, 4).reshape(-1, 4)
        if len(ev) < n:
            raise IndexError("index out of bounds: the len is %d but the index is %d" % (len(ev), len(ev)))      # eval_fr[i] in the reference's loop
        if n == 0:
            return np.zeros(4, dtype=np.uint64)
        ctx = self._ctx()
        out = np.zeros(4, dtype=np.uint64)
        ev = np.ascontiguousarray(ev[:n])
        rc = _lib.load().kzg_compute_quotient_eval_on_domain(ctx.handle, _lib.ptr(_lib.as_u64(z_fr, 0).reshape(4)), _lib.ptr(ev), n,
                                                             _lib.ptr(_lib.as_u64(value_fr, 0).reshape(4)), _lib.ptr(out))
        if rc == _lib.ERR_INVALID_ARG and not _lib.as_u64(z_fr, 0).any():
            raise ZeroDivisionError("compute_quotient_eval_on_domain: z = 0 (the reference divides by z)")
        ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(_lib.status_message(rc))
        return out

    # kzg.rs:263-285
    def g1_ifft(self, length: int, srs):
        """Lagrange-basis SRS of size `length` (natural order), (length, 8) uint64 wire points."""
        if length <= 0 or (length & (length - 1)) != 0:
            raise FFTError("length provided is not a power of 2")
        ctx = self._ctx()
        out = np.zeros((length, 8), dtype=np.uint64)
        rc = _lib.load().kzg_g1_ifft(ctx.handle, srs.handle, length, _lib.ptr(out))
        if rc == _lib.ERR_NOT_POWER_OF_TWO:
            raise FFTError("length provided is not a power of 2")
        if rc == _lib.ERR_DOMAIN:
            raise FFTError("Could not perform IFFT due to domain consturction error")
        if rc == _lib.ERR_SRS_CAPACITY_EXCEEDED:
            raise SrsCapacityExceeded(length, len(srs))
        ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(_lib.status_message(rc))
        return out

    # kzg.rs:288-309
    def compute_blob_proof(self, blob, commitment, srs, want_zy=False):
        """Bytes in, proof out (`kzg_compute_blob_proof`): commitment validation, Fiat-Shamir challenge (hashed in C on a host
        thread beside the upload) and the proof in one call."""
        ctx = self._ctx()
        data = blob.data()
        buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
        out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0); z = np.zeros(4, dtype=np.uint64); y = np.zeros(4, dtype=np.uint64)
        rc = _lib.load().kzg_compute_blob_proof(ctx.handle, srs.handle, buf.ctypes.data_as(_lib.u8p), len(data), len(self.expanded_roots_of_unity),
                                                _lib.ptr(_lib.as_u64(commitment, 0).reshape(8)), _lib.ptr(out), C.byref(inf), _lib.ptr(z), _lib.ptr(y))
        self._raise_proof_status(ctx, rc, blob, srs)
        return (out, z, y) if want_zy else out

    def commit_and_prove_blob(self, blob, srs):
        """`commit_blob` + `compute_blob_proof` of the same blob in one call (`kzg_commit_and_prove_blob`): the transcript hash
        runs on a host thread while the GPU computes the commitment.  Returns (commitment, proof, z, y)."""
        ctx = self._ctx()
        data = blob.data()
        buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
        com = np.zeros(8, dtype=np.uint64); cinf = C.c_uint8(0)
        out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0); z = np.zeros(4, dtype=np.uint64); y = np.zeros(4, dtype=np.uint64)
        rc = _lib.load().kzg_commit_and_prove_blob(ctx.handle, srs.handle, buf.ctypes.data_as(_lib.u8p), len(data), len(self.expanded_roots_of_unity),
                                                   _lib.ptr(com), C.byref(cinf), _lib.ptr(out), C.byref(inf), _lib.ptr(z), _lib.ptr(y))
        self._raise_proof_status(ctx, rc, blob, srs)
        return com, out, z, y

    # the same as a stream (`kzg_commit_and_prove_blob_begin` / `_end`): up to _lib.BLOB_JOBS blobs in flight, their transcript hashes side by side
    def commit_and_prove_blob_begin(self, blob, srs, job, commitment=None):
        """Starts job `job` (0 .. BLOB_JOBS-1): the transcript prefix on a host thread of the library, upload + bytes -> Fr + commitment on the
        GPU; returns at once.  `commitment` given: `compute_blob_proof` as it stands (validated, absorbed, not recomputed).  The blob's bytes are
        kept alive here until the job's end call."""
        ctx = self._ctx()
        data = blob.data()
        buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
        cptr = _lib.ptr(_lib.as_u64(commitment, 0).reshape(8)) if commitment is not None else None
        rc = _lib.load().kzg_commit_and_prove_blob_begin(ctx.handle, srs.handle, buf.ctypes.data_as(_lib.u8p), len(data), len(self.expanded_roots_of_unity),
                                                         cptr, job)
        self._raise_proof_status(ctx, rc, blob, srs)
        if not hasattr(self, "_blob_jobs"):
            self._blob_jobs = {}
        self._blob_jobs[job] = (buf, data, blob, srs)

    def commit_and_prove_blob_end(self, job):
        """(commitment, proof, z, y) of job `job`; waits for what is still missing and moves the other jobs on meanwhile."""
        ctx = self._ctx()
        com = np.zeros(8, dtype=np.uint64); cinf = C.c_uint8(0)
        out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0); z = np.zeros(4, dtype=np.uint64); y = np.zeros(4, dtype=np.uint64)
        rc = _lib.load().kzg_commit_and_prove_blob_end(ctx.handle, job, _lib.ptr(com), C.byref(cinf), _lib.ptr(out), C.byref(inf), _lib.ptr(z), _lib.ptr(y))
        held = getattr(self, "_blob_jobs", {}).pop(job, None)
        if rc != _lib.OK and held is not None:
            self._raise_proof_status(ctx, rc, held[2], held[3])
        ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(ctx.last_error() or _lib.status_message(rc))
        return com, out, z, y

    def commit_and_prove_blobs(self, blobs, srs, inflight=8):
        """Generator over an iterable of blobs: yields (commitment, proof, z, y) per blob, in order, with `inflight` jobs in flight."""
        inflight = max(1, min(int(inflight), _lib.BLOB_JOBS))
        pending = []                                   # job indices in begin order
        free = list(range(inflight))
        try:
            for blob in blobs:
                if not free:
                    j = pending.pop(0)
                    res = self.commit_and_prove_blob_end(j)
                    free.append(j)
                    yield res
                j = free.pop(0)
                self.commit_and_prove_blob_begin(blob, srs, j)
                pending.append(j)
            while pending:
                j = pending.pop(0)
                res = self.commit_and_prove_blob_end(j)
                free.append(j)
                yield res
        finally:
            for j in pending:                          # (an exception or an abandoned generator: nothing stays in flight)
                try:
                    self.commit_and_prove_blob_end(j)
                except Exception:
                    pass

    @staticmethod
    def _raise_proof_status(ctx, rc, blob, srs):
        if rc == _lib.OK:
            return
        if rc == _lib.ERR_G1_NOT_ON_CURVE:
            raise NotOnCurveError("G1 point not on curve")
        if rc == _lib.ERR_ROOTS_LENGTH:
            raise GenericError("inconsistent length between blob and root of unities")
        if rc == _lib.ERR_SRS_CAPACITY_EXCEEDED:
            raise SrsCapacityExceeded(-(-len(blob.data()) // 32), len(srs))
        if rc == _lib.ERR_TOO_LARGE:
            raise GenericError("Input size exceeds maximum polynomial size")
        ctx.check_device(rc)
        raise GenericError(_lib.status_message(rc))
