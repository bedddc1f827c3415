"""Mirror of prover/src/srs.rs `SRS { g1, order }`, with the monomial G1 powers RESIDENT ON THE GPU
(uploaded once through `kzg_srs_upload`) instead of being copied on every commit (kzg.rs:119).
`SRS.new(path, order, points_to_load)` reads the gnark-format file and decompresses all points in one GPU kernel
(`kzg_srs_load_compressed_be`); `SRS(points)` takes already-decoded wire-format points."""
import ctypes as C
import os

import numpy as np

from . import _lib
from .errors import DeserializationError, GenericError, NotOnCurveError
from .fr import fr_from_int


class SRS:
    def __init__(self, g1_points, order=None, ctx=None):
        """g1_points: (n, 8) uint64 wire-format affine points (x || y, Montgomery)."""
        self.ctx = ctx or _lib.default_context()
        pts = _lib.as_u64(g1_points, 8).reshape(-1, 8)
        self.order = int(order if order is not None else pts.shape[0])
        if pts.shape[0] > self.order:
            raise GenericError("Number of points to load exceeds SRS order.")       # srs.rs:36-40
        h = C.c_void_p()
        rc = _lib.load().kzg_srs_upload(self.ctx.handle, _lib.ptr(pts) if pts.shape[0] else None, pts.shape[0], C.byref(h))
        self.ctx.check_device(rc)
        self.handle = h
        self._n = pts.shape[0]

    @classmethod
    def new(cls, path_to_g1_points, order, points_to_load, ctx=None, is_native=False):
        """SRS::new (srs.rs:35-49): read `points_to_load` compressed points (32 bytes each, gnark big-endian flags) and
        decompress them on the GPU (`kzg_srs_load_compressed_be`).  is_native=True: the arkworks little-endian compressed
        format of `parallel_read_g1_points_native(.., is_native = true)` (srs.rs:205-251; `kzg_srs_load_compressed_ark_le`)."""
        if points_to_load > order:
            raise GenericError("Number of points to load exceeds SRS order.")               # srs.rs:36-40
        with open(path_to_g1_points, "rb") as f:
            data = f.read(32 * points_to_load)
        if len(data) != 32 * points_to_load:                                                  # srs.rs:128-134
            raise GenericError(f"Expected {points_to_load} points, only read {len(data) // 32}")
        self = cls.__new__(cls)
        self.ctx = ctx or _lib.default_context()
        self.order = int(order)
        h = C.c_void_p()
        bad = C.c_uint64(0)
        buf = np.frombuffer(data, dtype=np.uint8) if data else np.zeros(1, np.uint8)
        load = _lib.load().kzg_srs_load_compressed_ark_le if is_native else _lib.load().kzg_srs_load_compressed_be
        rc = load(self.ctx.handle, buf.ctypes.data_as(_lib.u8p), points_to_load, C.byref(h), C.byref(bad))
        if rc == _lib.ERR_DESERIALIZE:
            raise DeserializationError("Deserialization failed" if is_native else "point at infinity not coded properly for g1")   # traits.rs:17 / helpers.rs:191-195
        if rc == _lib.ERR_NOT_ON_CURVE and is_native:
            raise DeserializationError("Deserialization failed")          # G1Affine::deserialize_compressed reports every bad encoding alike (traits.rs:34-36)
        if rc == _lib.ERR_NOT_ON_CURVE:
            chunk = list(data[32 * bad.value:32 * bad.value + 32])
            raise NotOnCurveError(f"compressed g1 point not on curve: {chunk}")
        self.ctx.check_device(rc)
        self.handle = h
        self._n = points_to_load
        return self

    @staticmethod
    def parallel_read_g1_points_native(file_path, points_to_load, is_native, ctx=None):
        """srs.rs:205-251 (and :76-126 for is_native = False): the first `points_to_load` points of the file as (n, 8) wire-format points.  The
        reference spreads the 32-byte chunks over one worker per core; here one kernel decodes them all and the points are read back."""
        srs = SRS.new(file_path, points_to_load, points_to_load, ctx=ctx, is_native=is_native)
        try:
            return srs.g1
        finally:
            srs.close()

    @staticmethod
    def process_chunks(receiver, ctx=None):
        """srs.rs:51-70: `receiver` yields (chunk, position, is_native) -> [(point, position)], in the order received.  The chunks of each format go
        through the decoding kernel together; a chunk that does not decode raises (the reference panics: "Failed to read point from bytes")."""
        items = list(receiver)
        out = [None] * len(items)
        for native in (False, True):
            idx = [k for k, it in enumerate(items) if bool(it[2]) == native]
            if not idx:
                continue
            data = b"".join(bytes(items[k][0]) for k in idx)
            if any(len(items[k][0]) != 32 for k in idx):
                raise DeserializationError("not enough bytes for g1 point")
            lib = _lib.load()
            c = ctx or _lib.default_context()
            h = C.c_void_p()
            bad = C.c_uint64(0)
            buf = np.frombuffer(data, dtype=np.uint8)
            load = lib.kzg_srs_load_compressed_ark_le if native else lib.kzg_srs_load_compressed_be
            rc = load(c.handle, buf.ctypes.data_as(_lib.u8p), len(idx), C.byref(h), C.byref(bad))
            if rc in (_lib.ERR_DESERIALIZE, _lib.ERR_NOT_ON_CURVE):
                raise DeserializationError("Failed to read point from bytes")
            c.check_device(rc)
            pts = np.zeros((len(idx), 8), dtype=np.uint64)
            try:
                c.check_device(lib.kzg_srs_download(c.handle, h, 0, len(idx), _lib.ptr(pts)))
            finally:
                lib.kzg_srs_free(h)
            for j, k in enumerate(idx):
                out[k] = (pts[j], items[k][1])
        return out

    def save_packed(self, path: str):
        """Write the decoded points in the library's packed form (`kzg_srs_save_packed`): `SRS.load_packed` reads them back without decoding
        the ceremony file again (digest-checked, curve-checked on the GPU)."""
        rc = _lib.load().kzg_srs_save_packed(self.ctx.handle, self.handle, os.fsencode(path))
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(f"{_lib.status_message(rc)}: {path}")

    @classmethod
    def load_packed(cls, path: str, points_to_load: int = 0, order=None, ctx=None):
        """`kzg_srs_load_packed`: the first `points_to_load` points (0: all) of a file written by `save_packed`."""
        self = cls.__new__(cls)
        self.ctx = ctx or _lib.default_context()
        h = C.c_void_p()
        rc = _lib.load().kzg_srs_load_packed(self.ctx.handle, os.fsencode(path), points_to_load, C.byref(h))
        if rc == _lib.ERR_DESERIALIZE:
            raise DeserializationError(f"{path}: not a packed SRS file, or damaged")
        if rc == _lib.ERR_NOT_ON_CURVE:
            raise NotOnCurveError(f"{path}: a point of the packed SRS is not on the curve")
        if rc == _lib.ERR_SRS_LENGTH:
            raise GenericError("Number of points to load exceeds SRS order.")               # srs.rs:36-40
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(f"{_lib.status_message(rc)}: {path}")
        self.handle = h
        self._n = int(_lib.load().kzg_srs_len(h))
        self.order = int(order if order is not None else self._n)
        return self

    @classmethod
    def generate(cls, tau: int, n: int, ctx=None, first_power: int = 0):
        """Synthetic SRS with known tau: P_i = tau^(first_power + i) * G1, generated on the device (tests / bench)."""
        self = cls.__new__(cls)
        self.ctx = ctx or _lib.default_context()
        self.order = n
        h = C.c_void_p()
        t = fr_from_int(tau)
        rc = _lib.load().kzg_srs_generate(self.ctx.handle, _lib.ptr(t), first_power, n, C.byref(h))
        self.ctx.check_device(rc)
        self.handle = h
        self._n = n
        return self

    @property
    def g1(self):
        """The points, read back from the device in wire format."""
        out = np.zeros((self._n, 8), dtype=np.uint64)
        if self._n:
            rc = _lib.load().kzg_srs_download(self.ctx.handle, self.handle, 0, self._n, _lib.ptr(out))
            self.ctx.check_device(rc)
        return out

    def __len__(self):
        return self._n

    def cache_lagrange(self, n: int):
        """Compute KZG::g1_ifft(n) once and keep it on the device (`kzg_srs_cache_lagrange`): `commit_eval_form` of exactly n
        evaluations then is one MSM over it, as in the reference (kzg.rs:98-100), instead of IFFT + MSM."""
        rc = _lib.load().kzg_srs_cache_lagrange(self.ctx.handle, self.handle, n)
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(_lib.status_message(rc))

    def drop_lagrange(self):
        _lib.load().kzg_srs_drop_lagrange(self.ctx.handle, self.handle)

    def lagrange(self, n: int) -> "SRS":
        """The Lagrange basis of the first n points as an SRS of its own (`kzg_srs_lagrange`)."""
        h = C.c_void_p()
        rc = _lib.load().kzg_srs_lagrange(self.ctx.handle, self.handle, n, C.byref(h))
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(_lib.status_message(rc))
        out = SRS.__new__(SRS)
        out.ctx, out.order, out.handle, out._n = self.ctx, n, h, n
        return out

    def lagrange_shard(self, n: int, lo: int, length: int) -> "SRS":
        """The points [lo, lo + length) of the Lagrange basis of the first n points as an SRS of their own (`kzg_srs_lagrange_shard`): one
        rank's shard of the basis for the evaluation-index sharding of BASELINE config 4 (sharding.ShardedKzgLagrange)."""
        h = C.c_void_p()
        rc = _lib.load().kzg_srs_lagrange_shard(self.ctx.handle, self.handle, n, lo, length, C.byref(h))
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(_lib.status_message(rc))
        out = SRS.__new__(SRS)
        out.ctx, out.order, out.handle, out._n = self.ctx, length, h, length
        return out

    def slice(self, lo: int, length: int) -> "SRS":
        """A copy of the points [lo, lo + length) as an SRS of their own, with its own tables (`kzg_srs_slice`)."""
        h = C.c_void_p()
        rc = _lib.load().kzg_srs_slice(self.ctx.handle, self.handle, lo, length, C.byref(h))
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise GenericError(_lib.status_message(rc))
        out = SRS.__new__(SRS)
        out.ctx, out.order, out.handle, out._n = self.ctx, length, h, length
        return out

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().kzg_srs_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
