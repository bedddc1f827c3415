"""Mirror of prover/src/srs.rs `SRS { g1, order }`, with the monomial G1 powers RESIDENT ON THE GPU
(uploaded once through `kzg_srs_upload`) instead of being copied on every commit (kzg.rs:119).
File loading / point decompression (srs.rs:81-251) stays a host-side concern and is listed as a
"next" row in DESIGN.md; construct from already-decoded points."""
import ctypes as C

import numpy as np

from . import _lib
from .errors import GenericError
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
    def new(cls, path_to_g1_points, order, points_to_load, ctx=None):
        if points_to_load > order:
            raise GenericError("Number of points to load exceeds SRS order.")
        raise NotImplementedError("SRS file loading is a 'next' row (DESIGN.md §9); build the SRS from decoded points")

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

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().kzg_srs_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
