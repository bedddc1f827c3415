"""ctypes binding of libkzg_bn254_mi355x.so — the C-ABI declared in include/kzg_bn254_mi355x.h.

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950).  If it is
missing, or no HIP device is usable, every operation FAILS LOUDLY (DeviceError): there is no CPU
fallback and nothing here imports oracle/.
"""
import ctypes as C
import os

import numpy as np

from .errors import DeviceError

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KZG_LIB_PATH") or os.path.join(HERE, "libkzg_bn254_mi355x.so")   # KZG_LIB_PATH: A/B builds of the same library (tools/)

# status codes (include/kzg_bn254_mi355x.h)
OK = 0
ERR_INVALID_ARG = -1
ERR_NO_DEVICE = -2
ERR_DEVICE = -3
ERR_MSM_LENGTH_MISMATCH = -4
ERR_SRS_CAPACITY_EXCEEDED = -5
ERR_POLY_LENGTH = -6
ERR_NOT_POWER_OF_TWO = -7
ERR_DOMAIN = -8
ERR_ROOTS_LENGTH = -9
ERR_INVALID_INPUT_LENGTH = -10
ERR_TOO_LARGE = -11
ERR_ROOT_NOT_FOUND = -12
ERR_ZERO_LENGTH = -13
ERR_SRS_LENGTH = -14
ERR_DESERIALIZE = -15
ERR_NOT_ON_CURVE = -16
BLOB_JOBS = 16                                                   # KZG_BLOB_JOBS of the header
NUM_SLOTS = int(os.environ.get("KZG_NUM_SLOTS", "4"))          # KZG_NUM_SLOTS of the header (env: variant builds with more slots)
ERR_G1_NOT_ON_CURVE = -17
ERR_G2_TAU_NOT_ON_CURVE = -18
ERR_TAU_EQUALS_Z = -19
ERR_PEER = -20
ERR_EXCHANGE_TIMEOUT = -21
ERR_IO = -22

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p
sz = C.c_size_t
i32 = C.c_int32

# name -> (restype, argtypes); every symbol include/kzg_bn254_mi355x.h declares
PROTOTYPES = {
    "kzg_status_message": (C.c_char_p, [i32]),
    "kzg_device_count": (i32, []),
    "kzg_ctx_create": (i32, [i32, C.POINTER(vp)]),
    "kzg_ctx_destroy": (None, [vp]),
    "kzg_ctx_last_error": (C.c_char_p, [vp]),
    "kzg_ctx_set_msm_window": (i32, [vp, i32, i32]),
    "kzg_ctx_set_reduction_lanes": (i32, [vp, i32]),
    "kzg_srs_upload": (i32, [vp, u64p, sz, C.POINTER(vp)]),
    "kzg_srs_load_compressed_be": (i32, [vp, u8p, sz, C.POINTER(vp), C.POINTER(C.c_uint64)]),
    "kzg_srs_load_compressed_ark_le": (i32, [vp, u8p, sz, C.POINTER(vp), C.POINTER(C.c_uint64)]),
    "kzg_srs_has_bit_tables": (i32, [vp, i32]),
    "kzg_srs_generate": (i32, [vp, u64p, C.c_uint64, sz, C.POINTER(vp)]),
    "kzg_ctx_set_profiling": (i32, [vp, i32]),
    "kzg_ctx_get_msm_profile": (i32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "kzg_ctx_get_msm_profile_entries": (i32, [vp, C.POINTER(C.c_uint64)]),
    "kzg_ctx_measure_valu_rates": (i32, [vp, i32, C.POINTER(C.c_double)]),
    "kzg_srs_download": (i32, [vp, vp, sz, sz, u64p]),
    "kzg_srs_save_packed": (i32, [vp, vp, C.c_char_p]),
    "kzg_srs_load_packed": (i32, [vp, C.c_char_p, sz, C.POINTER(vp)]),
    "kzg_srs_free": (None, [vp]),
    "kzg_srs_len": (sz, [vp]),
    "kzg_msm_g1": (i32, [vp, u64p, sz, u64p, sz, u64p, u8p]),
    "kzg_msm_g1_batch": (i32, [vp, u64p, u64p, sz, sz, u64p, u8p]),
    "kzg_msm_g1_srs": (i32, [vp, vp, sz, u64p, sz, u64p, u8p]),
    "kzg_msm_g1_srs_device": (i32, [vp, vp, sz, vp, sz, u64p, u8p]),
    "kzg_msm_g1_srs_partial_device": (i32, [vp, vp, sz, vp, sz, u64p]),
    "kzg_msm_g1_srs_partial": (i32, [vp, vp, sz, u64p, sz, u64p]),
    "kzg_msm_g1_srs_device_begin": (i32, [vp, vp, sz, vp, sz, i32]),
    "kzg_msm_g1_srs_begin": (i32, [vp, vp, sz, u64p, sz, i32]),
    "kzg_msm_g1_srs_end": (i32, [vp, i32, u64p, u8p, u64p]),
    "kzg_g1_fold_partials": (i32, [u64p, sz, u64p, u8p]),
    "kzg_fr_ntt": (i32, [vp, u64p, sz, i32]),
    "kzg_fr_ntt_device": (i32, [vp, vp, sz, i32]),
    "kzg_commit_coeff_form": (i32, [vp, vp, u64p, sz, u64p, u8p]),
    "kzg_commit_eval_form": (i32, [vp, vp, u64p, sz, u64p, u8p]),
    "kzg_msm_batch_capacity": (sz, [sz]),
    "kzg_msm_g1_srs_device_begin_batch": (i32, [vp, vp, sz, C.POINTER(C.c_void_p), sz, sz, i32]),
    "kzg_msm_g1_srs_end_batch": (i32, [vp, i32, sz, u64p, u8p, u64p]),
    "kzg_commit_coeff_form_batch": (i32, [vp, vp, u64p, sz, sz, u64p, u8p]),
    "kzg_commit_coeff_form_batch_device": (i32, [vp, vp, vp, sz, sz, u64p, u8p]),
    "kzg_commit_eval_form_batch": (i32, [vp, vp, u64p, sz, sz, u64p, u8p]),
    "kzg_commit_eval_form_batch_device": (i32, [vp, vp, vp, sz, sz, u64p, u8p]),
    "kzg_commit_eval_form_begin": (i32, [vp, vp, u64p, sz, i32]),
    "kzg_commit_blob_begin": (i32, [vp, vp, u8p, sz, i32]),
    "kzg_g1_ifft": (i32, [vp, vp, sz, u64p]),
    "kzg_srs_cache_lagrange": (i32, [vp, vp, sz]),
    "kzg_srs_lagrange": (i32, [vp, vp, sz, C.POINTER(vp)]),
    "kzg_srs_drop_lagrange": (i32, [vp, vp]),
    "kzg_blob_to_fr": (i32, [vp, u8p, sz, u64p, sz, C.POINTER(sz)]),
    "kzg_commit_blob": (i32, [vp, vp, u8p, sz, u64p, u8p]),
    "kzg_compute_proof": (i32, [vp, vp, u64p, sz, u64p, sz, u64p, u64p, u8p, u64p]),
    "kzg_compute_proof_begin": (i32, [vp, vp, u64p, sz, u64p, sz, u64p, i32]),
    "kzg_compute_proof_end": (i32, [vp, i32, u64p, u8p, u64p]),
    "kzg_multi_create": (i32, [C.POINTER(i32), i32, C.POINTER(vp)]),
    "kzg_multi_destroy": (None, [vp]),
    "kzg_multi_device_count": (i32, [vp]),
    "kzg_multi_srs_len": (sz, [vp]),
    "kzg_multi_srs_upload": (i32, [vp, u64p, sz]),
    "kzg_multi_srs_generate": (i32, [vp, u64p, sz]),
    "kzg_multi_cache_lagrange": (i32, [vp, sz]),
    "kzg_multi_commit_coeff_form": (i32, [vp, u64p, sz, u64p, u8p]),
    "kzg_multi_commit_eval_form": (i32, [vp, u64p, sz, u64p, u8p]),
    "kzg_multi_compute_proof": (i32, [vp, u64p, sz, sz, u64p, u64p, u8p, u64p]),
    "kzg_multi_scalars_upload": (i32, [vp, i32, u64p, sz]),
    "kzg_multi_commit_resident_stream": (i32, [vp, C.POINTER(i32), sz, u64p, u8p]),
    "kzg_compute_challenge": (i32, [u8p, sz, u64p, u64p]),
    "kzg_compute_blob_proof": (i32, [vp, vp, u8p, sz, sz, u64p, u64p, u8p, u64p, u64p]),
    "kzg_commit_and_prove_blob": (i32, [vp, vp, u8p, sz, sz, u64p, u8p, u64p, u8p, u64p, u64p]),
    "kzg_commit_and_prove_blob_begin": (i32, [vp, vp, u8p, sz, sz, u64p, i32]),
    "kzg_commit_and_prove_blob_end": (i32, [vp, i32, u64p, u8p, u64p, u8p, u64p, u64p]),
    "kzg_commit_eval_form_partial": (i32, [vp, vp, sz, u64p, sz, u64p]),
    "kzg_compute_proof_partial": (i32, [vp, vp, sz, u64p, sz, u64p, sz, u64p, u64p, u64p]),
    "kzg_evaluate_polynomial_in_evaluation_form": (i32, [vp, u64p, sz, u64p, u64p]),
    "kzg_calculate_roots_of_unity": (i32, [vp, C.c_uint64, u64p, sz, C.POINTER(sz)]),
    "kzg_g2_generator": (i32, [u64p]),
    "kzg_g2_tau_mainnet": (i32, [u64p]),
    "kzg_g2_mul_generator": (i32, [u64p, u64p]),
    "kzg_validate_g1_point": (i32, [u64p]),
    "kzg_hash_to_field_element": (i32, [u8p, C.c_size_t, u64p]),
    "kzg_g2_is_on_curve": (i32, [u64p, C.POINTER(i32)]),
    "kzg_validate_g2_point": (i32, [u64p, C.POINTER(i32)]),
    "kzg_compute_quotient_eval_on_domain": (i32, [vp, u64p, u64p, C.c_size_t, u64p, u64p]),
    "kzg_pairings_verify": (i32, [u64p, u64p, u64p, u64p, C.POINTER(i32)]),
    "kzg_verify_proof": (i32, [u64p, u64p, u64p, u64p, u64p, C.POINTER(i32)]),
    "kzg_verify_kzg_proof_batch": (i32, [vp, u64p, u64p, u64p, u64p, u64p, sz, u64p, C.POINTER(i32)]),
    "kzg_compute_r_powers": (i32, [u64p, u64p, u64p, u64p, u64p, sz, u64p]),
    "kzg_compute_challenges_and_evaluate_polynomial": (i32, [vp, C.POINTER(C.c_char_p), C.POINTER(sz), u64p, sz, u64p, u64p]),
    "kzg_evaluate_blobs_in_evaluation_form_batch": (i32, [vp, C.POINTER(C.c_char_p), C.POINTER(sz), u64p, sz, u64p]),
    "kzg_rccl_allgather_fold": (i32, [vp, vp, i32, u64p, u64p, u8p]),
    "kzg_commit_coeff_form_rccl": (i32, [vp, vp, vp, sz, vp, i32, u64p, u8p]),
    "kzg_srs_slice": (i32, [vp, vp, sz, sz, C.POINTER(vp)]),
    "kzg_srs_lagrange_shard": (i32, [vp, vp, sz, sz, sz, C.POINTER(vp)]),
    "kzg_commit_eval_form_lagrange_partial": (i32, [vp, vp, u64p, sz, u64p]),
    "kzg_commit_eval_form_lagrange_partial_device": (i32, [vp, vp, vp, sz, u64p]),
    "kzg_compute_proof_lagrange_begin": (i32, [vp, vp, sz, u64p, sz, sz, u64p, i32]),
    "kzg_compute_proof_lagrange_begin_device": (i32, [vp, vp, sz, vp, sz, sz, u64p, i32]),
    "kzg_commit_and_prove_lagrange_begin": (i32, [vp, vp, sz, u64p, sz, sz, u64p, i32, i32]),
    "kzg_commit_and_prove_lagrange_begin_device": (i32, [vp, vp, sz, vp, sz, sz, u64p, i32, i32]),
    "kzg_commit_and_prove_lagrange_end": (i32, [vp, i32, u64p, u64p]),
    "kzg_compute_proof_lagrange_partial_y": (i32, [vp, i32, u64p]),
    "kzg_compute_proof_lagrange_continue": (i32, [vp, i32, u64p]),
    "kzg_compute_proof_lagrange_end": (i32, [vp, i32, u64p]),
    "kzg_compute_proof_lagrange_abort": (i32, [vp, i32]),
    "kzg_lagrange_fold_y": (i32, [u64p, sz, sz, u64p, u64p]),
    "kzg_lagrange_fold_proof": (i32, [u64p, sz, sz, u64p, u64p, u8p]),
    "kzg_commit_eval_form_rccl": (i32, [vp, vp, u64p, sz, vp, i32, u64p, u8p]),
    "kzg_commit_eval_form_rccl_device": (i32, [vp, vp, vp, sz, vp, i32, u64p, u8p]),
    "kzg_compute_proof_rccl": (i32, [vp, vp, sz, u64p, sz, sz, u64p, vp, i32, u64p, u8p, u64p]),
    "kzg_compute_proof_rccl_device": (i32, [vp, vp, sz, vp, sz, sz, u64p, vp, i32, u64p, u8p, u64p]),
    "kzg_verify_blob_kzg_proof": (i32, [vp, u8p, sz, u64p, u64p, u64p, C.POINTER(i32)]),
    "kzg_verify_blob_kzg_proof_batch": (i32, [vp, C.POINTER(C.c_char_p), C.POINTER(sz), u64p, u64p, sz, u64p, C.POINTER(i32)]),
}

_lib = None


def load():
    """Load the shared library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DeviceError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950). No CPU fallback exists.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def status_message(status: int) -> str:
    return load().kzg_status_message(status).decode()


def as_u64(a, cols):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim == 1 and cols and a.size % cols == 0:
        a = a.reshape(-1, cols)
    return a


def ptr(a):
    return a.ctypes.data_as(u64p)


def blob_args(blobs):
    """(char* array, size_t array, keep-alive) for the `const uint8_t* const* blobs, const size_t* blob_lens` arguments of the batch
    verification calls; `blobs` = objects with .data() (Blob) or bytes."""
    datas = [b.data() if hasattr(b, "data") else bytes(b) for b in blobs]
    n = len(datas)
    ptrs = (C.c_char_p * max(n, 1))(*datas)
    lens = (sz * max(n, 1))(*[len(d) for d in datas])
    return ptrs, lens, datas


class Context:
    """One GPU.  `kzg_ctx` of the C-ABI."""

    def __init__(self, device_id=0):
        lib = load()
        h = vp()
        rc = lib.kzg_ctx_create(device_id, C.byref(h))
        if rc != OK:
            raise DeviceError(f"kzg_ctx_create(device {device_id}) failed: {status_message(rc)}")
        self.handle = h
        self.device_id = device_id

    def last_error(self):
        return load().kzg_ctx_last_error(self.handle).decode()

    def check_device(self, rc):
        if rc in (ERR_DEVICE, ERR_NO_DEVICE):
            raise DeviceError(f"{status_message(rc)}: {self.last_error()}")
        if rc == ERR_INVALID_ARG:
            raise ValueError("invalid argument passed to the C-ABI")

    def set_msm_window(self, c_bits=0, segment_len=0):
        rc = load().kzg_ctx_set_msm_window(self.handle, c_bits, segment_len)
        if rc != OK:
            raise ValueError("window bits must be 0 (auto) or in [2,16]")

    def set_reduction_lanes(self, lanes=0):
        """Lanes per point of the bucket-reduction kernels: 0 automatic, 2 pairs, 4 quads (same results; test / measurement hook)."""
        if load().kzg_ctx_set_reduction_lanes(self.handle, lanes) != OK:
            raise ValueError("lanes must be 0, 2 or 4")

    def close(self):
        if self.handle:
            load().kzg_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device_id=None):
    """Process-wide context for `device_id` (default: LOCAL_RANK or 0)."""
    if device_id is None:
        device_id = int(os.environ.get("KZG_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if device_id not in _default_ctx:
        _default_ctx[device_id] = Context(device_id)
    return _default_ctx[device_id]
