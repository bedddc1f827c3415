// kzg_bn254_mi355x.hpp -- C++17 host mirror of the reference's Rust API above the C-ABI (kzg_bn254_mi355x.h): header only.
//
// The reference is compiled Rust and this image has no Rust toolchain, so the host side a compiled program would use is given here in
// C++: the same types, method names, argument meaning and error behaviour as
//   prover/src/kzg.rs (KZG), prover/src/srs.rs (SRS), primitives/src/blob.rs (Blob), primitives/src/polynomial.rs
//   (PolynomialEvalForm / PolynomialCoeffForm), primitives/src/errors.rs (KzgError), verifier/src/verify.rs and batch.rs,
// so that tests/cpp/reference_tests.cpp reads like prover/tests/kzg_test.rs and verifier/tests/tests.rs.  Every method is a thin
// wrapper: the arithmetic runs in libkzg_bn254_mi355x.so (HIP kernels; no CPU fallback -- without a GPU the first call throws).
// `new` is a C++ keyword: Rust's `T::new(..)` is `T::new_(..)` here.  Rust's Result<T, KzgError> is a return value or a thrown KzgError
// whose what() is the reference's Display text.
//
// Wire formats: Fr = arkworks' in-memory 4 x u64 Montgomery limbs; G1Affine = x || y (8 u64), identity = zeros;
// G2Affine = x.c0 | x.c1 | y.c0 | y.c1 (16 u64).  The few host-side Fr operations below (conversion from / to canonical big-endian
// bytes) are data-format helpers, not part of the accelerated path.
#pragma once
#include "kzg_bn254_mi355x.h"

#include <array>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rust_kzg_bn254 {

constexpr size_t BYTES_PER_FIELD_ELEMENT = 32;                        // primitives/src/consts.rs:4
constexpr size_t MAINNET_SRS_G1_SIZE = 268435456;                     // primitives/src/consts.rs:66

// ---- errors (primitives/src/errors.rs:13-24, :32-86) ---------------------------------------------------------------------------------
class KzgError : public std::runtime_error {
public:
    enum class Kind { PolynomialError, MsmError, SerializationError, DeserializationError, SrsCapacityExceeded, G2GeneratorNotAcceptedError, NotOnCurveError,
                      CommitError, FFTError, GenericError, InvalidDenominator, InvalidInputLength, InvalidFieldElement, DeviceError };
    Kind kind;
    std::string message;                                              // the variant's payload (without the Display prefix)
    size_t polynomial_len = 0, srs_len = 0;                           // SrsCapacityExceeded only
    KzgError(Kind k, const std::string& payload) : std::runtime_error(prefix(k) + payload), kind(k), message(payload) {}
    static KzgError GenericError(const std::string& m) { return KzgError(Kind::GenericError, m); }
    static KzgError FFTError(const std::string& m) { return KzgError(Kind::FFTError, m); }
    static KzgError CommitError(const std::string& m) { return KzgError(Kind::CommitError, m); }
    static KzgError SerializationError(const std::string& m) { return KzgError(Kind::SerializationError, m); }
    static KzgError NotOnCurveError(const std::string& m) { return KzgError(Kind::NotOnCurveError, m); }
    static KzgError MsmError(const std::string& m) { return KzgError(Kind::MsmError, m); }
    static KzgError DeserializationError(const std::string& m) { return KzgError(Kind::DeserializationError, m); }
    static KzgError G2GeneratorNotAcceptedError(const std::string& m) { return KzgError(Kind::G2GeneratorNotAcceptedError, m); }
    static KzgError InvalidFieldElement(const std::string& m) { return KzgError(Kind::InvalidFieldElement, m); }
    static KzgError InvalidDenominator() { return KzgError(Kind::InvalidDenominator, "invalid denominator"); }
    static KzgError InvalidInputLength() { return KzgError(Kind::InvalidInputLength, "input length must be a multiple of 32"); }
    static KzgError SrsCapacityExceeded(size_t polynomial_len, size_t srs_len) {
        KzgError e(Kind::SrsCapacityExceeded, "polynomial degree " + std::to_string(polynomial_len) + " exceeds SRS capacity " + std::to_string(srs_len));
        e.polynomial_len = polynomial_len; e.srs_len = srs_len;
        return e;
    }
    bool operator==(const KzgError& o) const { return kind == o.kind && message == o.message; }
private:
    static std::string prefix(Kind k) {
        switch (k) {
            case Kind::PolynomialError: return "polynomial error: ";
            case Kind::MsmError: return "MSM error: ";
            case Kind::SerializationError: return "serialization error: ";
            case Kind::DeserializationError: return "deserialization error: ";
            case Kind::G2GeneratorNotAcceptedError: return "g2 generator not accepted error: ";
            case Kind::NotOnCurveError: return "not on curve error: ";
            case Kind::CommitError: return "commit error: ";
            case Kind::FFTError: return "FFT error: ";
            case Kind::GenericError: return "generic error: ";
            case Kind::InvalidFieldElement: return "invalid field element: ";
            case Kind::DeviceError: return "device error: ";
            default: return "";                                       // SrsCapacityExceeded, InvalidInputLength, InvalidDenominator: the payload is the text
        }
    }
};

namespace detail {
// status of the C-ABI -> the reference's error (Appendix B of SURVEY.md; kzg_status in the C header); OK returns
inline void check(int32_t rc, const kzg_ctx* ctx = nullptr, size_t polynomial_len = 0, size_t srs_len = 0) {
    if (rc == KZG_OK) return;
    const std::string text = kzg_status_message(rc);
    switch (rc) {
        case KZG_ERR_MSM_LENGTH_MISMATCH: throw KzgError::CommitError(std::to_string(polynomial_len < srs_len ? polynomial_len : srs_len));   // kzg.rs:102, :123
        case KZG_ERR_SRS_CAPACITY_EXCEEDED: throw KzgError::SrsCapacityExceeded(polynomial_len, srs_len);                                    // kzg.rs:89-94
        case KZG_ERR_POLY_LENGTH: throw KzgError::SerializationError(text);
        case KZG_ERR_NOT_POWER_OF_TWO: case KZG_ERR_DOMAIN: throw KzgError::FFTError(text);
        case KZG_ERR_INVALID_INPUT_LENGTH: throw KzgError::InvalidInputLength();
        case KZG_ERR_DESERIALIZE: throw KzgError(KzgError::Kind::DeserializationError, text);
        case KZG_ERR_NOT_ON_CURVE: case KZG_ERR_G1_NOT_ON_CURVE: case KZG_ERR_G2_TAU_NOT_ON_CURVE: throw KzgError::NotOnCurveError(text);
        case KZG_ERR_ROOTS_LENGTH: case KZG_ERR_TOO_LARGE: case KZG_ERR_ROOT_NOT_FOUND: case KZG_ERR_ZERO_LENGTH: case KZG_ERR_SRS_LENGTH:
        case KZG_ERR_TAU_EQUALS_Z: throw KzgError::GenericError(text);
        default: {                                                    // INVALID_ARG / NO_DEVICE / DEVICE: no counterpart in the reference
            std::string m = text;
            if (ctx) { const char* le = kzg_ctx_last_error(ctx); if (le && *le) m += std::string(" (") + le + ")"; }
            throw KzgError(KzgError::Kind::DeviceError, m);
        }
    }
}
}  // namespace detail

// ---- field and group elements in wire format -----------------------------------------------------------------------------------------
struct Fr {
    std::array<uint64_t, 4> limbs{};                                  // a * 2^256 mod r, little-endian u64 limbs (arkworks' Fr.0.0)
    bool operator==(const Fr& o) const { return limbs == o.limbs; }
    bool operator!=(const Fr& o) const { return !(*this == o); }
    Fr operator*(const Fr& o) const { return mont_mul(*this, o); }     // (host helper for tests: roots of unity, expected values)
    static Fr zero() { return Fr{}; }
    static Fr one() { return from_u64(1); }
    static Fr from_u64(uint64_t v) { Fr c; c.limbs = {v, 0, 0, 0}; return mont_mul(c, r2()); }
    // Fr::from_be_bytes_mod_order of exactly 32 bytes (helpers.rs:52, set_bytes_canonical_manual)
    static Fr from_be_bytes_mod_order(const uint8_t b[32]) {
        Fr c;
        for (int i = 0; i < 4; ++i) { uint64_t w = 0; for (int j = 0; j < 8; ++j) w = (w << 8) | b[(3 - i) * 8 + j]; c.limbs[i] = w; }
        return mont_mul(c, r2());                                     // any 256-bit value: the product is reduced below 2 r, then once more
    }
    // Fr::from_be_bytes_mod_order of any length (helpers::set_bytes_canonical, helpers.rs:32-34): Horner over the bytes
    static Fr from_be_bytes_mod_order(const uint8_t* b, size_t len) {
        const Fr k256 = from_u64(256);
        Fr acc = zero();
        for (size_t i = 0; i < len; ++i) acc = add(mont_mul(acc, k256), from_u64(b[i]));
        return acc;
    }
    static Fr add(const Fr& a, const Fr& b) {                         // both < r
        using u128 = unsigned __int128;
        uint64_t t[4]; u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a.limbs[j] + b.limbs[j]; t[j] = (uint64_t)c; c >>= 64; }
        bool ge = c != 0;
        if (!ge) { ge = true; for (int j = 3; j >= 0; --j) if (t[j] != N[j]) { ge = t[j] > N[j]; break; } }
        Fr out;
        if (ge) { u128 br = 0; for (int j = 0; j < 4; ++j) { const u128 d = (u128)t[j] - N[j] - (uint64_t)br; out.limbs[j] = (uint64_t)d; br = (d >> 64) & 1; } }
        else for (int j = 0; j < 4; ++j) out.limbs[j] = t[j];
        return out;
    }
    std::array<uint8_t, 32> to_be_bytes() const {                     // canonical big-endian (into_bigint().to_bytes_be())
        Fr u; u.limbs = {1, 0, 0, 0};
        const Fr c = mont_mul(*this, u);
        std::array<uint8_t, 32> out{};
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) out[(3 - i) * 8 + j] = (uint8_t)(c.limbs[i] >> (8 * (7 - j)));
        return out;
    }
private:
    static constexpr uint64_t N[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    static constexpr uint64_t INV = 0xc2e1f593efffffffull;            // -r^-1 mod 2^64
    static Fr r2() { Fr c; c.limbs = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull}; return c; }
    static Fr mont_mul(const Fr& a, const Fr& b) {                    // a b 2^-256 mod r, b < r
        using u128 = unsigned __int128;
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u128 c = 0;
            for (int j = 0; j < 4; ++j) { c += (u128)a.limbs[j] * b.limbs[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * INV;
            c = (u128)m * N[0] + t[0]; c >>= 64;
            for (int j = 1; j < 4; ++j) { c += (u128)m * N[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        bool ge = t[4] != 0;
        if (!ge) { ge = true; for (int j = 3; j >= 0; --j) if (t[j] != N[j]) { ge = t[j] > N[j]; break; } }
        Fr out;
        if (ge) { u128 br = 0; for (int j = 0; j < 4; ++j) { const u128 d = (u128)t[j] - N[j] - (uint64_t)br; out.limbs[j] = (uint64_t)d; br = (d >> 64) & 1; } }
        else for (int j = 0; j < 4; ++j) out.limbs[j] = t[j];
        return out;
    }
};

struct G1Affine {
    std::array<uint64_t, 8> xy{};                                     // x || y Montgomery; all zeros = the point at infinity
    bool operator==(const G1Affine& o) const { return xy == o.xy; }
    bool operator!=(const G1Affine& o) const { return !(*this == o); }
    bool is_zero() const { for (uint64_t w : xy) if (w) return false; return true; }
    static G1Affine identity() { return G1Affine{}; }
};
struct G2Affine {
    std::array<uint64_t, 16> w{};
    static G2Affine generator() { G2Affine g; kzg_g2_generator(g.w.data()); return g; }
    static G2Affine mul_generator(const Fr& s) { G2Affine g; detail::check(kzg_g2_mul_generator(s.limbs.data(), g.w.data())); return g; }   // [s]G2 (custom setups, tests)
    static G2Affine identity() { return G2Affine{}; }
    bool operator==(const G2Affine& o) const { return w == o.w; }
};
static_assert(sizeof(Fr) == 32 && sizeof(G1Affine) == 64 && sizeof(G2Affine) == 128, "wire formats are passed by pointer");

// ---- device context (no counterpart in the reference: it owns the GPU the calls run on) ---------------------------------------------
class Context {
public:
    explicit Context(int32_t device_id = 0) {
        kzg_ctx* c = nullptr;
        detail::check(kzg_ctx_create(device_id, &c));                 // KZG_ERR_NO_DEVICE without a GPU: there is no CPU fallback
        h_.reset(c, kzg_ctx_destroy);
    }
    kzg_ctx* handle() const { return h_.get(); }
    static Context& default_context() { static Context c(0); return c; }
private:
    std::shared_ptr<kzg_ctx> h_;
};

// ---- primitives/src/polynomial.rs ------------------------------------------------------------------------------------------------------
class PolynomialCoeffForm;
class PolynomialEvalForm {
public:
    // polynomial.rs:41-57: zero-padded to the next power of two; more than MAINNET_SRS_G1_SIZE elements is an error
    static PolynomialEvalForm new_(std::vector<Fr> evals) {
        if (evals.size() > MAINNET_SRS_G1_SIZE) throw KzgError::GenericError("Input size exceeds maximum polynomial size");
        PolynomialEvalForm p;
        p.len_underlying_blob_bytes_ = evals.size() * BYTES_PER_FIELD_ELEMENT;
        size_t n = 1; while (n < evals.size()) n <<= 1;               // usize::next_power_of_two (0 -> 1)
        evals.resize(n);
        p.evaluations_ = std::move(evals);
        return p;
    }
    const std::vector<Fr>& evaluations() const { return evaluations_; }
    size_t len() const { return evaluations_.size(); }
    size_t len_underlying_blob_bytes() const { return len_underlying_blob_bytes_; }
    size_t len_underlying_blob_field_elements() const { return len_underlying_blob_bytes_ / BYTES_PER_FIELD_ELEMENT; }
    std::optional<Fr> get_evalualtion(size_t i) const { return i < evaluations_.size() ? std::optional<Fr>(evaluations_[i]) : std::nullopt; }   // (sic, polynomial.rs:106)
    bool is_empty() const { return evaluations_.empty(); }
    inline PolynomialCoeffForm to_coeff_form(const Context& ctx = Context::default_context()) const;     // polynomial.rs:130-140 (IFFT)
private:
    friend class PolynomialCoeffForm;
    std::vector<Fr> evaluations_;
    size_t len_underlying_blob_bytes_ = 0;
};
class PolynomialCoeffForm {
public:
    static PolynomialCoeffForm new_(std::vector<Fr> coeffs) {         // polynomial.rs:169-185
        if (coeffs.size() > MAINNET_SRS_G1_SIZE) throw KzgError::GenericError("Input size exceeds maximum polynomial size");
        PolynomialCoeffForm p;
        p.len_underlying_blob_bytes_ = coeffs.size() * BYTES_PER_FIELD_ELEMENT;
        size_t n = 1; while (n < coeffs.size()) n <<= 1;
        coeffs.resize(n);
        p.coeffs_ = std::move(coeffs);
        return p;
    }
    const std::vector<Fr>& coeffs() const { return coeffs_; }
    size_t len() const { return coeffs_.size(); }
    size_t len_underlying_blob_bytes() const { return len_underlying_blob_bytes_; }
    size_t len_underlying_blob_field_elements() const { return len_underlying_blob_bytes_ / BYTES_PER_FIELD_ELEMENT; }
    std::optional<Fr> get_at_index(size_t i) const { return i < coeffs_.size() ? std::optional<Fr>(coeffs_[i]) : std::nullopt; }
    bool is_empty() const { return coeffs_.empty(); }
    PolynomialEvalForm to_eval_form(const Context& ctx = Context::default_context()) const {             // polynomial.rs:241-251 (FFT)
        PolynomialEvalForm p;
        p.evaluations_ = coeffs_;
        p.len_underlying_blob_bytes_ = len_underlying_blob_bytes_;
        const int32_t rc = kzg_fr_ntt(ctx.handle(), p.evaluations_.data()->limbs.data(), p.evaluations_.size(), 0);
        if (rc == KZG_ERR_DOMAIN || rc == KZG_ERR_NOT_POWER_OF_TWO) throw KzgError(KzgError::Kind::PolynomialError, "FFT error: Failed to construct domain for FFT");
        detail::check(rc, ctx.handle());
        return p;
    }
private:
    friend class PolynomialEvalForm;
    std::vector<Fr> coeffs_;
    size_t len_underlying_blob_bytes_ = 0;
};
inline PolynomialCoeffForm PolynomialEvalForm::to_coeff_form(const Context& ctx) const {
    PolynomialCoeffForm p;
    p.coeffs_ = evaluations_;
    p.len_underlying_blob_bytes_ = len_underlying_blob_bytes_;
    const int32_t rc = kzg_fr_ntt(ctx.handle(), p.coeffs_.data()->limbs.data(), p.coeffs_.size(), 1);
    if (rc == KZG_ERR_DOMAIN || rc == KZG_ERR_NOT_POWER_OF_TWO) throw KzgError(KzgError::Kind::PolynomialError, "FFT error: Failed to construct domain for IFFT");
    detail::check(rc, ctx.handle());
    return p;
}

// ---- primitives/src/blob.rs ------------------------------------------------------------------------------------------------------------
class Blob {
public:
    // Blob::new (blob.rs:30-35): every 32-byte chunk must be a canonical field element (helpers.rs:783-810)
    static Blob new_(const std::vector<uint8_t>& blob_data) {
        if (blob_data.size() % BYTES_PER_FIELD_ELEMENT != 0) throw KzgError::InvalidInputLength();
        static const uint8_t R_BE[32] = {0x30, 0x64, 0x4e, 0x72, 0xe1, 0x31, 0xa0, 0x29, 0xb8, 0x50, 0x45, 0xb6, 0x81, 0x81, 0x58, 0x5d,
                                         0x28, 0x33, 0xe8, 0x48, 0x79, 0xb9, 0x70, 0x91, 0x43, 0xe1, 0xf5, 0x93, 0xf0, 0x00, 0x00, 0x01};
        for (size_t i = 0; i < blob_data.size(); i += 32)
            if (std::memcmp(blob_data.data() + i, R_BE, 32) >= 0)
                throw KzgError(KzgError::Kind::InvalidFieldElement, "Field element at position " + std::to_string(i / 32) + " is not canonical or invalid");
        Blob b; b.blob_data_ = blob_data; return b;
    }
    // Blob::from_raw_data (blob.rs:41-44) = helpers::pad_payload (helpers.rs:823-840): 0x00 in front of every 31-byte chunk
    static Blob from_raw_data(const std::vector<uint8_t>& raw) {
        const size_t chunks = (raw.size() + 30) / 31;
        Blob b;
        b.blob_data_.assign(chunks * 32, 0);
        for (size_t e = 0; e < chunks; ++e) {
            const size_t s = e * 31, m = raw.size() - s < 31 ? raw.size() - s : 31;
            std::memcpy(b.blob_data_.data() + e * 32 + 1, raw.data() + s, m);
        }
        return b;
    }
    static Blob from(std::vector<uint8_t> padded) { Blob b; b.blob_data_ = std::move(padded); return b; }    // impl From<Vec<u8>> (blob.rs:90-97): unchecked
    std::vector<uint8_t> to_raw_data() const {                        // helpers::remove_internal_padding (helpers.rs:856-874)
        if (blob_data_.size() % BYTES_PER_FIELD_ELEMENT != 0) throw KzgError::InvalidInputLength();
        std::vector<uint8_t> out;
        out.reserve(blob_data_.size() / 32 * 31);
        for (size_t i = 0; i < blob_data_.size(); i += 32) out.insert(out.end(), blob_data_.begin() + i + 1, blob_data_.begin() + i + 32);
        return out;
    }
    const std::vector<uint8_t>& data() const { return blob_data_; }
    size_t len() const { return blob_data_.size(); }
    bool is_empty() const { return blob_data_.empty(); }
    // helpers::to_fr_array (helpers.rs:40-57) + PolynomialEvalForm::new: bytes -> Fr on the device (kzg_blob_to_fr)
    PolynomialEvalForm to_polynomial_eval_form(const Context& ctx = Context::default_context()) const { return PolynomialEvalForm::new_(to_fr_array(ctx)); }
    PolynomialCoeffForm to_polynomial_coeff_form(const Context& ctx = Context::default_context()) const { return PolynomialCoeffForm::new_(to_fr_array(ctx)); }
    bool operator==(const Blob& o) const { return blob_data_ == o.blob_data_; }
    std::vector<Fr> to_polynomial_eval_form_elements(const Context& ctx = Context::default_context()) const { return to_fr_array(ctx); }   // helpers::to_fr_array of the bytes (no padding to a power of two)
private:
    std::vector<Fr> to_fr_array(const Context& ctx) const {
        const size_t n = (blob_data_.size() + 31) / 32;               // get_num_element
        std::vector<Fr> out(n);
        if (n == 0) return out;
        size_t cap = 1; while (cap < n) cap <<= 1;
        std::vector<Fr> padded(cap);
        size_t n_out = 0;
        detail::check(kzg_blob_to_fr(ctx.handle(), blob_data_.data(), blob_data_.size(), padded.data()->limbs.data(), cap, &n_out), ctx.handle());
        std::copy(padded.begin(), padded.begin() + n, out.begin());
        return out;
    }
    std::vector<uint8_t> blob_data_;
};

// ---- prover/src/srs.rs -----------------------------------------------------------------------------------------------------------------
class SRS {
public:
    uint32_t order = 0;
    // SRS::new (srs.rs:35-49): `points_to_load` compressed points (32 B each, gnark big-endian flags) read from the file and
    // decompressed in one GPU kernel; they stay resident (the reference copies them on every commit, kzg.rs:119)
    static SRS new_(const std::string& path_to_g1_points, uint32_t order, uint32_t points_to_load, const Context& ctx = Context::default_context()) {
        if (points_to_load > order) throw KzgError::GenericError("Number of points to load exceeds SRS order.");          // srs.rs:36-40
        std::ifstream f(path_to_g1_points, std::ios::binary);
        if (!f) throw KzgError::GenericError("Error opening the file: " + path_to_g1_points);
        std::vector<uint8_t> bytes((size_t)points_to_load * 32);
        f.read(reinterpret_cast<char*>(bytes.data()), (std::streamsize)bytes.size());
        if ((size_t)f.gcount() != bytes.size()) throw KzgError::GenericError("Expected " + std::to_string(points_to_load) + " points, only read " + std::to_string((size_t)f.gcount() / 32));
        kzg_srs* h = nullptr;
        uint64_t bad = 0;
        detail::check(kzg_srs_load_compressed_be(ctx.handle(), bytes.data(), points_to_load, &h, &bad), ctx.handle());
        return SRS(h, order, ctx);
    }
    // srs.rs:205-251 (is_native = true: arkworks' compressed little-endian format; false: the gnark format of SRS::new): the decoded points
    static std::vector<G1Affine> parallel_read_g1_points_native(const std::string& file_path, uint32_t points_to_load, bool is_native,
                                                                const Context& ctx = Context::default_context()) {
        std::ifstream f(file_path, std::ios::binary);
        if (!f) throw KzgError::GenericError("Error opening the file: " + file_path);
        std::vector<uint8_t> bytes((size_t)points_to_load * 32);
        f.read(reinterpret_cast<char*>(bytes.data()), (std::streamsize)bytes.size());
        if ((size_t)f.gcount() != bytes.size()) throw KzgError::GenericError("Expected " + std::to_string(points_to_load) + " points, only read " + std::to_string((size_t)f.gcount() / 32));
        kzg_srs* h = nullptr;
        uint64_t bad = 0;
        const int32_t rc = is_native ? kzg_srs_load_compressed_ark_le(ctx.handle(), bytes.data(), points_to_load, &h, &bad)
                                     : kzg_srs_load_compressed_be(ctx.handle(), bytes.data(), points_to_load, &h, &bad);
        if (is_native && (rc == KZG_ERR_DESERIALIZE || rc == KZG_ERR_NOT_ON_CURVE)) throw KzgError::DeserializationError("Deserialization failed");   // traits.rs:34-36
        detail::check(rc, ctx.handle());
        return SRS(h, points_to_load, ctx).g1();
    }
    // already decoded points (the `g1: Cow<[G1Affine]>` field), uploaded once
    static SRS from_points(const std::vector<G1Affine>& g1, uint32_t order, const Context& ctx = Context::default_context()) {
        if (g1.size() > order) throw KzgError::GenericError("Number of points to load exceeds SRS order.");
        kzg_srs* h = nullptr;
        detail::check(kzg_srs_upload(ctx.handle(), g1.empty() ? nullptr : g1.data()->xy.data(), g1.size(), &h), ctx.handle());
        return SRS(h, order, ctx);
    }
    // tests / benches: P_i = tau^i G1 generated on the device (a setup whose secret is known)
    static SRS generate(const Fr& tau, size_t n, const Context& ctx = Context::default_context()) {
        kzg_srs* h = nullptr;
        detail::check(kzg_srs_generate(ctx.handle(), tau.limbs.data(), 0, n, &h), ctx.handle());
        return SRS(h, (uint32_t)n, ctx);
    }
    size_t len() const { return kzg_srs_len(h_.get()); }              // g1.len()
    std::vector<G1Affine> g1() const {                                // the points, read back from the device
        std::vector<G1Affine> out(len());
        if (!out.empty()) detail::check(kzg_srs_download(ctx_.handle(), h_.get(), 0, out.size(), out.data()->xy.data()), ctx_.handle());
        return out;
    }
    kzg_srs* handle() const { return h_.get(); }
    const Context& context() const { return ctx_; }
private:
    SRS(kzg_srs* h, uint32_t ord, const Context& ctx) : order(ord), h_(h, kzg_srs_free), ctx_(ctx) {}
    std::shared_ptr<kzg_srs> h_;
    Context ctx_;
};

// ---- prover/src/kzg.rs -----------------------------------------------------------------------------------------------------------------
class KZG {
public:
    static KZG new_() { return KZG(); }                               // kzg.rs:37-41
    // kzg.rs:65-72 (helpers::calculate_roots_of_unity, helpers.rs:553-589)
    void calculate_and_store_roots_of_unity(uint64_t length_of_data_after_padding, const Context& ctx = Context::default_context()) {
        const uint64_t elems = (length_of_data_after_padding + 31) / 32;
        const bool refused = length_of_data_after_padding == 0 || elems > ((uint64_t)1 << 28);      // the call below returns the reference's error
        size_t n = 1; while (!refused && n < elems) n <<= 1;
        std::vector<Fr> roots(n);
        size_t n_out = 0;
        detail::check(kzg_calculate_roots_of_unity(ctx.handle(), length_of_data_after_padding, roots.data()->limbs.data(), roots.size(), &n_out), ctx.handle());
        roots.resize(n_out);
        expanded_roots_of_unity_ = std::move(roots);
    }
    std::vector<Fr> get_roots_of_unities() const { return expanded_roots_of_unity_; }
    std::optional<Fr> get_nth_root_of_unity(size_t i) const { return i < expanded_roots_of_unity_.size() ? std::optional<Fr>(expanded_roots_of_unity_[i]) : std::nullopt; }

    // kzg.rs:84-104
    G1Affine commit_eval_form(const PolynomialEvalForm& polynomial, const SRS& srs) const {
        if (polynomial.len() > srs.len()) throw KzgError::SrsCapacityExceeded(polynomial.len(), srs.len());
        G1Affine out; uint8_t inf = 0;
        detail::check(kzg_commit_eval_form(srs.context().handle(), srs.handle(), polynomial.evaluations().data()->limbs.data(), polynomial.len(), out.xy.data(), &inf),
                      srs.context().handle(), polynomial.len(), srs.len());
        return out;
    }
    // kzg.rs:107-125
    G1Affine commit_coeff_form(const PolynomialCoeffForm& polynomial, const SRS& srs) const {
        if (polynomial.len() > srs.len()) throw KzgError::SerializationError("polynomial length is not correct");
        G1Affine out; uint8_t inf = 0;
        detail::check(kzg_commit_coeff_form(srs.context().handle(), srs.handle(), polynomial.coeffs().data()->limbs.data(), polynomial.len(), out.xy.data(), &inf),
                      srs.context().handle(), polynomial.len(), srs.len());
        return out;
    }
    // kzg.rs:182-185: bytes in, point out (bytes -> Fr, IFFT and MSM on the device)
    G1Affine commit_blob(const Blob& blob, const SRS& srs) const {
        size_t n = 1; while (n < (blob.len() + 31) / 32) n <<= 1;
        if (n > srs.len()) throw KzgError::SrsCapacityExceeded(n, srs.len());
        G1Affine out; uint8_t inf = 0;
        detail::check(kzg_commit_blob(srs.context().handle(), srs.handle(), blob.data().data(), blob.len(), out.xy.data(), &inf), srs.context().handle(), n, srs.len());
        return out;
    }
    // kzg.rs:215-234 (compute_proof_impl :128-178, on-domain branch :237-260)
    G1Affine compute_proof(const PolynomialEvalForm& polynomial, const Fr& z_fr, const SRS& srs) const {
        if (polynomial.len() != expanded_roots_of_unity_.size()) throw KzgError::GenericError("inconsistent length between blob and root of unities");
        if (polynomial.len() > srs.len()) throw KzgError::SrsCapacityExceeded(polynomial.len(), srs.len());
        G1Affine out; uint8_t inf = 0;
        detail::check(kzg_compute_proof(srs.context().handle(), srs.handle(), polynomial.evaluations().data()->limbs.data(), polynomial.len(),
                                        expanded_roots_of_unity_.data()->limbs.data(), expanded_roots_of_unity_.size(), z_fr.limbs.data(), out.xy.data(), &inf, nullptr),
                      srs.context().handle(), polynomial.len(), srs.len());
        return out;
    }
    // kzg.rs:187-207
    G1Affine compute_proof_with_known_z_fr_index(const PolynomialEvalForm& polynomial, uint64_t index, const SRS& srs) const {
        const std::optional<Fr> z = get_nth_root_of_unity((size_t)index);
        if (!z) throw KzgError::GenericError("Root of unity not found");
        return compute_proof(polynomial, *z, srs);
    }
    // kzg.rs:237-260: sum over the stored roots w^i != z of (f_i - value) w^i / ((z - w^i) z), on the GPU (kzg_compute_quotient_eval_on_domain)
    Fr compute_quotient_eval_on_domain(const Fr& z_fr, const std::vector<Fr>& eval_fr, const Fr& value_fr, const Context& ctx = Context::default_context()) const {
        const size_t n = expanded_roots_of_unity_.size();
        if (eval_fr.size() < n) throw std::out_of_range("index out of bounds: eval_fr is shorter than the roots of unity");      // eval_fr[i] panics in the reference
        Fr out;
        if (n == 0) return out;
        detail::check(kzg_compute_quotient_eval_on_domain(ctx.handle(), z_fr.limbs.data(), eval_fr.data()->limbs.data(), n, value_fr.limbs.data(), out.limbs.data()), ctx.handle());
        return out;
    }
    // kzg.rs:263-285
    std::vector<G1Affine> g1_ifft(size_t length, const SRS& srs) const {
        if (length == 0 || (length & (length - 1)) != 0) throw KzgError::FFTError("length provided is not a power of 2");
        std::vector<G1Affine> out(length);
        detail::check(kzg_g1_ifft(srs.context().handle(), srs.handle(), length, out.data()->xy.data()), srs.context().handle(), length, srs.len());
        return out;
    }
    // kzg.rs:288-309
    G1Affine compute_blob_proof(const Blob& blob, const G1Affine& commitment, const SRS& srs) const {
        G1Affine out; uint8_t inf = 0;
        detail::check(kzg_compute_blob_proof(srs.context().handle(), srs.handle(), blob.data().data(), blob.len(), expanded_roots_of_unity_.size(),
                                             commitment.xy.data(), out.xy.data(), &inf, nullptr, nullptr),
                      srs.context().handle(), (blob.len() + 31) / 32, srs.len());
        return out;
    }
private:
    std::vector<Fr> expanded_roots_of_unity_;
};

// ---- primitives/src/helpers.rs ------------------------------------------------------------------------------------------------------------
// Byte codecs are host loops (data formats); curve checks, the transcript hash and the pairing are the library's host code; conversions of whole
// blobs, roots of unity, evaluations, linear combinations and point decoding run on the GPU through the C-ABI.
namespace helpers {
inline size_t get_num_element(size_t data_len, size_t symbol_size) { return (data_len + symbol_size - 1) / symbol_size; }     // helpers.rs:36-38
inline Fr set_bytes_canonical(const std::vector<uint8_t>& data) { return Fr::from_be_bytes_mod_order(data.data(), data.size()); }   // helpers.rs:32-34
inline std::vector<uint8_t> pad_payload(const std::vector<uint8_t>& input_data) { return Blob::from_raw_data(input_data).data(); }   // helpers.rs:823-840
inline std::vector<uint8_t> remove_internal_padding(const std::vector<uint8_t>& padded_data) { return Blob::from(padded_data).to_raw_data(); }   // helpers.rs:856-874
inline bool is_zeroed(uint8_t first_byte, const std::vector<uint8_t>& buf) {                                                    // helpers.rs:121-132
    if (first_byte != 0) return false;
    for (uint8_t b : buf) if (b != 0) return false;
    return true;
}
inline std::array<uint8_t, 8> usize_to_be_bytes(size_t number) {                                                                // helpers.rs:769-780
    std::array<uint8_t, 8> out{};
    for (int i = 0; i < 8; ++i) out[i] = (uint8_t)((uint64_t)number >> (8 * (7 - i)));
    return out;
}
// helpers.rs:40-57 (the last chunk right-padded with zeros), on the device
inline std::vector<Fr> to_fr_array(const std::vector<uint8_t>& data, const Context& ctx = Context::default_context()) {
    return Blob::from(data).to_polynomial_eval_form_elements(ctx);
}
inline std::vector<Fr> blob_to_polynomial(const std::vector<uint8_t>& blob, const Context& ctx = Context::default_context()) { return to_fr_array(blob, ctx); }   // helpers.rs:28-30
// helpers.rs:80-119: canonical big-endian bytes of every element, cut at max_output_size
inline std::vector<uint8_t> to_byte_array(const std::vector<Fr>& data_fr, size_t max_output_size) {
    const size_t n = data_fr.size() * BYTES_PER_FIELD_ELEMENT < max_output_size ? data_fr.size() * BYTES_PER_FIELD_ELEMENT : max_output_size;
    std::vector<uint8_t> out(n);
    for (size_t i = 0; i * 32 < n; ++i) {
        const std::array<uint8_t, 32> b = data_fr[i].to_be_bytes();
        std::memcpy(out.data() + 32 * i, b.data(), n - 32 * i < 32 ? n - 32 * i : 32);
    }
    return out;
}
// helpers.rs:784-810
inline void validate_blob_data_as_canonical_field_elements(const std::vector<uint8_t>& data) { (void)Blob::new_(data); }
// helpers.rs:298-315
inline std::vector<Fr> compute_powers(const Fr& base, size_t count) {
    std::vector<Fr> out(count);
    Fr cur = Fr::one();
    for (size_t i = 0; i < count; ++i) { out[i] = cur; cur = cur * base; }
    return out;
}
// helpers.rs:382-390
inline Fr hash_to_field_element(const std::vector<uint8_t>& msg) {
    Fr out;
    detail::check(kzg_hash_to_field_element(msg.data(), msg.size(), out.limbs.data()));
    return out;
}
// helpers.rs:538-551: consts::PRIMITIVE_ROOTS_OF_UNITY[power] = w_28^(2^(28 - power)), w_28 = 5^((r - 1) / 2^28)
inline Fr get_primitive_root_of_unity(size_t power) {
    if (power > 28) throw KzgError::GenericError("power must be <= 28");
    Fr w; w.limbs = {0x636e735580d13d9cull, 0xa22bf3742445ffd6ull, 0x56452ac01eb203d8ull, 0x1860ef942963f9e7ull};
    for (size_t k = power; k < 28; ++k) w = w * w;
    return w;
}
// helpers.rs:553-589, generated on the device
inline std::vector<Fr> calculate_roots_of_unity(uint64_t length_of_data_after_padding, const Context& ctx = Context::default_context()) {
    size_t n_out = 0;
    const int32_t probe = kzg_calculate_roots_of_unity(ctx.handle(), length_of_data_after_padding, nullptr, 0, &n_out);             // the count, or the reference's error
    if (probe != KZG_ERR_INVALID_ARG) detail::check(probe, ctx.handle());
    std::vector<Fr> roots(n_out);
    detail::check(kzg_calculate_roots_of_unity(ctx.handle(), length_of_data_after_padding, roots.data()->limbs.data(), roots.size(), &n_out), ctx.handle());
    return roots;
}
// helpers.rs:328-337: the MSM of batch verification, on the device
inline G1Affine g1_lincomb(const std::vector<G1Affine>& points, const std::vector<Fr>& scalars, const Context& ctx = Context::default_context()) {
    G1Affine out; uint8_t inf = 0;
    const int32_t rc = kzg_msm_g1(ctx.handle(), points.empty() ? nullptr : points.data()->xy.data(), points.size(), scalars.empty() ? nullptr : scalars.data()->limbs.data(),
                                  scalars.size(), out.xy.data(), &inf);
    if (rc == KZG_ERR_MSM_LENGTH_MISMATCH) throw KzgError::MsmError(std::to_string(points.size() < scalars.size() ? points.size() : scalars.size()));
    detail::check(rc, ctx.handle());
    return out;
}
// helpers.rs:239-261, :694-708 (cofactor 1: on the curve = in the subgroup; the identity passes)
inline bool is_on_curve_g1(const G1Affine& g1) { return kzg_validate_g1_point(g1.xy.data()) == KZG_OK; }
inline void validate_g1_point(const G1Affine& point) { if (!is_on_curve_g1(point)) throw KzgError::NotOnCurveError("G1 point not on curve"); }
// helpers.rs:263-285
inline bool is_on_curve_g2(const G2Affine& g2) { int32_t ok = 0; detail::check(kzg_g2_is_on_curve(g2.w.data(), &ok)); return ok != 0; }
// helpers.rs:740-766
inline void example_validate_g2_point(const G2Affine& point) {
    int32_t reason = 0;
    detail::check(kzg_validate_g2_point(point.w.data(), &reason));
    switch (reason) {
        case 1: throw KzgError::NotOnCurveError("G2 point not on curve");
        case 2: throw KzgError::NotOnCurveError("G2 point is point at infinity");
        case 3: throw KzgError::NotOnCurveError("G2 point not in correct subgroup");
        case 4: throw KzgError::G2GeneratorNotAcceptedError("G2 point cannot be the generator point");
        default: return;
    }
}
// helpers.rs:392-398
inline bool pairings_verify(const G1Affine& a1, const G2Affine& a2, const G1Affine& b1, const G2Affine& b2) {
    int32_t ok = 0;
    detail::check(kzg_pairings_verify(a1.xy.data(), a2.w.data(), b1.xy.data(), b2.w.data(), &ok));
    return ok != 0;
}
// helpers.rs:151-173: y > (p - 1) / 2, y given in Montgomery form
inline bool lexicographically_largest(const std::array<uint64_t, 4>& y_mont) {
    using u128 = unsigned __int128;
    static constexpr uint64_t P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    static constexpr uint64_t HALF[4] = {0x9e10460b6c3e7ea3ull, 0xcbc0b548b438e546ull, 0xdc2822db40c0ac2eull, 0x183227397098d014ull};   // (p - 1) / 2
    static constexpr uint64_t PINV = 0x87d20782e4866389ull;           // -p^-1 mod 2^64
    uint64_t t[5] = {y_mont[0], y_mont[1], y_mont[2], y_mont[3], 0};
    for (int i = 0; i < 4; ++i) {                                     // arith::montgomery_reduce: y 2^-256 mod p
        const uint64_t m = t[0] * PINV;
        u128 c = ((u128)m * P[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = (uint64_t)(c >> 64);
    }
    bool ge = t[4] != 0;
    if (!ge) { ge = true; for (int j = 3; j >= 0; --j) if (t[j] != P[j]) { ge = t[j] > P[j]; break; } }
    if (ge) { u128 br = 0; for (int j = 0; j < 4; ++j) { const u128 d = (u128)t[j] - P[j] - (uint64_t)br; t[j] = (uint64_t)d; br = (d >> 64) & 1; } }
    for (int j = 3; j >= 0; --j) if (t[j] != HALF[j]) return t[j] > HALF[j];
    return false;
}
// helpers.rs:175-227: one gnark-compressed point through the kernel that decodes a whole SRS file, and a read-back
inline G1Affine read_g1_point_from_bytes_be(const std::vector<uint8_t>& g1_bytes_be, const Context& ctx = Context::default_context()) {
    if (g1_bytes_be.size() != 32) throw KzgError::DeserializationError("not enough bytes for g1 point");
    kzg_srs* h = nullptr;
    uint64_t bad = 0;
    const int32_t rc = kzg_srs_load_compressed_be(ctx.handle(), g1_bytes_be.data(), 1, &h, &bad);
    if (rc == KZG_ERR_DESERIALIZE) throw KzgError::DeserializationError("point at infinity not coded properly for g1");
    if (rc == KZG_ERR_NOT_ON_CURVE) throw KzgError::NotOnCurveError("compressed g1 point not on curve");
    detail::check(rc, ctx.handle());
    G1Affine out;
    const int32_t rc2 = kzg_srs_download(ctx.handle(), h, 0, 1, out.xy.data());
    kzg_srs_free(h);
    detail::check(rc2, ctx.handle());
    return out;
}
// helpers.rs:475-535
inline Fr evaluate_polynomial_in_evaluation_form(const PolynomialEvalForm& polynomial, const Fr& z, const Context& ctx = Context::default_context()) {
    Fr y;
    detail::check(kzg_evaluate_polynomial_in_evaluation_form(ctx.handle(), polynomial.evaluations().data()->limbs.data(), polynomial.len(), z.limbs.data(), y.limbs.data()), ctx.handle());
    return y;
}
// helpers.rs:411-472
inline Fr compute_challenge(const Blob& blob, const G1Affine& commitment) {
    Fr z;
    detail::check(kzg_compute_challenge(blob.data().data(), blob.len(), commitment.xy.data(), z.limbs.data()));
    return z;
}
// helpers.rs:613-665: the transcripts on a pool of host threads, the evaluations as one batched launch
inline std::pair<std::vector<Fr>, std::vector<Fr>> compute_challenges_and_evaluate_polynomial(const std::vector<Blob>& blobs, const std::vector<G1Affine>& commitments,
                                                                                              const Context& ctx = Context::default_context()) {
    if (blobs.size() != commitments.size() && !blobs.empty()) throw KzgError::GenericError("length's of the input are not the same or is empty");   // helpers.rs:618-622
    std::vector<Fr> zs(blobs.size()), ys(blobs.size());
    if (blobs.empty()) return {zs, ys};
    std::vector<const uint8_t*> ptrs(blobs.size());
    std::vector<size_t> lens(blobs.size());
    for (size_t i = 0; i < blobs.size(); ++i) { ptrs[i] = blobs[i].data().data(); lens[i] = blobs[i].len(); }
    detail::check(kzg_compute_challenges_and_evaluate_polynomial(ctx.handle(), ptrs.data(), lens.data(), commitments.data()->xy.data(), blobs.size(), zs.data()->limbs.data(),
                                                                 ys.data()->limbs.data()), ctx.handle());
    return {zs, ys};
}
}  // namespace helpers

// ---- verifier/src/verify.rs, verifier/src/batch.rs ------------------------------------------------------------------------------------
// g2_tau = nullptr: consts::G2_TAU (the mainnet setup, primitives/src/consts.rs:55-64); tests with their own tau pass [tau]G2.
namespace verify {
inline bool verify_proof(const G1Affine& commitment, const G1Affine& proof, const Fr& value_fr, const Fr& z_fr, const G2Affine* g2_tau = nullptr) {   // verify.rs:10-72
    int32_t ok = 0;
    detail::check(kzg_verify_proof(commitment.xy.data(), proof.xy.data(), value_fr.limbs.data(), z_fr.limbs.data(), g2_tau ? g2_tau->w.data() : nullptr, &ok));
    return ok != 0;
}
inline bool verify_blob_kzg_proof(const Blob& blob, const G1Affine& commitment, const G1Affine& proof, const G2Affine* g2_tau = nullptr,
                                  const Context& ctx = Context::default_context()) {                                                                 // verify.rs:76-98
    int32_t ok = 0;
    detail::check(kzg_verify_blob_kzg_proof(ctx.handle(), blob.data().data(), blob.len(), commitment.xy.data(), proof.xy.data(), g2_tau ? g2_tau->w.data() : nullptr, &ok), ctx.handle());
    return ok != 0;
}
}  // namespace verify
namespace batch {
inline bool verify_blob_kzg_proof_batch(const std::vector<Blob>& blobs, const std::vector<G1Affine>& commitments, const std::vector<G1Affine>& proofs,
                                        const G2Affine* g2_tau = nullptr, const Context& ctx = Context::default_context()) {                         // batch.rs:16-69
    if (!(commitments.size() == blobs.size() && proofs.size() == blobs.size())) throw KzgError::GenericError("length's of the input are not the same");
    std::vector<const uint8_t*> ptrs(blobs.size());
    std::vector<size_t> lens(blobs.size());
    for (size_t i = 0; i < blobs.size(); ++i) { ptrs[i] = blobs[i].data().data(); lens[i] = blobs[i].len(); }
    int32_t ok = 0;
    detail::check(kzg_verify_blob_kzg_proof_batch(ctx.handle(), ptrs.data(), lens.data(), commitments.empty() ? nullptr : commitments.data()->xy.data(),
                                                  proofs.empty() ? nullptr : proofs.data()->xy.data(), blobs.size(), g2_tau ? g2_tau->w.data() : nullptr, &ok), ctx.handle());
    return ok != 0;
}
}  // namespace batch

}  // namespace rust_kzg_bn254
