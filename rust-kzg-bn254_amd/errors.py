"""Error types mirroring the reference's `KzgError` / `PolynomialError` enums
(primitives/src/errors.rs:13-24, :32-86); `str(e)` reproduces the reference's `Display` text."""


class KzgError(Exception):
    """Base of every error the reference returns as `Err(KzgError::..)`."""

    variant = "KzgError"
    prefix = ""

    def __init__(self, message=""):
        self.message = message
        super().__init__(f"{self.prefix}{message}" if self.prefix else message)


class GenericError(KzgError):
    variant = "GenericError"; prefix = "generic error: "


class MsmError(KzgError):
    variant = "MsmError"; prefix = "MSM error: "


class CommitError(KzgError):
    variant = "CommitError"; prefix = "commit error: "


class FFTError(KzgError):
    variant = "FFTError"; prefix = "FFT error: "


class SerializationError(KzgError):
    variant = "SerializationError"; prefix = "serialization error: "


class DeserializationError(KzgError):
    variant = "DeserializationError"; prefix = "deserialization error: "


class G2GeneratorNotAcceptedError(KzgError):
    variant = "G2GeneratorNotAcceptedError"; prefix = "g2 generator not accepted error: "


class InvalidDenominator(KzgError):
    variant = "InvalidDenominator"

    def __init__(self, message=""):
        super().__init__("invalid denominator")


class NotOnCurveError(KzgError):
    variant = "NotOnCurveError"; prefix = "not on curve error: "


class InvalidInputLength(KzgError):
    variant = "InvalidInputLength"

    def __init__(self, message=""):
        super().__init__("input length must be a multiple of 32")


class InvalidFieldElement(KzgError):
    variant = "InvalidFieldElement"; prefix = "invalid field element: "


class SrsCapacityExceeded(KzgError):
    variant = "SrsCapacityExceeded"

    def __init__(self, polynomial_len, srs_len):
        self.polynomial_len, self.srs_len = polynomial_len, srs_len
        super().__init__(f"polynomial degree {polynomial_len} exceeds SRS capacity {srs_len}")


class PolynomialFFTError(KzgError):
    """`PolynomialError::FFTError` (polynomial.rs:132-134, :243-245)."""
    variant = "PolynomialError::FFTError"; prefix = "FFT error: "


class DeviceError(RuntimeError):
    """HIP runtime failure or missing GPU / library.  The product has no CPU fallback."""
