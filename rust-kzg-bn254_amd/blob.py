"""Mirror of primitives/src/blob.rs: `Blob` wraps padded bytes (a multiple of 32, every chunk a
canonical field element)."""
from . import helpers
from .consts import BYTES_PER_FIELD_ELEMENT, FR_MODULUS
from .errors import InvalidFieldElement, InvalidInputLength
from .polynomial import PolynomialCoeffForm, PolynomialEvalForm


class Blob:
    def __init__(self, blob_data: bytes):
        """Blob::new (blob.rs:30-35): validates canonical elements (helpers.rs:783-810)."""
        helpers.validate_blob_data_as_canonical_field_elements(blob_data)
        self.blob_data = bytes(blob_data)

    @classmethod
    def from_raw_data(cls, raw_data: bytes) -> "Blob":
        b = cls.__new__(cls)
        b.blob_data = helpers.pad_payload(raw_data)
        return b

    @classmethod
    def from_padded_unchecked(cls, blob_data: bytes) -> "Blob":
        """`impl From<Vec<u8>> for Blob` (blob.rs:90-97): no validation."""
        b = cls.__new__(cls)
        b.blob_data = bytes(blob_data)
        return b

    def to_raw_data(self) -> bytes:
        return helpers.remove_internal_padding(self.blob_data)

    def data(self) -> bytes:
        return self.blob_data

    def __len__(self):
        return len(self.blob_data)

    def is_empty(self):
        return len(self.blob_data) == 0

    def to_polynomial_eval_form(self) -> PolynomialEvalForm:
        return PolynomialEvalForm(helpers.to_fr_array(self.blob_data))

    def to_polynomial_coeff_form(self) -> PolynomialCoeffForm:
        return PolynomialCoeffForm(helpers.to_fr_array(self.blob_data))

    def __eq__(self, other):
        return isinstance(other, Blob) and self.blob_data == other.blob_data
