"""Stand-in for py-ecc 6.0.0 (only `bn128.FQ`, `bn128.curve_order`, `utils.prime_field_inv`)."""
from . import bn128, utils  # noqa: F401
