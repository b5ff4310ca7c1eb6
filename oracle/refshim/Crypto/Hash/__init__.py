from . import keccak  # noqa: F401
