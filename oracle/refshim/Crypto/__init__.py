"""Stand-in for pycryptodome 3.14.1 (keccak-256 + the RNG helpers tests/common.py uses)."""
