"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU restatement (plain Python integers + numpy marshalling) of the reference's per-row gate and
lookup checks, used exclusively by `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg as the *checker* of the HIP path.  Nothing in `zkevm_specs_amd/` (the
product) imports this package; the product fails loudly when its HIP library is missing.

Parity pin: every function here is validated against the *unmodified* reference source
(`/root/reference/src`, imported in the build container through the dependency shims in
`oracle/refshim/`) by `oracle/gen_golden.py`, which also writes the golden vectors under
`tests/golden/` that travel to the GPU box.
"""
