"""py_ecc.utils.prime_field_inv — published algorithm: extended Euclid, inv(0) == 0."""


def prime_field_inv(a: int, n: int) -> int:
    if a == 0:
        return 0
    lm, hm = 1, 0
    low, high = a % n, n
    while low > 1:
        r = high // low
        nm, new = hm - lm * r, high - low * r
        lm, low, hm, high = nm, new, lm, low
    return lm % n
