"""BN254 G1 affine arithmetic (enough for `import` of the out-of-scope ECC circuit).

G2 / FQ2 / FQ12 are placeholders: the ECC and pairing circuits are out of scope
 and their tests are not part of the golden set.
"""
from . import FQ, curve_order, field_modulus  # noqa: F401

b = FQ(3)
b2 = None
b12 = None
G1 = (FQ(1), FQ(2))
Z1 = None


class FQ2(object):
    def __init__(self, coeffs):
        self.coeffs = tuple(coeffs)


class FQ12(object):
    def __init__(self, coeffs):
        self.coeffs = tuple(coeffs)


def is_inf(pt):
    return pt is None


def is_on_curve(pt, b_):
    if is_inf(pt):
        return True
    x, y = pt
    return y * y - x * x * x == b_


def double(pt):
    if is_inf(pt):
        return pt
    x, y = pt
    m = 3 * x * x / (2 * y)
    newx = m * m - 2 * x
    newy = -m * newx + m * x - y
    return (newx, newy)


def add(p1, p2):
    if p1 is None or p2 is None:
        return p1 if p2 is None else p2
    x1, y1 = p1
    x2, y2 = p2
    if x2 == x1 and y2 == y1:
        return double(p1)
    elif x2 == x1:
        return None
    m = (y2 - y1) / (x2 - x1)
    newx = m * m - x1 - x2
    newy = -m * newx + m * x1 - y1
    return (newx, newy)


def multiply(pt, n):
    if n == 0:
        return None
    elif n == 1:
        return pt
    elif not n % 2:
        return multiply(double(pt), n // 2)
    else:
        return add(multiply(double(pt), int(n // 2)), pt)


def eq(p1, p2):
    return p1 == p2


def neg(pt):
    if pt is None:
        return None
    x, y = pt
    return (x, -y)
