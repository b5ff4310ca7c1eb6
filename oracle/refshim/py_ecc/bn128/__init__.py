"""py_ecc.bn128 stand-in: prime-field element class with py-ecc 6.0.0 semantics.

Semantics reproduced (py_ecc/fields/field_elements.py in the 6.0.0 wheel):
ctor from int/FQ reducing mod `field_modulus`; + - * / and reflected forms accept FQ or int
and return `type(self)`; `**` by square-and-multiply; `==`/`!=` vs FQ or int, TypeError
otherwise; unary minus; `.n`; `one()`/`zero()`; `__int__`; NO `__bool__`, NO ordering.
"""
from ..utils import prime_field_inv

field_modulus = 21888242871839275222246405745257275088696311157297823662689037894645226208583
curve_order = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class FQ(object):
    n = None
    field_modulus = field_modulus

    def __init__(self, val):
        if self.field_modulus is None:
            raise AttributeError("Field Modulus hasn't been specified")
        if isinstance(val, FQ):
            self.n = val.n
        elif isinstance(val, int):
            self.n = val % self.field_modulus
        else:
            raise TypeError(f"Expected an int or FQ object, but got object of type {type(val)}")

    @staticmethod
    def _on(other):
        if isinstance(other, FQ):
            return other.n
        if isinstance(other, int):
            return other
        raise TypeError(f"Expected an int or FQ object, but got object of type {type(other)}")

    def __add__(self, other):
        return type(self)((self.n + self._on(other)) % self.field_modulus)

    def __mul__(self, other):
        return type(self)((self.n * self._on(other)) % self.field_modulus)

    def __rmul__(self, other):
        return self * other

    def __radd__(self, other):
        return self + other

    def __rsub__(self, other):
        return type(self)((self._on(other) - self.n) % self.field_modulus)

    def __sub__(self, other):
        return type(self)((self.n - self._on(other)) % self.field_modulus)

    def __mod__(self, other):
        raise NotImplementedError("Modulo Operation not yet supported by fields")

    def __div__(self, other):
        on = self._on(other)
        return type(self)(self.n * prime_field_inv(on, self.field_modulus) % self.field_modulus)

    def __truediv__(self, other):
        return self.__div__(other)

    def __rdiv__(self, other):
        on = self._on(other)
        return type(self)(prime_field_inv(self.n, self.field_modulus) * on % self.field_modulus)

    def __rtruediv__(self, other):
        return self.__rdiv__(other)

    def __pow__(self, other):
        if other == 0:
            return type(self)(1)
        elif other == 1:
            return type(self)(self.n)
        elif other % 2 == 0:
            return (self * self) ** (other // 2)
        else:
            return ((self * self) ** int(other // 2)) * self

    def __eq__(self, other):
        if isinstance(other, FQ):
            return self.n == other.n
        elif isinstance(other, int):
            return self.n == other
        else:
            raise TypeError(f"Expected an int or FQ object, but got object of type {type(other)}")

    def __ne__(self, other):
        return not self == other

    def __neg__(self):
        return type(self)(-self.n)

    def __repr__(self):
        return repr(self.n)

    def __int__(self):
        return self.n

    @classmethod
    def one(cls):
        return cls(1)

    @classmethod
    def zero(cls):
        return cls(0)


from . import bn128_curve, bn128_pairing  # noqa: E402,F401
