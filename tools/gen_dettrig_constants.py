"""Print the constants of include/mas_dettrig.h: correctly rounded 1/n! Taylor coefficients and
the 33-bit pieces of pi/2 used by the three-stage Cody-Waite reduction (exact arithmetic with
fractions / decimal; no libm involved)."""
from decimal import Decimal, getcontext
from fractions import Fraction
import math
import struct

getcontext().prec = 120
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899862803482534211706798214808651328230664709384460955058223172535940812848111745")
pio2 = PI / 2


def trunc33(d: Decimal) -> float:
    f = float(d)
    bits = struct.unpack("<Q", struct.pack("<d", f))[0]
    bits &= ~((1 << 20) - 1)          # keep the top 33 bits of the 53-bit significand
    g = struct.unpack("<d", struct.pack("<Q", bits))[0]
    return g


rest = pio2
out = {}
for i in (1, 2, 3):
    p = trunc33(rest)
    rest = rest - Decimal(p)
    out[f"pio2_{i}"] = p
    out[f"pio2_{i}t"] = float(rest)
out["invpio2"] = float(Decimal(2) / PI)
for k, v in out.items():
    print(f"{k} = {v!r}  /* {v.hex()} */")
print("sin coefficients (-1)^k / (2k+1)!  k=1..8")
for k in range(1, 9):
    c = Fraction((-1) ** k, math.factorial(2 * k + 1))
    print(f"  S{2*k+1} = {float(c)!r}  /* {float(c).hex()} */")
print("cos coefficients (-1)^k / (2k)!  k=2..8")
for k in range(2, 9):
    c = Fraction((-1) ** k, math.factorial(2 * k))
    print(f"  C{2*k} = {float(c)!r}  /* {float(c).hex()} */")
