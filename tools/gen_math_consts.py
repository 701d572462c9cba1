#!/usr/bin/env python3
"""Generates the constants of hope_amd/csrc/hope_math.h with exact rational / 80-digit decimal arithmetic."""
from decimal import Decimal, getcontext
from fractions import Fraction
import math, struct

getcontext().prec = 90


def pi_dec():
    # Machin: pi = 16 atan(1/5) - 4 atan(1/239)
    def atan_inv(n):
        x = Decimal(1) / n
        x2 = x * x
        t, s, k = x, x, 1
        while abs(t) > Decimal(10) ** -88:
            t = -t * x2
            k += 2
            s += t / k
        return s
    return 16 * atan_inv(5) - 4 * atan_inv(239)


def to_double(d):
    """correctly rounded double of a Decimal"""
    return float(Fraction(d))      # Fraction -> float is correctly rounded


def trunc_bits(x, bits):
    """double x truncated to its leading `bits` mantissa bits"""
    m, e = math.frexp(x)
    q = math.floor(m * 2 ** bits)
    return math.ldexp(q, e - bits)


def split3(d, bits=33):
    p1 = trunc_bits(to_double(d), bits)
    r = d - Decimal(p1)
    p2 = trunc_bits(to_double(r), bits)
    r2 = r - Decimal(p2)
    p3 = to_double(r2)
    return p1, p2, p3


def hilo(d):
    hi = to_double(d)
    lo = to_double(d - Decimal(hi))
    return hi, lo


def atan_dec(xd):
    # atan(x) for |x| <= 1 by argument halving: atan(x) = 2 atan(x / (1 + sqrt(1 + x^2)))
    x = Decimal(xd)
    n = 0
    while abs(x) > Decimal('0.01'):
        x = x / (1 + (1 + x * x).sqrt())
        n += 1
    x2 = x * x
    t, s, k = x, x, 1
    while abs(t) > Decimal(10) ** -88:
        t = -t * x2
        k += 2
        s += t / k
    return s * (2 ** n)


def ln2_dec():
    # ln 2 = 2 atanh(1/3)
    x = Decimal(1) / 3
    x2 = x * x
    t, s, k = x, x, 1
    while t > Decimal(10) ** -88:
        t = t * x2
        k += 2
        s += t / k
    return 2 * s


def c(v):
    return repr(v) if isinstance(v, float) else str(v)


pi = pi_dec()
print('// pi check', str(pi)[:40])
p1, p2, p3 = split3(pi / 2)
print(f'HM_PIO2_1 = {p1!r}, HM_PIO2_2 = {p2!r}, HM_PIO2_3 = {p3!r};')
print(f'HM_2_PI = {to_double(Decimal(2) / pi)!r};')
print('HM_PI hi/lo', hilo(pi), ' PIO2 hi/lo', hilo(pi / 2), ' PIO4 hi/lo', hilo(pi / 4))
print('ATAN_HALF hi/lo', hilo(atan_dec('0.5')))
l1, l2, l3 = split3(ln2_dec(), 32)
print(f'LN2 parts {l1!r} {l2!r} {l3!r}; INV_LN2 = {to_double(1 / ln2_dec())!r}')
S = [float(Fraction((-1) ** n, math.factorial(2 * n + 1))) for n in range(1, 11)]
Cc = [float(Fraction((-1) ** n, math.factorial(2 * n))) for n in range(2, 12)]
print('SIN', ', '.join(repr(v) for v in S))
print('COS', ', '.join(repr(v) for v in Cc))
A = [float(Fraction((-1) ** n, 2 * n + 1)) for n in range(1, 16)]
print('ATAN', ', '.join(repr(v) for v in A))
E = [float(Fraction(1, math.factorial(n))) for n in range(2, 15)]
print('EXP', ', '.join(repr(v) for v in E))
TH = [Fraction(-1, 3), Fraction(2, 15), Fraction(-17, 315), Fraction(62, 2835), Fraction(-1382, 155925),
      Fraction(21844, 6081075), Fraction(-929569, 638512875), Fraction(6404582, 10854718875),
      Fraction(-443861162, 1856156927625)]
print('TANH', ', '.join(repr(float(v)) for v in TH))
