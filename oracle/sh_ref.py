"""CPU oracle: real spherical-harmonics basis encoder (degree 1..8) and its input gradient.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Restates /root/reference/shencoder/src/shencoder.cu:
  kernel_sh           :28-355   outputs[b, 0..C*C) and dy_dx[b, d, 0..C*C)
  kernel_sh_backward  :359-382  grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]
The reference hard-codes one Cartesian polynomial per basis function (:50-120)
and its three partial derivatives (:130-350).  Here the same polynomials are
*derived* instead of transcribed:

    Y_l^0    =            K_l^0 * P_l(z)
    Y_l^m    = (-1)^m  sqrt2 K_l^m * (d^m/dz^m P_l)(z) * Re (x+iy)^m      (m > 0)
    Y_l^-m   = (-1)^m  sqrt2 K_l^m * (d^m/dz^m P_l)(z) * Im (x+iy)^m      (m > 0)
    K_l^m    = sqrt((2l+1)/(4 pi) * (l-m)!/(l+m)!),   output index = l*l + l + m

and dy_dx are the plain partial derivatives of those polynomials in (x, y, z)
(no unit-norm constraint, exactly as :130-350 treats them).  Pinned by
tests/golden/g6_sh_encoder.npz, which evaluates the reference's own table.
"""
from __future__ import annotations

import math
from functools import lru_cache

import numpy as np


def _legendre(l):
    """Coefficients (ascending powers of z) of P_l."""
    p0, p1 = np.array([1.0]), np.array([0.0, 1.0])
    if l == 0:
        return p0
    for n in range(1, l):
        a = np.zeros(n + 2)
        a[1:] += (2 * n + 1) * p1
        a[:len(p0)] -= n * p0
        p0, p1 = p1, a / (n + 1)
    return p1


def _deriv(c, m):
    for _ in range(m):
        c = np.array([i * c[i] for i in range(1, len(c))]) if len(c) > 1 else np.array([0.0])
    return c


def _xy_power(m):
    """Re and Im of (x+iy)^m as {(i,j): coef} over x^i y^j."""
    re, im = {}, {}
    for k in range(m + 1):
        c = math.comb(m, k)
        tgt = re if k % 2 == 0 else im
        sgn = (-1) ** (k // 2)
        tgt[(m - k, k)] = tgt.get((m - k, k), 0) + sgn * c
    return re, im


@lru_cache(maxsize=None)
def basis_polynomials(degree):
    """List of degree^2 polynomials, each a dict {(i,j,k): coef} over x^i y^j z^k."""
    polys = []
    for l in range(degree):
        P = _legendre(l)
        row = {}
        for m in range(0, l + 1):
            K = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m))
            Q = _deriv(P, m)
            if m == 0:
                row[0] = {(0, 0, k): K * Q[k] for k in range(len(Q)) if Q[k] != 0}
                continue
            re, im = _xy_power(m)
            f = (-1) ** m * math.sqrt(2.0) * K
            for sign, part in ((+1, re), (-1, im)):
                poly = {}
                for (i, j), c in part.items():
                    for k in range(len(Q)):
                        if Q[k] != 0:
                            poly[(i, j, k)] = poly.get((i, j, k), 0.0) + f * c * Q[k]
                row[sign * m] = poly
        for m in range(-l, l + 1):
            polys.append(row[m])
    return polys


def _eval(poly, x, y, z):
    out = np.zeros_like(x, dtype=np.float64)
    for (i, j, k), c in poly.items():
        out = out + c * x ** i * y ** j * z ** k
    return out


def _partial(poly, axis):
    out = {}
    for e, c in poly.items():
        if e[axis] == 0:
            continue
        ne = list(e)
        ne[axis] -= 1
        out[tuple(ne)] = out.get(tuple(ne), 0.0) + c * e[axis]
    return out


def sh_encode_forward(inputs: np.ndarray, degree: int, calc_grad_inputs: bool = False):
    """inputs [B,3] -> outputs [B, degree^2] (fp32) and dy_dx [B, 3*degree^2] or None."""
    assert inputs.shape[1] == 3 and 1 <= degree <= 8
    x, y, z = (inputs[:, i].astype(np.float64) for i in range(3))
    polys = basis_polynomials(degree)
    out = np.stack([_eval(p, x, y, z) for p in polys], axis=1).astype(np.float32)
    dy_dx = None
    if calc_grad_inputs:
        parts = [np.stack([_eval(_partial(p, ax), x, y, z) for p in polys], axis=1) for ax in range(3)]
        dy_dx = np.concatenate(parts, axis=1).astype(np.float32)      # [B, d*C2 + ch]
    return out, dy_dx


def sh_encode_backward(grad: np.ndarray, dy_dx: np.ndarray, degree: int):
    """grad [B, C2], dy_dx [B, 3*C2] -> grad_inputs [B,3]."""
    B, C2 = grad.shape
    d = dy_dx.reshape(B, 3, C2).astype(np.float64)
    return (d * grad[:, None, :].astype(np.float64)).sum(-1).astype(np.float32)
