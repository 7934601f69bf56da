"""A SECOND restatement of the hot path's stream semantics, written in numpy / plain integers from the
reference's sources alone - not from oracle/mfm_oracle.c, and by different means (vectorised integer algebra where the oracle
loops in C, decimal arithmetic for the table, Python ints for the wrap-around).  Test infrastructure: tests compare it with the
oracle so that the oracle is no longer the only reading of the reference the parity tests rest on.  It stays "parity unpinned"
(no compiled reference object behind it) and says so.

What each function follows:
  taps ............ multifm/demod.c:204-269 (_demod_fir_prepare): t = (gain * cexp(j f_offs i)) * h[i], Q14, truncation
  rot_increment ... filter/direct_fir.c:72-79
  r14 ............. filter/complex.h:30-34 (round_q30_q15)
  fir ............. filter/direct_fir.c:328-417 + filter/complex.h:40-46 (cmul_q15_q30), int32 wrap
  derotate ........ filter/direct_fir.c:151-172, :406-413 + filter/complex.h:51-62 (cmul_q15_q15)
  fast_atan2f ..... multifm/fast_atan2f.c:101-174, table :14-81
  discriminate .... multifm/fm_demod.c:36-85
"""
import cmath
import math
from decimal import Decimal, getcontext

import numpy as np

F = np.float32


def _wrap32(a):
    """two's complement wrap of (arrays of) Python / int64 integers to int32"""
    return ((np.asarray(a, dtype=np.int64) + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)


def r14(a):
    """(int16)((a >> 14) + ((a >> 13) & 1)) on int32 input (arithmetic shifts), result wrapped to int16"""
    a = np.asarray(a, dtype=np.int64)
    v = (a >> 14) + ((a >> 13) & 1)
    return ((v + (1 << 15)) & 0xFFFF) - (1 << 15)


def taps(h, offset_hz, fs, gain=1.0):
    """demod.c:210,234,242-243: f_offs = -2.0*M_PI*off/fs (left to right); t = (gain * cexp(j f_offs i)) * h[i];
    (int16)(re * 16384.0), (int16)(im * 16384.0) - C's cast truncates toward zero"""
    f_offs = -2.0 * math.pi * float(offset_hz) / float(fs)
    cre, cim = [], []
    for i, hv in enumerate(h):
        e = cmath.exp(complex(0.0, f_offs * float(i)))
        g = complex(gain * e.real, gain * e.imag)      # real x complex: component-wise
        t = complex(g.real * float(hv), g.imag * float(hv))
        cre.append(int(t.real * 16384.0))              # int() truncates toward zero, like the C cast
        cim.append(int(t.imag * 16384.0))
    return np.array(cre, np.int64), np.array(cim, np.int64)


def rot_increment(offset_hz, fs, decimation):
    """direct_fir.c:73-78: fwt0 = 2.0*M_PI*off/fs; w = cexp(-j fwt0 D); (int16)(int32)(re * 16384), same for im"""
    fwt0 = 2.0 * math.pi * float(offset_hz) / float(fs)
    w = cmath.exp(complex(0.0, -fwt0 * float(decimation)))
    return int(w.real * 16384.0), int(w.imag * 16384.0)


_TABLE = None


def atan_table():
    """fast_atan2f.c:14-81: the literals are atan(i/255) at seven significant digits (entry 256 repeats 255), read as floats.
    Generated here with decimal arithmetic from math.atan's double (the oracle uses printf/strtof)."""
    global _TABLE
    if _TABLE is None:
        getcontext().prec = 40
        t = []
        for i in range(257):
            d = Decimal(math.atan(min(i, 255) / 255.0))
            if d == 0:
                t.append(F(0.0))
                continue
            exp10 = d.adjusted()
            q = Decimal(1).scaleb(exp10 - 6)          # seven significant digits
            t.append(F(float(d.quantize(q))))
        _TABLE = np.array(t, dtype=np.float32)
    return _TABLE


def fast_atan2f(y, x):
    """fast_atan2f.c:101-174 on float32 arrays, every operation rounded to float32 on its own (no contraction)"""
    y = np.asarray(y, F)
    x = np.asarray(x, F)
    T = atan_table()
    ya, xa = np.abs(y), np.abs(x)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = np.where(ya < xa, ya / xa, xa / ya).astype(F)              # :114-117 (equal magnitudes: xa / ya = 1)
    small = z.astype(np.float64) < 0.003921569                          # :121, the comparison is in double
    a = (z * F(255.0)).astype(F)                                        # :125
    with np.errstate(invalid="ignore"):
        k = np.where(np.isnan(a), 0, a).astype(np.int64) & 0xFF         # :126
    a2 = (a - k.astype(F)).astype(F)                                    # :127
    d = (T[k + 1] - T[k]).astype(F)
    interp = (T[k] + (d * a2).astype(F)).astype(F)                      # :130-131, unfused
    b = np.where(small, z, interp).astype(F)
    PI, HPI = F(math.pi), F(math.pi / 2)
    wide = xa > ya
    r_wide = np.where(x >= 0, np.where(y >= 0, b, -b), np.where(y >= 0, (PI - b).astype(F), (b - PI).astype(F)))
    r_tall = np.where(y >= 0, np.where(x >= 0, (HPI - b).astype(F), (HPI + b).astype(F)),
                      np.where(x >= 0, (-HPI + b).astype(F), (-HPI - b).astype(F)))
    out = np.where(wide, r_wide, r_tall).astype(F)
    return np.where((ya == 0) & (xa == 0), F(0.0), out).astype(F)      # :111-112


def discriminate(q_re, q_im):
    """fm_demod.c:55-72 over a whole stream: s = q conj(prev) in int32, phi = fast_atan2f((float)s_im, (float)s_re),
    pcm = (int16)(float)(((double)phi / M_PI) * 16384.0); prev starts at zero"""
    q_re = np.asarray(q_re, np.int64)
    q_im = np.asarray(q_im, np.int64)
    p_re = np.concatenate([[0], q_re[:-1]])
    p_im = np.concatenate([[0], q_im[:-1]])
    s_re = _wrap32(q_re * p_re + q_im * p_im)
    s_im = _wrap32(q_im * p_re - q_re * p_im)
    phi = fast_atan2f(s_im.astype(F), s_re.astype(F))
    v = ((phi.astype(np.float64) / math.pi) * 16384.0).astype(F)
    return np.trunc(v).astype(np.int64).astype(np.int16)


def channel(iq, cre, cim, decimation, incr):
    """the whole per-channel loop: returns (pcm int16[n], filtered IQ int16[n][2])"""
    x = np.asarray(iq, np.int64).reshape(-1, 2)
    T = len(cre)
    n_out = (x.shape[0] - T) // decimation + 1 if x.shape[0] >= T else 0
    if n_out <= 0:
        return np.zeros(0, np.int16), np.zeros((0, 2), np.int16)
    idx = (np.arange(n_out) * decimation)[:, None] + np.arange(T)[None, :]
    xr, xi = x[idx, 0], x[idx, 1]                                       # [n_out][T] windows, no zero history
    cr, ci = np.asarray(cre, np.int64), np.asarray(cim, np.int64)
    # exact sums in int64 (|sum| < 2^38), then the int32 wrap the reference's accumulator performs term by term
    ar = _wrap32((xr * cr).sum(1) - (xi * ci).sum(1))
    ai = _wrap32((xi * cr).sum(1) + (xr * ci).sum(1))
    fr, fi = r14(ar), r14(ai)
    ir, ii = int(incr[0]), int(incr[1])
    if ir == 0 and ii == 0:                                             # direct_fir.c:406: no derotation at all
        qr, qi = fr, fi
    else:
        rr, ri = 16384, 0
        qr = np.empty(n_out, np.int64)
        qi = np.empty(n_out, np.int64)
        for n in range(n_out):                                          # the recurrence is sequential (it rounds)
            f_r, f_i = int(fr[n]), int(fi[n])
            qr[n] = int(r14(_wrap32(f_r * rr - f_i * ri)))
            qi[n] = int(r14(_wrap32(f_r * ri + f_i * rr)))
            nr = int(r14(_wrap32(rr * ir - ri * ii)))
            ni = int(r14(_wrap32(rr * ii + ri * ir)))
            rr, ri = nr, ni
    pcm = discriminate(qr, qi)
    return pcm, np.stack([qr, qi], axis=1).astype(np.int16)
