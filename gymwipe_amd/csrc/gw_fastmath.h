// gw_fastmath.h -- exact replacements for three expensive f64 operations of the step, shared by the
// kernels and by the host code that VALIDATES them at gw_create.  "Exact" = bit-identical to the plain
// IEEE operation the reference performs; whenever the validation cannot establish that, the flag in
// GwDevConst stays 0 and the kernels use the plain form.
#pragma once
#include <math.h>
#include <stdint.h>
#include "gw_queue.h"   // GW_HD

// t % slot for t >= 0 (simtools.py:53).  q = floor(RN(t * RN(1/slot))) is within +-1 of floor(t/slot)
// while t/slot < 2^40; the FMA residual t - q*slot is then exactly representable provided the
// mantissa of slot leaves 2^-12 relative headroom below 2^53 (checked by gw_fast_fmod_ok), and one
// conditional +-slot (exact by Sterbenz) lands on the true remainder.
GW_HD double gw_fast_fmod(double t, double slot, double inv_slot)
{
    const double q = floor(t * inv_slot);
    double r = fma(-q, slot, t);
    if (r < 0.0) r += slot;
    else if (r >= slot) r -= slot;
    return r;
}

// a / b for the integer-valued numerators the step produces (bit counts), b = data rate:
// Markstein's q1 = fma(fma(-q0, b, a), rcp, q0) with q0 = a*rcp.  Validated exhaustively for every
// numerator 8*k, k <= max packet bytes, by gw_fast_div_ok.
GW_HD double gw_fast_div(double a, double b, double rcp)
{
    const double q0 = a * rcp;
    const double r = fma(-q0, b, a);
    return fma(r, rcp, q0);
}

// ---- host-side validation (plain host functions) ----
inline bool gw_fast_fmod_ok(double slot, double* limit_out)
{
    int ex = 0;
    const double m = frexp(slot, &ex);                     // slot = m * 2^ex, m in [0.5, 1)
    if (!(slot > 0) || !isfinite(slot)) return false;
    if (!(m * (1.0 + 1.0 / 4096.0) < 1.0)) return false;   // mantissa headroom (see above)
    *limit_out = ldexp(slot, 40);
    // belt and braces: compare with the library on a structured + pseudo-random sample
    const double inv = 1.0 / slot;
    uint64_t x = 88172645463325252ull;
    for (int i = 0; i < 200000; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        double t;
        switch (i & 3) {
        case 0:  t = (double)(x >> 11) * ldexp(1.0, -53) * (*limit_out) * 1e-3; break;  // uniform
        case 1:  t = (double)(x % 100000000ull) * slot; break;                          // near multiples
        case 2:  t = nextafter((double)(x % 100000000ull) * slot, (x & 1) ? 0.0 : 1e30); break;
        default: t = (double)(x % 1000000ull) * 1e-3 + (double)((x >> 32) % 1000) * 3.4e-10; break;
        }
        if (!(t < *limit_out)) continue;
        const double a = gw_fast_fmod(t, slot, inv), b = fmod(t, slot);
        if (!(a == b)) return false;
    }
    return true;
}

inline bool gw_fast_div_ok(double b, int64_t max_bytes)
{
    if (!(b > 0) || !isfinite(b)) return false;
    const double rcp = 1.0 / b;
    for (int64_t k = 0; k <= max_bytes; ++k) {
        volatile double a = (double)(k * 8);
        volatile double plain = a / b;
        if (!(gw_fast_div(a, b, rcp) == plain)) return false;
    }
    return true;
}
