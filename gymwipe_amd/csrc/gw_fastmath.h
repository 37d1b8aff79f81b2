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
    const double r = fma(-q, slot, t);
    const double up = r + slot, down = r - slot;          // selects, not branches: cheaper than exec-mask regions on the GPU
    return (r < 0.0) ? up : ((r >= slot) ? down : r);
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

// Counter ticks in one jump.  The reference's tick times are a running f64 sum w <- fl(w + c)
// (counter_traffic.py:61).  While w stays inside one binade it moves on that binade's ulp grid, so every
// step adds the same grid multiple delta = fl(w + c) - w (exact by Sterbenz) -- unless c sits exactly half
// way between two grid points (then the first step from an odd mantissa differs: caught by comparing the
// first two increments).  Hence w_j = w + j*delta exactly (j*delta: <= 5 + 46 significant bits for w >= 2^-4;
// the sum is a grid point below the binade's end), and the number of ticks up to a time t is a floor
// division corrected by the exact FMA residual.  Returns false whenever any precondition is not met (early
// times, a binade boundary within reach, a rounding tie, an estimate off by more than one): the caller then
// runs the plain loop.  gw_fast_ticks_ok() checks the jump against that loop over every binade.
//   in : wake = time of the next tick, t >= 0, c = interval, inv_c = RN(1/c)
//   out: *n = ticks with w_j <= t (inclusive) or w_j < t (exclusive); *wake_out = time of the next tick after
//        them; *tie = one of the counted ticks is exactly t (inclusive only)
GW_HD bool gw_tick_jump(double wake, double t, double c, double inv_c, bool inclusive,
                        uint32_t* n, double* wake_out, bool* tie)
{
    // straight-line on purpose: on the GPU a chain of early returns becomes a nest of exec-mask regions that costs
    // more than the arithmetic it skips; every quantity below is harmless to compute when a precondition fails
    const bool any = inclusive ? (wake <= t) : (wake < t);
    const double w1 = wake + c, w2 = w1 + c;
    const double delta = w1 - wake;
    union { double f; uint64_t u; } hi;                   // 2^(exponent(wake) + 1): the end of wake's binade
    hi.f = wake;
    hi.u = (hi.u & 0x7ff0000000000000ull) + 0x0010000000000000ull;
    const double d = t - wake;                            // exact when wake <= t < 2*wake
    double n0 = floor(d * inv_c);
    double r = fma(-n0, delta, d);                        // exact
    const bool low = r < 0.0, high = r >= delta;          // the estimate is off by at most one
    n0 = low ? n0 - 1.0 : (high ? n0 + 1.0 : n0);
    r = low ? r + delta : (high ? r - delta : r);
    const bool hit = r == 0.0;                            // tick number n0 falls exactly on t
    const double cnt = (inclusive || !hit) ? n0 + 1.0 : n0;
    const double nw = fma(cnt, delta, wake);
    const bool ok = (wake >= 0.0625) && (wake < 2097152.0) && ((w2 - w1) == delta) && (d < 0.0625) &&
                    (r >= 0.0) && (r < delta) && (n0 >= 0.0) && (nw < hi.f);   // last: the next tick stays in the binade
    const bool use = any && ok;
    *n = use ? (uint32_t)cnt : 0u;
    *wake_out = use ? nw : wake;
    *tie = use && inclusive && hit;
    return ok || !any;
}

// ---- host-side validation (plain host functions) ----
inline bool gw_fast_ticks_ok(double c)
{
    if (!(c > 0) || !isfinite(c) || !(c >= 1e-5) || !(c <= 0.0625)) return false;
    const double inv_c = 1.0 / c;
    uint64_t x = 0x9e3779b97f4a7c15ull;
    long jumps = 0;
    for (int e = -4; e < 21; ++e) {                        // every binade the jump may be used in
        for (int i = 0; i < 4000; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            // a grid point of the binade: the true tick sequence visits only such points
            double wake = ldexp(1.0 + (double)(x >> 12) * ldexp(1.0, -52), e);
            if ((i & 7) == 0) wake = nextafter(ldexp(1.0, e + 1), 0.0) - (double)(x % 64) * c * 0.5;   // near the end
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            double t;
            switch (i & 3) {
            case 0:  t = wake + (double)(x % 25000) * 1e-6; break;
            case 1: { double w = wake; for (int j = (int)(x % 24); j > 0; --j) w = w + c; t = w; break; }    // exactly a tick
            case 2: { double w = wake; for (int j = (int)(x % 24); j > 0; --j) w = w + c; t = nextafter(w, (x & 64) ? 0.0 : 1e30); break; }
            default: t = wake - (double)(x % 3) * 1e-4; break;                                              // no tick at all
            }
            for (int incl = 0; incl < 2; ++incl) {
                uint32_t n = 0, nl = 0; double wo = 0; bool tie = false, tl = false;
                if (!gw_tick_jump(wake, t, c, inv_c, incl != 0, &n, &wo, &tie)) continue;
                ++jumps;
                double w = wake;                            // the reference's loop
                while (incl ? (w <= t) : (w < t)) { if (incl && w == t) tl = true; w = w + c; ++nl; }
                if (n != nl || !(wo == w) || tie != tl) return false;
            }
        }
    }
    if (!(jumps > 50000)) return false;                     // it must actually apply (not decline everywhere)
    // the true tick sequence from t = 0 (first 2^20 ticks): jump from every 13th tick to targets at, just below,
    // just above and between later ticks; expected counts come from the stored sequence itself
    const int M = 1 << 20;
    double* W = new double[M + 64];
    { double w = 0.0; for (int k = 0; k < M + 64; ++k) { W[k] = w; w = w + c; } }
    bool ok = true;
    long used = 0;
    for (int j = 0; j < M && ok; j += 13) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const int m = (int)(x % 24);
        const double targets[4] = {W[j + m], nextafter(W[j + m], 0.0), nextafter(W[j + m], 1e30),
                                   W[j + m] + (W[j + m + 1] - W[j + m]) * 0.37};
        for (int ti = 0; ti < 4 && ok; ++ti)
            for (int incl = 0; incl < 2 && ok; ++incl) {
                const double t = targets[ti];
                uint32_t n = 0; double wo = 0; bool tie = false;
                if (!gw_tick_jump(W[j], t, c, inv_c, incl != 0, &n, &wo, &tie)) continue;
                ++used;
                int last = j - 1;                              // index of the last tick counted
                while (last + 1 < M + 64 && (incl ? (W[last + 1] <= t) : (W[last + 1] < t))) ++last;
                const bool tl = incl && last >= j && W[last] == t;
                if ((int)n != last - j + 1 || !(wo == W[last + 1]) || tie != tl) ok = false;
            }
    }
    delete[] W;
    return ok && used > 100000;
}


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
