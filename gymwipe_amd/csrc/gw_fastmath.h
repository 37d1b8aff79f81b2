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

// The same with a ONE-SIDED quotient estimate: inv_lo = RN(1/slot) * (1 - 2^-44) makes q = floor(RN(t * inv_lo)) either
// floor(t/slot) or one less while t/slot < 2^40 (the scaling outweighs the two roundings, 2^-52 each, and costs at most
// 2^-4 of a quotient), so the residual lies in [0, 2 slot) and ONE conditional subtraction remains: 7 instructions instead
// of 11 on the default step kernel's per-packet path.  Validated next to gw_fast_fmod by gw_fast_fmod_ok.
GW_HD double gw_fast_fmod_lo(double t, double slot, double inv_lo)
{
    const double q = floor(t * inv_lo);
    const double r = fma(-q, slot, t);
    return (r >= slot) ? r - slot : r;
}
GW_HD double gw_inv_lo(double x) { return (1.0 / x) * (1.0 - 1.0 / 17592186044416.0); }   // RN(1/x) * (1 - 2^-44)

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

// The same jump with its time-independent preconditions established ONCE per env.step() (the default step kernel counts
// ticks after every data packet: one precondition block per packet was a fifth of the window loop's instructions).
//
// gw_tick_span_ok(wake, t_last, c): every tick counted during the step lies in [wake, t_last], every base the jump starts
// from is `wake` or a later point of the running sum, and the next tick after the last counted one is at most one interval
// past t_last.  It checks, for the WHOLE span, what gw_tick_jump checks per call:
//   * wake >= 2^-4 and the span ends (two intervals of slack) below the end of wake's binade and below 2^21: all those
//     sums move on one ulp grid, so every step adds the same grid multiple `delta`;
//   * the first two increments agree (no round-to-even alternation from an odd mantissa; from the second sum on the
//     mantissas are even and the increments agree for good -- so the test at the span's first base covers the later ones);
//   * t_last - wake < 2^-4: every t - base is exact (base <= t < 2 base) and the floor estimate is off by at most one:
//     |delta - c| <= ulp/2 <= 2^-33, c >= 1e-5 (gw_fast_ticks_ok), at most 6 250 ticks: error < 0.1.
// gw_tick_jump_pre then needs no test but `any`; `inrange` reports the one data-dependent condition (the corrected
// remainder lies in [0, delta)), which the bound above says always holds: callers fall back to the plain loop on it anyway,
// and gw_fast_ticks_ok checks over every binade that it never fires.
GW_HD bool gw_tick_span_ok(double wake, double t_last, double c, double* delta_out)
{
    const double w1 = wake + c, w2 = w1 + c;
    const double delta = w1 - wake;
    union { double f; uint64_t u; } hi;                   // 2^(exponent(wake) + 1): the end of wake's binade
    hi.f = wake;
    hi.u = (hi.u & 0x7ff0000000000000ull) + 0x0010000000000000ull;
    *delta_out = delta;
    return (wake >= 0.0625) && (wake < 2097152.0) && ((w2 - w1) == delta) && ((t_last - wake) < 0.0625) &&
           (t_last >= wake - c) && ((t_last + (c + c)) < hi.f);
}

GW_HD void gw_tick_jump_pre(double wake, double t, double delta, double inv_c, bool inclusive,
                            uint32_t* n, double* wake_out, bool* tie, bool* inrange)
{
    const bool any = inclusive ? (wake <= t) : (wake < t);
    const double d = t - wake;                            // exact (span precondition)
    const double n0 = floor(d * inv_c);
    const double r = fma(-n0, delta, d);                  // exact; the estimate is off by at most one:
    const bool low = r < 0.0, high = r >= delta;          //   r in [-delta, 2 delta)
    const bool hit = (r == 0.0) || (r == delta) || (r == -delta);   // some tick falls exactly on t
    const double n1 = low ? n0 - 1.0 : (high ? n0 + 1.0 : n0);     // index of the last tick <= t
    // (no tick at all: count 0, and fma(0, delta, wake) is wake itself -- one select instead of a guarded region)
    const double cnt = any ? ((inclusive || !hit) ? n1 + 1.0 : n1) : 0.0;
    *inrange = !any || ((r >= -delta) && (r < delta + delta));
    *n = (uint32_t)cnt;
    *wake_out = fma(cnt, delta, wake);
    *tie = any && inclusive && hit;
}

// gw_tick_jump_pre with a one-sided estimate and integer corrections -- the form the default step kernel runs once per data
// packet.  inv_c_lo = RN(1/c) * (1 - 2^-31 / c) <= 1/delta for every increment delta the span can have (|delta - c| <= half an
// ulp <= 2^-33 below 2^21 s), so n0 = floor(d * inv_c_lo) is the index of the last tick <= t or one less (the deficit
// n * 2^-30 / c stays below one for n <= 6 250 ticks and c >= 1e-5): the residual lies in [0, 2 delta), one comparison
// corrects it.  No `any` test either: for base - delta < t < base the same arithmetic gives n0 = -1, residual in (0, delta),
// count 0.  `sane` is the self-check of those bounds (residual not negative, next tick beyond t); the span preconditions
// make it always true (gw_fast_ticks_ok counts on it never firing), and a caller that does not want a fallback path raises
// GW_FLAG_INTERNAL on it.
GW_HD double gw_inv_c_lo(double c) { return (1.0 / c) * (1.0 - (1.0 / 2147483648.0) / c); }
GW_HD void gw_tick_jump_lo(double wake, double t, double delta, double inv_c_lo, bool inclusive,
                           uint32_t* n, double* wake_out, bool* tie, bool* sane)
{
    const double d = t - wake;                            // exact (span precondition)
    const double n0 = floor(d * inv_c_lo);
    const double r = fma(-n0, delta, d);                  // exact; in [0, 2 delta)
    const bool high = r >= delta;
    const bool hit = (r == 0.0) || (r == delta);          // a tick falls exactly on t
    const int32_t ni = (int32_t)n0 + (high ? 1 : 0);      // index of the last tick <= t (-1: none)
    int32_t cnt = ni + ((inclusive || !hit) ? 1 : 0);
    if (!inclusive && cnt < 0) cnt = 0;                   // t exactly ON the last tick already counted (ni = -1, hit)
    const double nw = fma((double)cnt, delta, wake);
    union { double f; uint64_t u; } rb;                   // (sign bit of the residual: an integer test; an exact zero is +0)
    rb.f = r;
    *sane = (int32_t)(rb.u >> 32) >= 0 && (inclusive ? (nw > t) : (nw >= t));
    *n = (uint32_t)cnt;
    *wake_out = nw;
    *tie = inclusive && hit && ni >= 0;
}

// ---- host-side validation (plain host functions) ----
inline bool gw_fast_ticks_ok(double c)
{
    if (!(c > 0) || !isfinite(c) || !(c >= 1e-5) || !(c <= 0.0625)) return false;
    const double inv_c = 1.0 / c;
    uint64_t x = 0x9e3779b97f4a7c15ull;
    long jumps = 0, pre_jumps = 0;
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
                double delta = 0;                           // the per-step form on the same case
                if (gw_tick_span_ok(wake, t, c, &delta)) {
                    uint32_t n2 = 0; double wo2 = 0; bool tie2 = false, inr = false;
                    gw_tick_jump_pre(wake, t, delta, inv_c, incl != 0, &n2, &wo2, &tie2, &inr);
                    if (!inr || n2 != nl || !(wo2 == w) || tie2 != tl) return false;
                    gw_tick_jump_lo(wake, t, delta, gw_inv_c_lo(c), incl != 0, &n2, &wo2, &tie2, &inr);
                    if (!inr || n2 != nl || !(wo2 == w) || tie2 != tl) return false;
                    ++pre_jumps;
                }
            }
        }
    }
    if (!(jumps > 50000) || !(pre_jumps > 25000)) return false;   // they must actually apply (not decline everywhere)
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
    // the per-step form (gw_tick_span_ok once, then chained gw_tick_jump_pre calls from the running base) on the same true
    // sequence: spans of up to 60 ticks starting at every 7th tick, 1-6 chained jumps each to increasing targets at, just
    // below, just above and between ticks, both inclusive and exclusive; `inrange` must never be false
    long spans = 0;
    const int max_span = (0.06 / c) < 1.0 ? 1 : ((0.06 / c) > 60.0 ? 60 : (int)(0.06 / c));   // a span is shorter than 2^-4 s
    for (int j = 0; j < M - 64 && ok; j += 7) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const int span = 1 + (int)(x % (uint64_t)max_span);
        const double t_last = W[j + span] - (double)((x >> 8) % 3) * 0.25 * c;
        double delta = 0;
        if (!gw_tick_span_ok(W[j], t_last, c, &delta)) continue;
        ++spans;
        int base = j;                                      // index of the next uncounted tick
        double wake = W[j];
        const int hops = 1 + (int)((x >> 16) % 6);
        double t_prev = j > 0 ? W[j - 1] : W[j];           // a target is never below the last tick already counted
        for (int h = 0; h < hops && ok; ++h) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            int m = base + (int)(x % 8);
            if (m > j + span) m = j + span;
            double t;
            switch ((x >> 20) & 3) {
            case 0:  t = W[m]; break;
            case 1:  t = nextafter(W[m], 0.0); break;
            case 2:  t = nextafter(W[m], 1e30); break;
            default: t = W[m] + (W[m + 1] - W[m]) * 0.61; break;
            }
            if (t > t_last) t = t_last;
            if (t < t_prev) t = t_prev;                    // targets never move backwards (nor below the previous base - c)
            const bool incl = ((x >> 24) & 1) != 0;
            uint32_t n = 0; double wo = 0; bool tie = false, inr = false;
            gw_tick_jump_pre(wake, t, delta, inv_c, incl, &n, &wo, &tie, &inr);
            int last = base - 1;
            while (last + 1 < M + 64 && (incl ? (W[last + 1] <= t) : (W[last + 1] < t))) ++last;
            const bool tl = incl && last >= base && W[last] == t;
            if (!inr || (int)n != last - base + 1 || !(wo == W[last + 1]) || tie != tl) ok = false;
            gw_tick_jump_lo(wake, t, delta, gw_inv_c_lo(c), incl, &n, &wo, &tie, &inr);
            if (!inr || (int)n != last - base + 1 || !(wo == W[last + 1]) || tie != tl) ok = false;
            base = last + 1;
            wake = wo;
            t_prev = t;
        }
    }
    delete[] W;
    return ok && used > 100000 && spans > 50000;
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
        if (!(gw_fast_fmod_lo(t, slot, gw_inv_lo(slot)) == b)) return false;
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
