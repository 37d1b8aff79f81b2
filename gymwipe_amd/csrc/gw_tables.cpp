// gw_tables.cpp -- host-side static link tables for the vectorised CounterTraffic step.
//
// Geometry is static in the band-assignment envs, so everything that needs a
// transcendental is evaluated ONCE per handle here, on the host, through the same
// glibc libm entry points CPython uses (log10 / pow / sqrt) -- bit-identical to the
// reference's Python expressions -- and the kernels only do +,-,*,/,fmod,rint in f64.
//
// The one piece of per-env floating-point state the PHY model carries is
// phy._receivedPower (simple_stack.py:80-86): thermal noise plus the residue of
// every (+p, -p) pair it has seen.  With at most one transmission on the air
// (contention-free MAC), hearing sender `i` maps the resting value a to
//      g(a, p) = fl(fl(a + p) - p),
// and that SAME value is the noise term during the reception (noise = rx - p,
// simple_stack.py:166-167).  g lands on the ulp grid of p, so the set of reachable
// resting values per radio is tiny (<= 8 for 32 senders on the default circle).  We
// enumerate that closure here and let the kernels carry a one-byte state index per
// radio instead of an f64, with the BER of every (listener, talker, state) tabulated.
// This is exact, not an approximation: tests compare the reconstructed f64 with the
// oracle's running value bit for bit.
#include "gw_internal.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

namespace {

// devices/core.py:88-95 (x**2 is pow(x, 2.0) in CPython) + attenuation_models.py:28-36
double fspl_db(const gw_config& c, int a, int b)
{
    const double ax = c.pos[a][0], ay = c.pos[a][1], bx = c.pos[b][0], by = c.pos[b][1];
    if (ax == bx && ay == by) return 0.0;          // co-located: the model keeps attenuation 0
    const double dist = sqrt(pow(ax - bx, 2.0) + pow(ay - by, 2.0));
    return 20 * log10(dist) + 20 * log10(c.frequency) - 147.55;
}

// physical.py:25-58 (Eb/N0, Q approximation), :82-98 (dBm helpers), :208-212 (BPSK)
double ber_bpsk(const gw_config& c, double sig_mw, double noise_mw)
{
    volatile double euler = M_E;                   // keep pow() a runtime libm call
    const double s = 10 * log10(sig_mw);
    const double n = 10 * log10(noise_mw);
    if (s <= n) return 0.5;
    const double ratio_db = s - n - 10 * log10(c.bit_rate);
    const double ratio = pow(10.0, ratio_db / 10);
    const double x = sqrt(2 * ratio);
    const double sqrt2pi = sqrt(2 * M_PI);
    return (1 - pow(euler, -1.4 * x)) * pow(euler, -(pow(x, 2.0) / 2)) / (1.135 * sqrt2pi * x);
}

// Decode certainty of a link with bit error rate `ber` (simple_stack.py:180-188,269-286).
// Only claimed when the integer decision rule applies (maxBER == 0.25, integral coded bits):
//   ok  <=>  4*round(err) <= codedBits,  err_hdr = ber*(t_h - t_s)*bitRate,
//                                        err_pay = 2*ber*(t_e - t_h)*bitRate   (counted twice).
// The event-time differences equal the nominal durations up to f64 rounding of times below
// fmod_limit (< 1.2e6 s): relative 1e-6 for the header, 1e-5 for payloads of >= 1 byte.  With
// |round(x) - x| <= 0.5 the outcome is CERTAIN for every payload size p >= 1 when
//   header : 4*(ber*hd*br*(1+1e-6) + 0.5) <= Hb            (ok)   /  4*(ber*hd*br*(1-1e-6) - 0.5) > Hb  (fail)
//   payload: p*(8cf - 64*ber*(br/dr)*(1+1e-5)) >= 2 at p=1  (ok)   /  p*(64*ber*(br/dr)*(1-1e-5) - 8cf) > 2 (fail)
// Anything else stays GW_CLS_COMPUTE and is decided by the exact arithmetic in the kernel.
uint8_t decode_class(const gw_config& c, const GwHostTables& t, double ber)
{
    const double cf = t.coded_factor, dr = t.data_rate, br = c.bit_rate;
    if (!(c.max_ber == 0.25) || cf * 8.0 != floor(cf * 8.0)) return GW_CLS_COMPUTE;
    const double hb = (double)(c.mac_header_bytes * 8);
    const double Hb = hb * cf, hd = hb / dr;
    const double eh = ber * hd * br;
    const bool hdr_ok = 4.0 * (eh * (1 + 1e-6) + 0.5) <= Hb * (1 - 1e-12);
    const bool hdr_fail = 4.0 * (eh * (1 - 1e-6) - 0.5) > Hb * (1 + 1e-12);
    if (hdr_fail) return GW_CLS_HDR_FAIL;
    if (!hdr_ok) return GW_CLS_COMPUTE;
    const double g = 64.0 * ber * (br / dr);
    if (8.0 * cf - g * (1 + 1e-5) >= 2.0 * (1 + 1e-12)) return GW_CLS_OK;
    if (g * (1 - 1e-5) - 8.0 * cf > 2.0 * (1 + 1e-12)) return GW_CLS_PAY_FAIL;
    return GW_CLS_COMPUTE;
}

} // namespace

int gw_build_tables(const gw_config& cfg, GwHostTables& t, char* msg, size_t msglen)
{
    memset(&t, 0, sizeof t);
    const int D = cfg.num_devices, R = D + 1;
    t.D = D; t.R = R;
    t.thermal = 1.38e-23 * (cfg.temperature_c + 273.15) * cfg.bandwidth * 1000;   // physical.py:71, simple_stack.py:77
    t.data_rate = cfg.code_rate * cfg.bit_rate;                                     // physical.py:197
    t.coded_factor = 2 - cfg.code_rate;                                             // physical.py:259-263
    for (int a = 0; a < R; ++a)
        for (int b = 0; b < R; ++b) {
            if (a == b) continue;
            t.att[a][b] = fspl_db(cfg, a, b);
            if (cfg.extra_att_db[a][b] != 0.0) {                                    // joined model: sum([fspl, custom]) = (0 + fspl) + custom
                volatile double joined = 0.0 + t.att[a][b];
                joined = joined + cfg.extra_att_db[a][b];
                t.att[a][b] = joined;
            }
            t.prx[a][b] = pow(10.0, (cfg.tx_power_dbm - t.att[a][b]) / 10);        // simple_stack.py:111
        }

    // closure of the resting rx-power values of each listener j under g(., p_ij)
    for (int j = 0; j < R; ++j) {
        int n = 1;
        t.state_val[j][0] = t.thermal;
        for (int head = 0; head < n; ++head) {
            const double a = t.state_val[j][head];
            for (int i = 0; i < R; ++i) {
                if (i == j) continue;
                const double p = t.prx[i][j];
                volatile double up = a + p;                      // simple_stack.py:82 (+p)
                volatile double b = up + (-p);                   // simple_stack.py:154 (-p)
                int idx = -1;
                for (int k = 0; k < n; ++k)
                    if (memcmp(&t.state_val[j][k], (const void*)&b, sizeof(double)) == 0) { idx = k; break; }
                if (idx < 0) {
                    if (n == GW_MAX_NSTATES) {
                        // the (+p, -p) residue of this radio does not close into a small set (it drifts by an ulp per
                        // packet): no state machine for this geometry -- the live-PHY kernel carries the f64 itself
                        snprintf(msg, msglen, "rx-power state closure of radio %d exceeds %d states", j, GW_MAX_NSTATES);
                        t.overflow = 1;
                        for (int r = 0; r < R; ++r) if (t.nstates[r] == 0) t.nstates[r] = 1;
                        return GW_OK;
                    }
                    idx = n;
                    t.state_val[j][n++] = b;
                }
                const size_t at = ((size_t)j * R + i) * GW_MAX_NSTATES + head;
                t.trans[at] = (uint8_t)idx;
            }
        }
        t.nstates[j] = n;
        // BER while j listens to i, indexed by the state AFTER the transition
        // (noise = fl(a+p) - p = new resting value)            simple_stack.py:161-173
        for (int i = 0; i < R; ++i) {
            if (i == j) continue;
            for (int s = 0; s < n; ++s) {
                const double noise = t.state_val[j][s];
                if (!(noise >= 0)) {
                    snprintf(msg, msglen, "negative noise power for radio %d (the reference asserts here)", j);
                    return GW_EUNSUPPORTED;
                }
                const size_t at = ((size_t)j * R + i) * GW_MAX_NSTATES + s;
                t.ber[at] = ber_bpsk(cfg, t.prx[i][j], noise);
                t.cls[at] = decode_class(cfg, t, t.ber[at]);
            }
        }
    }
    return GW_OK;
}
