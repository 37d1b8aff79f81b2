// ct_step_sfx.hip -- the default step kernel.  MAC queues in the "suffix" encoding of gw_queue.h:
// one length byte per sender, a tick counter and the reset breakpoints per env.  A counter tick is
// `len = min(len + mult, 100)`, a pop is `len -= 1`, the head packet's size is arithmetic on the tick
// index -- no queue memory is touched by the step at all, so the kernel issues every load up front,
// walks the step's event horizon in registers, and stores once.
//
// Step walk, reference file:line as in ct_common.hip.h / ct_step.hip:
//   A.1  t_s = t_a + (slot - t_a % slot)                                   simtools.py:44-53
//   A.2  announcement heard by the addressed sender (header, payload)      simple_stack.py:214-286,536-558
//   A.3  window: pop + transmit while (stop - now) > bits/dataRate         simple_stack.py:397-434
//   A.4  RRM decodes each data packet -> interpreter                       networking/devices.py:163-168
//   A.5  t_end = t_r + (slots+1)*slot; ticks every 1 ms (running f64 sum)   counter_traffic.py:53-61
//   A.6  equal-time events: earlier-inserted first; process initialisation URGENT.
#include "ct_common.hip.h"
#include "gw_queue.h"

using namespace gwk;

namespace {

template <int DT, bool PER_ENV_STATS>
__global__ __launch_bounds__(256) void ct_step_sfx_kernel(GwState st,
                                                         const int32_t* __restrict__ device,
                                                         const int32_t* __restrict__ duration,
                                                         int32_t* __restrict__ obs,
                                                         float* __restrict__ reward,
                                                         uint8_t* __restrict__ done)
{
    const int64_t N = st.N;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const GwDevConst& c = *st.cst;
    const int D = DT > 0 ? DT : c.D;
    const int R = D + 1, S = c.S, RRM = D;

    Tally k = {0, 0, 0, 0, 0};
    uint32_t k_bad = 0, k_steps = 0, fl_new = 0;

    if (e < N) {
        const int d = device[e];
        const int du = duration[e];
        uint32_t rvm = st.rvmask[e];
        int32_t last_abs = st.last_abs[e];
        uint8_t dn = st.done[e];
        const int pv = c.payload_value;
        uint32_t fl = 0;

        if ((unsigned)d >= (unsigned)D || (unsigned)du >= (unsigned)c.max_duration) {
            // counter_traffic.py:147 asserts; a batched step cannot raise per env: flag + skip
            fl = GW_FLAG_BADACT;
            k_bad = 1;
            const int32_t latest = pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u));
            obs[e] = latest + c.counter_bound;
            reward[e] = 0.0f;
            done[e] = dn;
        } else {
            k_steps = 1;
            const StepMath m(c);
            const double slot = c.slot, br = c.bit_rate;
            const double hd = c.hdr_dur, hdr_bits = c.hdr_bits;
            const double interval = c.counter_interval;
            const uint32_t bound = (uint32_t)c.counter_bound;
            const uint32_t base_bytes = (uint32_t)(c.mac_hdr + c.net_hdr);
            const int mh = c.mac_hdr;
            const bool idem = c.idem_states != 0;

            // ---- every load of the step is issued up here, before anything is stored ------------
            const double t_a = st.now[e];
            double wake = st.wake[e];
            const uint32_t tau0 = st.tau[e];
            const uint32_t nbp = st.nbp[e];
            const GwBp bpc = st.bpc[e];
            const GwBp bpp = st.bpp[e];
            const GwBp* hist = st.bph + ((int64_t)e << 7);
            uint32_t len_d = st.qlen[(int64_t)d * N + e];
            uint8_t li[DT > 0 ? DT : 1], sj[DT > 0 ? DT : 1];
            if (DT > 0) {
#pragma unroll
                for (int i = 0; i < DT; ++i) { li[i] = st.qlen[(int64_t)i * N + e]; sj[i] = st.rxs[(int64_t)i * N + e]; }
            }
            const uint8_t s_d_old = st.rxs[(int64_t)d * N + e];
            const uint8_t s_r_old = st.rxs[(int64_t)RRM * N + e];
            // what the addressed sender / the RRM become after hearing the RRM / sender d once
            const uint8_t s_d = st.trans[((int64_t)d * R + RRM) * S + s_d_old];
            const double ber_a = st.ber[((int64_t)d * R + RRM) * S + s_d];
            uint8_t s_r = s_r_old;
            const uint8_t s_r1 = st.trans[((int64_t)RRM * R + d) * S + s_r_old];
            const double ber_x1 = st.ber[((int64_t)RRM * R + d) * S + s_r1];
            const uint32_t mult_d = (uint32_t)c.mult[d];
            const uint32_t inv16_d = c.inv16[d];

            const int slots = du * c.duration_factor;                     // counter_traffic.py:149

            // ---- A.1 / A.2: announcement ------------------------------------------------------
            const int L = ndigits(slots);
            const TxTimes an = tx_times(m, t_a, hd, m.over_rate((double)(L * 8)));
            k.tx++;
            const bool granted = receive(m, ber_a, an, br, hdr_bits, (double)(L * 8) * c.coded_factor, fl);
            const double t_r = an.t_e;
            const double t_end = t_r + (double)(slots + 1) * slot;       // simple_stack.py:557-558

            // ---- A.3: window at sender d ------------------------------------------------------
            uint32_t tau = tau0;
            int n_data = 0;
            Tally kd = {0, 0, 0, 0, 0};                                   // appends/drops of sender d only

            // all counter ticks with wake < t (or <= t): counted in f64, applied to d's queue length
            auto ticks_to = [&](double t, bool inclusive) {
                uint32_t kk = 0;
                while (wake < t || (inclusive && wake == t)) {
                    if (wake == t) fl |= GW_FLAG_TIE;
                    wake = wake + interval;                               // running sum, not k*dt
                    kk++;
                }
                tau += kk;
                len_d = gw_len_after_ticks(len_d, kk, mult_d, kd);
            };

            if (granted) {
                const double total = (double)slots * slot;               // simple_stack.py:400
                const double stopw = t_r + total;                        // :401 (== timeout time :406)
                double cur = t_r;
                // ties at the window start: the MAC's process initialisation is URGENT, so it runs first
                ticks_to(cur, false);
                for (;;) {
                    if (len_d == 0) {                                     // :409-416
                        if (wake < stopw) {
                            cur = wake;
                            wake = wake + interval;
                            tau++;
                            len_d = gw_len_after_ticks(0u, 1u, mult_d, kd);
                        } else break;
                    }
                    const uint32_t age = gw_ceil_div(len_d, mult_d, inv16_d);
                    const uint32_t s = base_bytes + gw_tick_value(tau - age, bpc, bpp, nbp, hist, bound);
                    const double need = m.over_rate((double)(s * 8u));    // messages.py:67-75
                    if (!((stopw - cur) > need)) break;                   // :418-420 idle until the window ends
                    len_d--;                                              // :425
                    k.pop++;
                    const int pay = (int)s - mh;
                    const TxTimes x = tx_times(m, cur, hd, m.over_rate((double)(pay * 8)));
                    k.tx++;
                    n_data++;
                    double ber_x = ber_x1;
                    if (idem) {
                        s_r = s_r1;
                    } else {
                        s_r = st.trans[((int64_t)RRM * R + d) * S + s_r];
                        ber_x = st.ber[((int64_t)RRM * R + d) * S + s_r];
                    }
                    const bool ok = receive(m, ber_x, x, br, hdr_bits, (double)(pay * 8) * c.coded_factor, fl);
                    if (ok) {                                             // devices.py:163-168, counter_traffic.py:75-80
                        k.deliv++;
                        rvm |= (1u << d);
                        if (pv == c.counter_bound) dn = 1;
                    }
                    if (!(x.t_e < t_end)) fl |= GW_FLAG_CARRY;
                    ticks_to(x.t_e, true);                                // ticks are older events than the MAC's resume
                    cur = x.t_e;
                    if (!(cur < stopw)) break;                            // window timeout already processed
                }
            }

            // ---- A.5: remaining ticks up to the end of the step -------------------------------
            ticks_to(t_end, true);
            const uint32_t n_ticks = tau - tau0;
            k.app += kd.app;
            k.drop += kd.drop;

            // ---- rx-power state of the radios that only listened (simple_stack.py:130-157) -----
            // (table lookups are loads too: they come before the first store)
            auto heard = [&](int j, uint8_t s0) {
                uint8_t s = st.trans[((int64_t)j * R + RRM) * S + s0];
                for (int n = 0; n < n_data; ++n) {
                    const uint8_t s2 = st.trans[((int64_t)j * R + d) * S + s];
                    if (s2 == s) break;                                   // fixed point: g(g(a,p),p) == g(a,p)
                    s = s2;
                }
                return s;
            };
            uint8_t sn[DT > 0 ? DT : 1];
            if (DT > 0) {
#pragma unroll
                for (int j = 0; j < DT; ++j) sn[j] = heard(j, sj[j]);
            }

            // ---- stores ------------------------------------------------------------------------
            st.qlen[(int64_t)d * N + e] = (uint8_t)len_d;
            // every other sender saw the same n_ticks ticks
            if (DT > 0) {
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    if (i == d) continue;
                    st.qlen[(int64_t)i * N + e] = (uint8_t)gw_len_after_ticks(li[i], n_ticks, (uint32_t)c.mult[i], k);
                    if (sn[i] != sj[i]) st.rxs[(int64_t)i * N + e] = sn[i];
                }
            } else {
                for (int i = 0; i < D; ++i) {
                    if (i == d) continue;
                    const uint32_t l0 = st.qlen[(int64_t)i * N + e];
                    const uint8_t s0 = st.rxs[(int64_t)i * N + e];
                    const uint8_t s1 = heard(i, s0);
                    st.qlen[(int64_t)i * N + e] = (uint8_t)gw_len_after_ticks(l0, n_ticks, (uint32_t)c.mult[i], k);
                    if (s1 != s0) st.rxs[(int64_t)i * N + e] = s1;
                }
            }
            if (s_d != s_d_old) st.rxs[(int64_t)d * N + e] = s_d;         // d hears only the announcement
            if (s_r != s_r_old) st.rxs[(int64_t)RRM * N + e] = s_r;

            // ---- interpreter feedback (counter_traffic.py:85-112, envs/core.py:142-153) -----------
            const int32_t latest = pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u));
            const int32_t abs_d = latest < 0 ? -latest : latest;
            int32_t r = last_abs - abs_d;
            last_abs = abs_d;
            r = r > 10 ? 10 : (r < -10 ? -10 : r);
            obs[e] = latest + c.counter_bound;
            reward[e] = (float)r;
            done[e] = dn;

            st.now[e] = t_end;
            st.wake[e] = wake;
            st.tau[e] = tau;
            st.rvmask[e] = rvm;
            st.last_abs[e] = last_abs;
            st.done[e] = dn;
            if (PER_ENV_STATS) {
                st.pe_stats[0 * N + e] += k.tx;
                st.pe_stats[1 * N + e] += k.deliv;
                st.pe_stats[2 * N + e] += k.app;
                st.pe_stats[3 * N + e] += k.pop;
                st.pe_stats[4 * N + e] += k.drop;
            }
        }
        if (fl) st.flags[e] |= fl;                                         // rare: sticky flags
        fl_new = fl;
    }
    publish_totals(st.totals, k, k_steps, k_bad, fl_new);
}

// counter_traffic.py:135-144 + :69-73 -- counters and interpreter only; time is NOT rewound.
// In the suffix encoding "counters <- 0" is a new breakpoint (tau, 0).
__global__ void ct_reset_sfx_kernel(GwState st, const uint8_t* __restrict__ mask, int32_t* __restrict__ obs)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= st.N) return;
    if (!mask || mask[e]) {
        const uint32_t tau = st.tau[e];
        const uint32_t n = st.nbp[e];
        GwBp cur = st.bpc[e];
        if (cur.t0 == tau) {                       // no tick since the last breakpoint: overwrite it
            cur.c0 = 0u;
            st.bpc[e] = cur;
            st.bph[((int64_t)e << 7) + ((n - 1u) & GW_RING_MASK)] = cur;
        } else {
            st.bpp[e] = cur;
            cur.t0 = tau; cur.c0 = 0u;
            st.bpc[e] = cur;
            st.bph[((int64_t)e << 7) + (n & GW_RING_MASK)] = cur;
            st.nbp[e] = n + 1u;
        }
        st.rvmask[e] = 0u;
        st.last_abs[e] = 0;
        st.done[e] = 0;
    }
    if (obs) {
        const uint32_t rvm = st.rvmask[e];
        obs[e] = st.cst->payload_value * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u)) + st.cst->counter_bound;
    }
}

template <int DT>
int launch(const GwState& st, const int32_t* device, const int32_t* duration,
           int32_t* obs, float* reward, uint8_t* done, void* stream)
{
    const unsigned blk = (unsigned)st.block;
    const unsigned grid = (unsigned)((st.N + blk - 1) / blk);
    if (st.pe_stats)
        hipLaunchKernelGGL((ct_step_sfx_kernel<DT, true>), dim3(grid), dim3(blk), 0, (hipStream_t)stream,
                           st, device, duration, obs, reward, done);
    else
        hipLaunchKernelGGL((ct_step_sfx_kernel<DT, false>), dim3(grid), dim3(blk), 0, (hipStream_t)stream,
                           st, device, duration, obs, reward, done);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}

} // namespace

int gw_launch_step_sfx(const GwState& st, const int32_t* device, const int32_t* duration,
                       int32_t* obs, float* reward, uint8_t* done, void* stream)
{
    switch (st.D) {
    case 2:  return launch<2>(st, device, duration, obs, reward, done, stream);
    case 4:  return launch<4>(st, device, duration, obs, reward, done, stream);
    case 8:  return launch<8>(st, device, duration, obs, reward, done, stream);
    case 16: return launch<16>(st, device, duration, obs, reward, done, stream);
    default: return launch<0>(st, device, duration, obs, reward, done, stream);
    }
}

int gw_launch_reset_sfx(const GwState& st, const uint8_t* mask, int32_t* obs, void* stream)
{
    const unsigned grid = (unsigned)((st.N + 255) / 256);
    hipLaunchKernelGGL(ct_reset_sfx_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, st, mask, obs);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}
