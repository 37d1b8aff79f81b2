// ct_step_dyn.hip -- the step kernel with a LIVE physical layer: every radio carries its f64 received power
// (phy._receivedPower, simple_stack.py:77-86) and bit error rates are evaluated on the device, instead of the default
// kernel's one-byte noise-state machine over host tables.  Two cases need it:
//   * a static geometry whose rx-power residue never closes into a small state set (the (+p, -p) pairs of
//     simple_stack.py:81-86,154 drift by an ulp per packet for some layouts: gw_tables.cpp finds > 16 states) --
//     link powers still come from the host's glibc tables, the received powers are exact f64 sums;
//   * GW_CFG_PER_ENV_GEOMETRY: positions[N][R][2] per environment, Position.set between steps
//     (devices/core.py:52-86 -> PositionalAttenuationModel, physical.py:380-386 -> FsplAttenuation,
//     attenuation_models.py:28-36): link powers per env, rebuilt on the device by gw_set_position(s).
// MAC queues keep the exact suffix encoding of the default kernel (gw_queue.h): queues do not depend on the PHY.
// (With GW_CFG_EXPLICIT_QUEUE the same live PHY runs inside the generic kernel: ct_step.hip, instantiation DYN.)
// Same walk as ct_step_sfx.hip; what differs is A.2/A.4: every transmission i -> j adds its
// power to the listeners' received power, the receiver's BER is physical.py:25-58,208-212 on the device libm
// (log10 / pow / sqrt: last-ulp differences from CPython's libm -- they only enter error sums that are rounded to
// integers, so decisions agree except on a measure-zero boundary), and the power is subtracted again.
#include "ct_common.hip.h"
#include "gw_queue.h"

using namespace gwk;

namespace {

template <class T>
__device__ __forceinline__ T ld(const void* base, uint32_t byte_off)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ void st_(void* base, uint32_t byte_off, const T& v)
{
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

template <bool PER_ENV>
__global__ __launch_bounds__(256) void ct_step_dyn_kernel(GwState st, GwDevConst c,
                                                          const int32_t* __restrict__ device,
                                                          const int32_t* __restrict__ duration,
                                                          int32_t* __restrict__ obs,
                                                          float* __restrict__ reward,
                                                          uint8_t* __restrict__ done)
{
    const uint32_t N = (uint32_t)st.N;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int D = c.D, R = D + 1, RRM = D;
    const uint32_t RB = (uint32_t)st.RB;
    const uint32_t o16 = e << 4, oq = e * RB;

    const int d = device[e];
    const int du = duration[e];
    const uint4 ip = ld<uint4>(st.ip, o16);
    const double2 tw = ld<double2>(st.tw, o16);
    const uint4 tk = ld<uint4>(st.tk, o16);

    const StepMath m(c);
    uint32_t rvm = tk.z;                          // (record layout: ct_step_sfx.hip)
    int32_t last_abs = (int32_t)(tk.w & 0x7fffffffu);
    uint32_t dn = tk.w >> 31;
    uint32_t fl = 0, k_bad = 0;
    Tally k = {0, 0, 0, 0, 0};
    const int pv = c.payload_value;

    if ((unsigned)d >= (unsigned)D || (unsigned)du >= (unsigned)c.max_duration) {
        // counter_traffic.py:147 asserts; a batched step cannot raise per env: flag + skip
        fl = GW_FLAG_BADACT;
        k_bad = 1;
        const int32_t latest = pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u));
        obs[e] = latest + c.counter_bound;
        reward[e] = 0.0f;
        done[e] = (uint8_t)dn;
    } else {
        const double slot = c.slot, br = c.bit_rate, hd = c.hdr_dur, hdr_bits = c.hdr_bits, interval = c.counter_interval;
        const uint32_t bound = (uint32_t)c.counter_bound, base_bytes = (uint32_t)(c.mac_hdr + c.net_hdr);
        const int mh = c.mac_hdr;
        const double ten_log_br = c.ten_log_br;

        uint32_t len_d = st.qb[oq + (uint32_t)d];
        const uint32_t mult_d = (uint32_t)c.mult[d], inv16_d = c.inv16[d];
        const double t_a = tw.x;
        double wake = tw.y;
        const uint32_t tau0 = tk.x, nbp = tk.y;
        GwBp bpc, bpp;
        bpc.t0 = ip.x; bpc.c0 = ip.y;
        bpp.t0 = ip.z; bpp.c0 = ip.w;
        const GwBp* hist = st.bph + ((size_t)e << 7);

        const int slots = du * c.duration_factor;                         // counter_traffic.py:149

        // ---- A.1 / A.2: announcement, heard by the addressed sender ------------------------------------
        const int Ld = ndigits(slots);
        const TxTimes an = tx_times(m, t_a, hd, m.over_rate((double)(Ld * 8)));
        k.tx++;
        const double p_a = gw_link<PER_ENV>(st, R, RRM, d, e);               // simple_stack.py:111
        const double rx_d0 = st.rxp[(size_t)d * N + e];
        const double up_d = rx_d0 + p_a;                                  // :82  (+p) at the start of the transmission
        const double noise_d = up_d - p_a;                                // :166-167 noise = received - signal
        if (!(noise_d >= 0.0)) fl |= GW_FLAG_REFEXC;                      // :168 assert noisePower >= 0
        const double ber_a = ber_bpsk_dev(p_a, noise_d, ten_log_br);
        const bool granted = receive(m, ber_a, an, br, hdr_bits, (double)(Ld * 8) * c.coded_factor, fl);
        if (!(an.t_e >= an.stop)) fl |= GW_FLAG_REFEXC;
        st.rxp[(size_t)d * N + e] = up_d + (-p_a);                        // :154 (-p) when it completes
        const double t_r = an.t_e;
        const double t_end = t_r + (double)(slots + 1) * slot;            // simple_stack.py:557-558

        // ---- A.3: window at sender d -------------------------------------------------------------------
        uint32_t tau = tau0;
        int n_data = 0;
        Tally kd = {0, 0, 0, 0, 0};
        auto ticks_to = [&](double t, bool inclusive) __attribute__((always_inline)) {
            uint32_t kk = 0;
            for (;;) {
                const double w1 = wake + interval, w2 = w1 + interval, w3 = w2 + interval, w4 = w3 + interval;
                const bool b0 = inclusive ? (wake <= t) : (wake < t);
                const bool b1 = inclusive ? (w1 <= t) : (w1 < t);
                const bool b2 = inclusive ? (w2 <= t) : (w2 < t);
                const bool b3 = inclusive ? (w3 <= t) : (w3 < t);
                if (inclusive && (wake == t || w1 == t || w2 == t || w3 == t)) fl |= GW_FLAG_TIE;
                kk += (uint32_t)b0 + (uint32_t)b1 + (uint32_t)b2 + (uint32_t)b3;
                wake = b3 ? w4 : (b2 ? w3 : (b1 ? w2 : (b0 ? w1 : wake)));
                if (!b3) break;
            }
            tau += kk;
            len_d = gw_len_after_ticks(len_d, kk, mult_d, kd);
        };

        const double p_x = gw_link<PER_ENV>(st, R, d, RRM, e);               // the RRM hears sender d
        double rx_r = st.rxp[(size_t)RRM * N + e];
        const double rx_r0 = rx_r;
        double ber_x = 0.0, noise_prev = -1.0;
        if (granted) {
            const double total = (double)slots * slot;                    // simple_stack.py:400
            const double stopw = t_r + total;                             // :401
            double cur = t_r;
            ticks_to(cur, false);                                         // the MAC's process initialisation is URGENT
            for (;;) {
                if (len_d == 0) {                                         // :409-416
                    if (mult_d != 0u && wake < stopw) {
                        cur = wake;
                        wake = wake + interval;
                        tau++;
                        len_d = gw_len_after_ticks(0u, 1u, mult_d, kd);
                    } else break;
                }
                const uint32_t age = gw_ceil_div(len_d, mult_d, inv16_d);
                const uint32_t s = base_bytes + gw_tick_value(tau - age, bpc, bpp, nbp, hist, bound);
                const double need = m.over_rate((double)(s * 8u));        // messages.py:67-75
                if (!((stopw - cur) > need)) break;                       // :418-420
                len_d--;                                                  // :425
                k.pop++;
                const int pay = (int)s - mh;
                const TxTimes x = tx_times(m, cur, hd, m.over_rate((double)(pay * 8)));
                k.tx++;
                n_data++;
                const double up = rx_r + p_x;
                const double noise = up - p_x;
                if (!(noise >= 0.0)) fl |= GW_FLAG_REFEXC;
                if (noise != noise_prev) { ber_x = ber_bpsk_dev(p_x, noise, ten_log_br); noise_prev = noise; }
                if (!(x.t_e >= x.stop)) fl |= GW_FLAG_REFEXC;
                const bool ok = receive(m, ber_x, x, br, hdr_bits, (double)(pay * 8) * c.coded_factor, fl);
                rx_r = up + (-p_x);
                k.deliv += ok ? 1u : 0u;                                  // devices.py:163-168, counter_traffic.py:75-80
                rvm |= ok ? (1u << d) : 0u;
                dn = (ok && pv == c.counter_bound) ? 1u : dn;
                fl |= !(x.t_e < t_end) ? (uint32_t)GW_FLAG_CARRY : 0u;
                ticks_to(x.t_e, true);
                cur = x.t_e;
                if (!(cur < stopw)) break;
            }
        }
        if (rx_r != rx_r0) st.rxp[(size_t)RRM * N + e] = rx_r;

        // ---- A.5: remaining ticks up to the end of the step ------------------------------------------------
        {
            uint32_t nj = 0;
            double wj = wake;
            bool tiej = false;
            if (c.fast_ticks && gw_tick_jump(wake, t_end, interval, c.inv_interval, true, &nj, &wj, &tiej)) {
                wake = wj;
                tau += nj;
                if (tiej) fl |= GW_FLAG_TIE;
                len_d = gw_len_after_ticks(len_d, nj, mult_d, kd);
            } else {
                ticks_to(t_end, true);
            }
        }
        const uint32_t n_ticks = tau - tau0;
        k.app += kd.app;
        k.drop += kd.drop;

        // ---- every other sender saw the same ticks and heard the announcement and d's data
        //      (simple_stack.py:130-157: += p at the start, += -p at the end of each transmission) ---------
        st.qb[oq + (uint32_t)d] = (uint8_t)len_d;
        for (int j = 0; j < D; ++j) {
            if (j == d) continue;
            const uint32_t l0 = st.qb[oq + (uint32_t)j];
            st.qb[oq + (uint32_t)j] = (uint8_t)gw_len_after_ticks(l0, n_ticks, (uint32_t)c.mult[j], k);
            const double a0 = st.rxp[(size_t)j * N + e];
            const double pa = gw_link<PER_ENV>(st, R, RRM, j, e);
            double a = (a0 + pa) + (-pa);
            if (n_data) {
                const double pd = gw_link<PER_ENV>(st, R, d, j, e);
                for (int n = 0; n < n_data; ++n) {
                    const double b = (a + pd) + (-pd);
                    if (b == a) break;                                    // a fixed point of the (+p, -p) pair stays one
                    a = b;
                }
            }
            if (!(a >= 0.0)) fl |= GW_FLAG_REFEXC;
            if (a != a0) st.rxp[(size_t)j * N + e] = a;
        }

        // ---- interpreter feedback (counter_traffic.py:85-112, envs/core.py:142-153) -----------------------
        const int32_t latest = pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u));
        const int32_t abs_d = latest < 0 ? -latest : latest;
        int32_t r = last_abs - abs_d;
        last_abs = abs_d;
        r = r > 10 ? 10 : (r < -10 ? -10 : r);
        obs[e] = latest + c.counter_bound;
        reward[e] = (float)r;
        done[e] = (uint8_t)dn;

        st_(st.tw, o16, make_double2(t_end, wake));
        st_(st.tk, o16, make_uint4(tau, nbp, rvm, (uint32_t)last_abs | (dn << 31)));
    }
    publish_env_counters(st.sa, N, e, k.pop, k.deliv, k_bad, fl, 1u);
}

// FsplAttenuation._update + dbmToMilliwatts with the device libm (devices/core.py:88-95, attenuation_models.py:28-36,
// simple_stack.py:111); co-located radios keep attenuation 0
__device__ __forceinline__ double link_power(double ax, double ay, double bx, double by, double extra_db, double tx_dbm,
                                             double twenty_log_f)
{
    double att = 0.0;
    if (!(ax == bx && ay == by)) {
        const double dist = sqrt(pow(ax - bx, 2.0) + pow(ay - by, 2.0));
        att = 20 * log10(dist) + twenty_log_f - 147.55;
    }
    if (extra_db != 0.0) att = (0.0 + att) + extra_db;                    // joined model: sum([fspl, custom])
    return pow(10.0, (tx_dbm - att) / 10);
}

// Position.set on radio `radio` (or on every radio when radio < 0) of the envs selected by mask, then the attenuation
// models of every link of a moved radio recompute (devices/core.py:77-86 -> physical.py:380-386).  Nothing is on the
// air between two env.step() calls, so no reception is re-integrated (simple_stack.py:119-128 acts on active ones only).
__global__ void ct_set_position_kernel(GwState st, GwDevConst c, int radio, const double* __restrict__ xs,
                                       const double* __restrict__ ys, const double* __restrict__ all_pos,
                                       const uint8_t* __restrict__ mask)
{
    const uint32_t N = (uint32_t)st.N;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    if (mask && !mask[e]) return;
    const int R = c.D + 1;
    if (radio >= 0) {
        st.pos_env[((size_t)radio * 2 + 0) * N + e] = xs[e];
        st.pos_env[((size_t)radio * 2 + 1) * N + e] = ys[e];
    } else {
        for (int r = 0; r < R; ++r) {
            st.pos_env[((size_t)r * 2 + 0) * N + e] = all_pos[((size_t)e * R + r) * 2 + 0];
            st.pos_env[((size_t)r * 2 + 1) * N + e] = all_pos[((size_t)e * R + r) * 2 + 1];
        }
    }
    for (int a = 0; a < R; ++a) {
        if (radio >= 0 && a != radio) continue;
        const double ax = st.pos_env[((size_t)a * 2 + 0) * N + e], ay = st.pos_env[((size_t)a * 2 + 1) * N + e];
        for (int b = 0; b < R; ++b) {
            if (b == a) continue;
            const double bx = st.pos_env[((size_t)b * 2 + 0) * N + e], by = st.pos_env[((size_t)b * 2 + 1) * N + e];
            const double p = link_power(ax, ay, bx, by, st.extra_tab[a * R + b], c.tx_power_dbm, c.twenty_log_f);
            st.prx_env[((size_t)(a * R + b)) * N + e] = p;               // symmetric: attenuation depends on the pair only
            st.prx_env[((size_t)(b * R + a)) * N + e] = p;
        }
    }
}

// fresh handle in the live-PHY mode: every radio at thermal noise; per-env geometry starts as the handle's geometry
// with the host's link powers (bit-identical to the reference until a radio is moved)
__global__ void ct_init_dyn_kernel(GwState st, GwDevConst c, double thermal)
{
    const uint32_t N = (uint32_t)st.N;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int R = c.D + 1;
    for (int r = 0; r < R; ++r) st.rxp[(size_t)r * N + e] = thermal;
    if (st.prx_env) {
        for (int i = 0; i < R * R; ++i) st.prx_env[(size_t)i * N + e] = st.prx_tab[i];
        for (int i = 0; i < R * 2; ++i) st.pos_env[(size_t)i * N + e] = st.pos_tab[i];
    }
}

inline int ok_or_ehip() { return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP; }

} // namespace

int gw_launch_step_dyn(const GwState& st, const GwDevConst& cst, const int32_t* device, const int32_t* duration,
                       int32_t* obs, float* reward, uint8_t* done, void* stream)
{
    const unsigned blk = 64;
    const unsigned grid = (unsigned)((st.N + blk - 1) / blk);
    if (st.prx_env)
        hipLaunchKernelGGL(ct_step_dyn_kernel<true>, dim3(grid), dim3(blk), 0, (hipStream_t)stream, st, cst, device, duration, obs, reward, done);
    else
        hipLaunchKernelGGL(ct_step_dyn_kernel<false>, dim3(grid), dim3(blk), 0, (hipStream_t)stream, st, cst, device, duration, obs, reward, done);
    return ok_or_ehip();
}

int gw_launch_init_dyn(const GwState& st, const GwDevConst& cst, double thermal, void* stream)
{
    hipLaunchKernelGGL(ct_init_dyn_kernel, dim3((unsigned)((st.N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, st, cst, thermal);
    return ok_or_ehip();
}

int gw_launch_set_position(const GwState& st, const GwDevConst& cst, int radio, const double* xs, const double* ys,
                           const double* all_pos, const uint8_t* mask, void* stream)
{
    hipLaunchKernelGGL(ct_set_position_kernel, dim3((unsigned)((st.N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       st, cst, radio, xs, ys, all_pos, mask);
    return ok_or_ehip();
}
