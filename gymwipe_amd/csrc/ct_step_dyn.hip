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
//
// The step is the default kernel's walk with A.2 / A.4 done in f64, in two parts:
//   WALK, one lane per env: announcement, window, counter ticks, interpreter -- serial per env.  The two receptions that
//     decide anything (the addressed sender hears the RRM, the RRM hears the sender's data) take their BER from a per-link
//     cache keyed by the noise power (GwState::bcache): BpskMcs.calculateBitErrorRate (physical.py:25-58,208-212) is
//     evaluated only when a link meets a noise value it has not just seen.
//   ALL-PAIRS, every OTHER radio j heard the announcement and, if any, d's n data packets (simple_stack.py:99-157):
//     rx_j <- (rx_j + p(RRM, j)) - p(RRM, j), then n times (rx_j + p(d, j)) - p(d, j) (a fixed point ends it).
//       D <= 6: the lane that walked the env does it, from its env's rows held in registers (16-byte loads);
//       D = 8, 16, 32: the wave re-maps its 64 lanes to groups of D lanes, one group per env and one lane per radio, D passes
//         over the wave's 64 envs: every load and store covers the contiguous rows of 64 / D envs, the talker and packet
//         count of the group's env come by wave shuffle from the lane that walked it, and the listeners' "the reference
//         would raise" bits are OR-reduced across the group (__shfl_xor butterfly) into one atomic per env.
//     With per-env geometry this part is what moves the bytes (D = 16: three 144-byte rows per env and step); thread-per-env
//     with a serial loop over listeners read them as 64 scattered 8-byte accesses per instruction.
// Device libm where the reference calls libm (log10, 10**x, e**x, sqrt): last-ulp differences from CPython's only enter
// error sums that are rounded to integers, so decisions agree except on a measure-zero boundary; with the host's link
// tables (no per-env geometry) the received powers themselves are exact sums and compare bit for bit.
#include "ct_common.hip.h"
#include "gw_queue.h"

using namespace gwk;

namespace {

// (through the GLOBAL address space: the pointers of this kernel's state come from the header in memory, and an access through
//  a pointer read from memory is a FLAT instruction unless told otherwise -- ct_common.hip.h, hdr_state)
template <class T>
__device__ __forceinline__ T ld(const void* base, size_t byte_off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(1))) T*)(reinterpret_cast<const char*>(base) + byte_off);
#else
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);       // (host pass: never executed)
#endif
}
template <class T>
__device__ __forceinline__ void st_(void* base, size_t byte_off, const T& v)
{
    gwk::gw_store_wt(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off), v);
}
__device__ __forceinline__ uint32_t word_of(const uint4& w, int i)          // i compile-time after unrolling
{
    return i == 0 ? w.x : (i == 1 ? w.y : (i == 2 ? w.z : w.w));
}
__device__ __forceinline__ double half_of(const double2& v, int i) { return i ? v.y : v.x; }

// physical.py:25-58 (Eb/N0, Q approximation), :82-98 (dBm helpers), :208-212 (BPSK).  exp10 / exp instead of pow(10, .) /
// pow(e, .): the same function values up to the device libm's last ulp (see the header), a third of the instructions.
__device__ __forceinline__ double ber_live(double sig_mw, double noise_mw, double ten_log_br)
{
    const double s = 10 * log10(sig_mw);
    const double n = 10 * log10(noise_mw);
    if (s <= n) return 0.5;
    const double ratio = exp10((s - n - ten_log_br) / 10);
    const double x = sqrt(2 * ratio);
    const double sqrt2pi = 2.5066282746310002;
    return (1 - exp(-1.4 * x)) * exp(-((x * x) / 2)) / (1.135 * sqrt2pi * x);
}

// BER of a link at a noise power, through the link's one-entry cache {noise, ber}
__device__ __forceinline__ double ber_cached(double2& entry, bool& dirty, double sig, double noise, double ten_log_br)
{
#ifndef GW_EXP_NO_BCACHE
    if (noise == entry.x) return entry.y;
#endif
    entry.x = noise;
    entry.y = ber_live(sig, noise, ten_log_br);
    dirty = true;
    return entry.y;
}

// one listener's received power after the announcement and n data packets of d (simple_stack.py:130-157)
__device__ __forceinline__ double heard(double a0, double pa, double pd, int n)
{
    double a = (a0 + pa) + (-pa);
    for (int i = 0; i < n; ++i) {
        const double b = (a + pd) + (-pd);
        if (b == a) break;                                       // a fixed point of the (+p, -p) pair stays one
        a = b;
    }
    return a;
}

// DT > 0: compile-time sender count, the qb record and (DT <= 6) the env's PHY rows in registers; DT == 0: any count.
template <int DT, bool PER_ENV>
__global__ __launch_bounds__(64) void ct_step_live_kernel(GW_LEAD_PARAMS, int32_t* __restrict__ obs, float* __restrict__ reward,
                                                         uint8_t* __restrict__ done)
{
    const int n_dev = (int)(dev_stage & 0xffu);
    const GwState st = hdr_state<DT>(ip, tw, tk, qb, n_envs, n_dev);
    const GwDevConst c = hdr_const<DT>(ip, n_dev);
    // (the arrays of the constants are indexed by the lane's action: through a pointer, as memory reads -- indexing the by-value
    //  copy would put it on the stack)
    const GwDevConst* cp = reinterpret_cast<const GwDevConst*>(reinterpret_cast<const uint8_t*>(ip) - gw_blob_header(DT > 0 ? DT : n_dev) +
                                                               gw_hdr_cst_off(DT > 0 ? DT : n_dev));
    constexpr bool PACKED = DT > 0;
    constexpr bool ROWS = DT > 0 && DT <= 6;                     // the env's rows in the walking lane's registers
    constexpr bool COOP = DT >= 8;                               // all-pairs part by groups of DT lanes
    constexpr bool RXR = DT >= 16;                               // the RRM's received power from its mirror (GwState::rxr)
    const int D = DT > 0 ? DT : c.D;
    const int R = D + 1, RRM = D;
    constexpr int NWC = DT > 0 ? (2 * DT + 1 + 15) / 16 : 1;
    constexpr int RPC = DT > 0 ? gw_rp(DT + 1) : 2;              // row pitch at compile time (gw_internal.h: whole memory lines)
    const int RP = DT > 0 ? RPC : gw_rp(R);
    constexpr int NH = DT > 0 ? ((DT + 2) & ~1) / 2 : 1;         // double2 chunks of a row that hold radios
    const uint32_t N = n_envs;
    const uint32_t e = blockIdx.x * 64u + threadIdx.x;
    const bool live = e < N;
    const uint32_t el = live ? e : 0u;
    const uint32_t RB = PACKED ? 16u * NWC : (uint32_t)st.RB;
    const size_t o16 = (size_t)el << 4, oq = (size_t)el * RB;
    const size_t orx = (size_t)el * RP * 8u;                     // the env's row of received powers
    const size_t olk = (size_t)el * R * RP * 8u;                 // the env's link matrix (PER_ENV)

    STAMP(0);
    // shared geometry: the handle's [R][R] link table -> LDS once per workgroup
    __shared__ double s_prx[PER_ENV ? 1 : (GW_MAX_RADIOS * GW_MAX_RADIOS)];
    if (!PER_ENV) {
        for (int i = threadIdx.x; i < R * R; i += 64) s_prx[i] = st.prx_tab[i];
    }
    // {multiplicity, ceil(65536 / multiplicity)} per sender: looked up by the lane's action right after it arrives -- from LDS,
    // not by a second global round trip
    __shared__ uint2 s_mi[DT > 0 ? DT : GW_MAX_DEVICES];
    for (int i = threadIdx.x; i < D; i += 64) s_mi[i] = make_uint2((uint32_t)cp->mult[i], cp->inv16[i]);

    // ---- loads, all issued before anything is waited for ---------------------------------------------------------------
    int d = device[el];
    int du = duration[el];
    uint4 bp = ld<uint4>(st.ip, o16);
    const double2 tw0 = ld<double2>(st.tw, o16);
    const uint4 tk0 = ld<uint4>(st.tk, o16);
    uint4 qw[NWC];
#pragma unroll
    for (int w = 0; w < NWC; ++w) qw[w] = PACKED ? ld<uint4>(st.qb, oq + 16u * w) : make_uint4(0u, 0u, 0u, 0u);
    double2 rxv[ROWS ? NH : 1], pav[ROWS && PER_ENV ? NH : 1], cav[ROWS ? DT : 1];
    if (ROWS) {
#pragma unroll
        for (int h = 0; h < NH; ++h) rxv[h] = ld<double2>(st.rxp, orx + 16u * h);
        // all D "sender i hears the RRM" cache entries: which one the step needs depends on the action, and a load issued
        // only once the action has arrived is a second memory round trip in front of the announcement's decision
#pragma unroll
        for (int i = 0; i < DT; ++i) cav[i] = ld<double2>(st.bcache, ((size_t)el * 2 * DT + 2 * i) * 16u);
        if (PER_ENV) {
#pragma unroll
            for (int h = 0; h < NH; ++h) pav[h] = ld<double2>(st.prx_env, olk + ((size_t)RRM * RP) * 8u + 16u * h);
        }
    }
    asm volatile("" : "+v"(d), "+v"(du));
    asm volatile("" : "+v"(bp.x), "+v"(bp.y), "+v"(bp.z), "+v"(bp.w));
    __syncthreads();                                             // s_prx, s_mi
    STAMP(1);

    const bool bad = (unsigned)d >= (unsigned)D || (unsigned)du >= (unsigned)c.max_duration;
    const int dq = bad ? 0 : d;                                   // a valid index for the dependent loads below
    // dependent on the action: the two cache entries, the talker's row (its link to every listener) and, unless the rows
    // are in registers already, the four scalars of the two deciding receptions
    double2 ca;
    if (ROWS) {
        ca = cav[0];
#pragma unroll
        for (int i = 1; i < DT; ++i) { ca.x = (i == dq) ? cav[i].x : ca.x; ca.y = (i == dq) ? cav[i].y : ca.y; }
    } else {
        ca = ld<double2>(st.bcache, ((size_t)el * 2 * D + 2 * dq) * 16u);
    }
    double2 cx = ld<double2>(st.bcache, ((size_t)el * 2 * D + 2 * dq + 1) * 16u);   // (first needed at the first data packet)
    double2 pdv[ROWS && PER_ENV ? NH : 1];
    if (ROWS && PER_ENV) {
#pragma unroll
        for (int h = 0; h < NH; ++h) pdv[h] = ld<double2>(st.prx_env, olk + ((size_t)dq * RP) * 8u + 16u * h);
    }
    auto link = [&](int from, int to) -> double {
        return PER_ENV ? ld<double>(st.prx_env, (((size_t)el * R + from) * RP + to) * 8u) : s_prx[from * R + to];
    };
    double rx_d0, rx_r0, p_a, p_x;
    if (ROWS) {
        rx_d0 = 0.0; rx_r0 = half_of(rxv[RRM >> 1], RRM & 1);
#pragma unroll
        for (int j = 0; j < DT; ++j) rx_d0 = (j == dq) ? half_of(rxv[j >> 1], j & 1) : rx_d0;
        if (PER_ENV) {
            p_a = 0.0;
#pragma unroll
            for (int j = 0; j < DT; ++j) p_a = (j == dq) ? half_of(pav[j >> 1], j & 1) : p_a;
        } else {
            p_a = link(RRM, dq);
        }
    } else {
        rx_d0 = ld<double>(st.rxp, ((size_t)el * RP + dq) * 8u);
        rx_r0 = RXR ? ld<double>(st.rxr, (size_t)el * 8u) : ld<double>(st.rxp, ((size_t)el * RP + RRM) * 8u);
        p_a = link(RRM, dq);
    }
    // the attenuation of a pair is one number (one model per unordered pair, physical.py:500-528; gw_create refuses an
    // asymmetric extra_att_db) and every radio sends with the same power: the link matrix is symmetric bit for bit, so the
    // RRM hears d with the power d hears the RRM with -- no load that waits for the action
    p_x = p_a;

    // ---- all-pairs, D = 8 / 16 / 32, first half: the listeners' loads, NOW.  Groups of D lanes, one lane per listening radio, one
    // env per group and pass; what a lane reads (its radio's received power and its links to the RRM and to the env's talker)
    // depends on the env's action only, so every pass's loads are issued here, in front of the walk, and their round trip to
    // memory (in-kernel stamps at D = 16: ~4 000 cycles each for the four batches the pass used to run in) passes while the
    // walker walks.  The second half -- the (+p, -p) updates, which need the walk's packet count -- follows the walk.
    constexpr int CG = COOP ? DT : 1;                             // lanes per env
    constexpr int CEPP = 64 / CG;                                 // envs per pass
    constexpr int CNP = COOP ? CG : 1;                            // passes
    const uint32_t c_lane = threadIdx.x, c_j = c_lane % CG;       // this lane's radio
    const uint32_t c_act = (live && !bad) ? 1u : 0u;
    const uint32_t e_wave = blockIdx.x * 64u;                     // first env of the wave (always < N)
    double c_a0[CNP], c_pa[CNP], c_pd[CNP];
    uint32_t c_off[CNP];
    uint64_t c_on = 0ull;                                         // bit u = pass u has a listener for this lane
    if (COOP) {
        const uint32_t info0 = (uint32_t)dq | (c_act << 31);
        const double* rx_wave = st.rxp + (size_t)e_wave * RP;
        const double* lk_wave = PER_ENV ? st.prx_env + (size_t)e_wave * R * RP : nullptr;
#pragma unroll
        for (int u = 0; u < CNP; ++u) {
            const uint32_t src = (uint32_t)(u * CEPP) + c_lane / CG;          // the lane that walks this group's env
            const uint32_t w = (uint32_t)__shfl((int)info0, (int)src);
            const uint32_t dsrc = w & 0xffu;
            const bool on = (w >> 31) != 0u && c_j != dsrc;
            c_on |= on ? (1ull << u) : 0ull;
            const uint32_t es = on ? src : 0u;                    // idle lanes read the wave's first env: a valid row
            c_off[u] = es * (uint32_t)RP + c_j;
            c_a0[u] = ld<double>(rx_wave, (size_t)c_off[u] * 8u);
            c_pa[u] = PER_ENV ? ld<double>(lk_wave, (size_t)((es * (uint32_t)R + (uint32_t)RRM) * (uint32_t)RP + c_j) * 8u) : s_prx[RRM * R + c_j];
            const uint32_t dr = on ? dsrc : 0u;
            c_pd[u] = PER_ENV ? ld<double>(lk_wave, (size_t)((es * (uint32_t)R + dr) * (uint32_t)RP + c_j) * 8u) : s_prx[dr * R + c_j];
        }
    }

    const StepMath m(c);
    uint32_t rvm = tk0.z;                                         // (record layout: ct_step_sfx.hip)
    int32_t last_abs = (int32_t)(tk0.w & 0x7fffffffu);
    uint32_t dn = tk0.w >> 31;
    uint32_t fl = 0, k_bad = 0, k_pop = 0, k_deliv = 0;
    const int pv = c.payload_value;
    int n_data = 0;                                               // data packets of d this step (0 for a bad action / dead lane)
    double rx_d = rx_d0, rx_r = rx_r0;

    // COOP: what the walk leaves behind is held in registers and stored BEHIND the lane groups' all-pairs pass -- vmcnt counts
    // loads and stores in one order, so the groups' loads, issued behind the walk's write-through stores, each sat out those
    // stores' round trip to memory before their own data counted as arrived
    bool out_commit = false, ca_dirty = false, cx_dirty = false;
    uint4 out_qb[NWC];
    int32_t out_obs = 0;
    float out_rew = 0.0f;
    double2 out_tw = make_double2(0.0, 0.0);
    uint4 out_tk = make_uint4(0u, 0u, 0u, 0u);

    if (live && bad) {
        // counter_traffic.py:147 asserts; a batched step cannot raise per env: flag + skip
        fl = GW_FLAG_BADACT;
        k_bad = 1;
        const int32_t latest = pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u));
        obs[e] = latest + c.counter_bound;
        reward[e] = 0.0f;
        done[e] = (uint8_t)dn;
    } else if (live) {
        const double slot = c.slot, br = c.bit_rate, hd = c.hdr_dur, hdr_bits = c.hdr_bits, interval = c.counter_interval;
        const double ten_log_br = c.ten_log_br, coded_factor = c.coded_factor;
        const uint32_t bound = (uint32_t)c.counter_bound, base_bytes = (uint32_t)(c.mac_hdr + c.net_hdr);
        const int mh = c.mac_hdr;

        uint32_t nb[PACKED ? 16 * NWC : 1];
        uint32_t len_d = 0;
        if (PACKED) {
#pragma unroll
            for (int b = 0; b < 16 * NWC; ++b) nb[b] = (word_of(qw[b >> 4], (b >> 2) & 3) >> ((b & 3) * 8)) & 0xffu;
#pragma unroll
            for (int i = 0; i < DT; ++i) len_d = (i == d) ? nb[i] : len_d;
        } else {
            len_d = st.qb[oq + (uint32_t)d];
        }
        const uint2 mi_d = s_mi[d];
        const uint32_t mult_d = mi_d.x, inv16_d = mi_d.y;
        const double t_a = tw0.x;
        double wake = tw0.y;
        const uint32_t tau0 = tk0.x, nbp = tk0.y;
        GwBp bpc, bpp;
        bpc.t0 = bp.x; bpc.c0 = bp.y;
        bpp.t0 = bp.z; bpp.c0 = bp.w;
        const GwBp* hist = st.bph + ((size_t)e << 7);
        const int slots = du * c.duration_factor;                         // counter_traffic.py:149

        STAMP(2);
        // ---- A.1 / A.2: announcement, heard by the addressed sender ------------------------------------
        const int Ld = ndigits(slots);
        const TxTimes an = tx_times(m, t_a, hd, m.over_rate((double)(Ld * 8)));
        const double up_d = rx_d0 + p_a;                                  // simple_stack.py:82  (+p) at the start
        const double noise_d = up_d - p_a;                                // :166-167 noise = received - signal
        if (!(noise_d >= 0.0)) fl |= GW_FLAG_REFEXC;                      // :168 assert noisePower >= 0
        const double ber_a = ber_cached(ca, ca_dirty, p_a, noise_d, ten_log_br);
        const bool granted = receive(m, ber_a, an, br, hdr_bits, (double)(Ld * 8) * coded_factor, fl);
        if (!(an.t_e >= an.stop)) fl |= GW_FLAG_REFEXC;
        rx_d = up_d + (-p_a);                                             // :154 (-p) when it completes
        const double t_r = an.t_e;
        const double t_end = t_r + (double)(slots + 1) * slot;            // simple_stack.py:557-558

        // ---- counter ticks: one jump per call where the step qualifies (gw_fastmath.h), else the running-sum loop -----
        uint32_t tau = tau0;
        GwTally kd = {0, 0, 0, 0, 0};
        auto ticks_to = [&](double t, bool inclusive) {
            uint32_t kk = 0;
            for (;;) {
                const double w1 = wake + interval, w2 = w1 + interval, w3 = w2 + interval, w4 = w3 + interval;
                const bool b0 = inclusive ? (wake <= t) : (wake < t);
                const bool b1 = inclusive ? (w1 <= t) : (w1 < t);
                const bool b2 = inclusive ? (w2 <= t) : (w2 < t);
                const bool b3 = inclusive ? (w3 <= t) : (w3 < t);
                const double last = b3 ? w3 : (b2 ? w2 : (b1 ? w1 : wake));
                if (inclusive && b0 && last == t) fl |= GW_FLAG_TIE;
                kk += (uint32_t)b0 + (uint32_t)b1 + (uint32_t)b2 + (uint32_t)b3;
                wake = b3 ? w4 : (b2 ? w3 : (b1 ? w2 : (b0 ? w1 : wake)));
                if (!b3) break;
            }
            tau += kk;
            len_d = gw_len_after_ticks(len_d, kk, mult_d, kd);
        };
        double delta = 0.0;
        const bool span_ok = c.fast_ticks && gw_tick_span_ok(wake, t_end, interval, &delta);
        auto ticks_upto = [&](double t, bool inclusive) {
            uint32_t nj = 0;
            double wj = wake;
            bool tiej = false, sane = false;
            gw_tick_jump_lo(wake, t, delta, c.inv_interval_lo, inclusive, &nj, &wj, &tiej, &sane);
            if (span_ok && sane) {
                wake = wj;
                tau += nj;
                if (tiej) fl |= GW_FLAG_TIE;
                len_d = gw_len_after_ticks(len_d, nj, mult_d, kd);
            } else {
                ticks_to(t, inclusive);
            }
        };

        STAMP(3);
        // ---- A.3 / A.4: window at sender d, every data packet heard by the RRM -----------------------------------
        if (granted) {
            const double total = (double)slots * slot;                    // simple_stack.py:400
            const double stopw = t_r + total;                             // :401
            double cur = t_r;
            ticks_upto(cur, false);                                       // the MAC's process initialisation is URGENT
            // ---- the window loop, straight-line form (the default kernel's, ct_step_sfx.hip: a lone wave per SIMD issues one
            //      instruction per ~8 cycles, and the wave lasts as long as its busiest lane's 8-9 packets; in-kernel stamps put
            //      the general loop below at 1 900 + 1 700 cycles per packet of that lane).  Taken when tick jumps apply, the exact
            //      fast forms of fmod and division hold up to t_end and t_r >= 2 (t_end - t_r): then a packet's header end and
            //      stop time ARE t_s + hd and t_s + (hd + pd) (Sterbenz), the completion event fires at stop, and the reference's
            //      `not t.completed` case cannot arise.  The live PHY's part stays per packet: the RRM's (+p, -p) residue and the
            //      BER at that noise (through the link's cache) decide every reception.  One exit condition; the general loop
            //      takes over on an empty queue (it waits for the tick) or a breakpoint-ring lookup.
            bool more = true;
            {
                const double span = t_end - t_r;
                const bool straight = span_ok && mult_d != 0u && m.fast_fmod && m.fast_div && t_end < m.fmod_limit && t_r >= span + span;
                if (straight && len_d != 0u) {
                    auto head = [&](uint32_t len, uint32_t tk_now, bool& deep) {
                        const uint32_t age = __umul24(len + mult_d - 1u, inv16_d) >> 16;      // gw_ceil_div
                        const uint32_t ht = tk_now - age;                 // tick of the head packet
                        const bool older = ht < bpc.t0;
                        deep = older && ht < bpp.t0;                      // > 2 resets inside the queue's span
                        return base_bytes + gw_min_u32((older ? bpp.c0 : bpc.c0) + (ht - (older ? bpp.t0 : bpc.t0)), bound);
                    };
                    bool deep = false;
                    uint32_t chk = 0, pops = 0;
                    uint32_t s = head(len_d, tau, deep);
                    bool go = !deep && (stopw - cur) > gw_fast_div((double)(s * 8u), m.dr, m.rcp_dr);       // :418-420
                    while (go) {
                        const int pay = (int)s - mh;
                        const double pd = gw_fast_div((double)(pay * 8), m.dr, m.rcp_dr);
                        TxTimes x;
                        x.t_s = cur + (m.slot - gw_fast_fmod_lo(cur, m.slot, c.inv_slot_lo));
                        x.t_h = x.t_s + hd;
                        x.stop = x.t_s + (hd + pd);
                        x.t_e = x.stop;
                        const double up = rx_r + p_x;
                        const double noise = up - p_x;
                        chk |= !(noise >= 0.0) ? (uint32_t)GW_FLAG_REFEXC : 0u;
                        const double ber_x = ber_cached(cx, cx_dirty, p_x, noise, ten_log_br);
                        uint32_t unused = 0;
                        const bool ok = receive(m, ber_x, x, br, hdr_bits, (double)(pay * 8) * coded_factor, unused);
                        rx_r = up + (-p_x);
                        k_deliv += ok ? 1u : 0u;                          // devices.py:163-168, counter_traffic.py:75-80
                        rvm |= ok ? (1u << d) : 0u;
                        dn = (ok && pv == c.counter_bound) ? 1u : dn;
                        uint32_t nj = 0;
                        double wj = wake;
                        bool tiej = false, sane = false;
                        gw_tick_jump_lo(wake, x.t_e, delta, c.inv_interval_lo, true, &nj, &wj, &tiej, &sane);
                        chk |= (sane ? 0u : (uint32_t)GW_FLAG_INTERNAL) | (tiej ? (uint32_t)GW_FLAG_TIE : 0u);
                        len_d = gw_min_u32(len_d - 1u + __umul24(nj, mult_d), (uint32_t)GW_QUEUE_CAP);
                        tau += nj;
                        wake = wj;
                        cur = x.t_e;
                        pops++;
                        bool deep_n = false;
                        s = head(len_d, tau, deep_n);
                        go = cur < stopw && len_d != 0u && !deep_n && (stopw - cur) > gw_fast_div((double)(s * 8u), m.dr, m.rcp_dr);
                    }
                    (void)head(len_d, tau, deep);
                    more = cur < stopw && (len_d == 0u || deep);
                    fl |= chk | ((pops && !(cur < t_end)) ? (uint32_t)GW_FLAG_CARRY : 0u);
                    k_pop += pops;
                    n_data += (int)pops;
                }
            }
            if (more)
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
                k_pop++;
                const int pay = (int)s - mh;
                const TxTimes x = tx_times(m, cur, hd, m.over_rate((double)(pay * 8)));
                n_data++;
                const double up = rx_r + p_x;
                const double noise = up - p_x;
                if (!(noise >= 0.0)) fl |= GW_FLAG_REFEXC;
                const double ber_x = ber_cached(cx, cx_dirty, p_x, noise, ten_log_br);
                if (!(x.t_e >= x.stop)) fl |= GW_FLAG_REFEXC;
                const bool ok = receive(m, ber_x, x, br, hdr_bits, (double)(pay * 8) * coded_factor, fl);
                rx_r = up + (-p_x);
                k_deliv += ok ? 1u : 0u;                                  // devices.py:163-168, counter_traffic.py:75-80
                rvm |= ok ? (1u << d) : 0u;
                dn = (ok && pv == c.counter_bound) ? 1u : dn;
                fl |= !(x.t_e < t_end) ? (uint32_t)GW_FLAG_CARRY : 0u;
                ticks_upto(x.t_e, true);
                cur = x.t_e;
                if (!(cur < stopw)) break;
            }
        }
        STAMP(4);
        // ---- A.5: remaining ticks up to the end of the step ------------------------------------------------
        ticks_upto(t_end, true);
        const uint32_t n_ticks = tau - tau0;

        // ---- queue lengths of every sender (all tick together) -----------------------------------------------
        if (PACKED) {
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                GwTally ki = {0, 0, 0, 0, 0};
                const uint32_t li = gw_len_after_ticks(nb[i], n_ticks, (uint32_t)c.mult[i], ki);   // (i compile-time)
                nb[i] = (i == d) ? len_d : li;
                // bytes D .. 2D of the record (the default kernel's noise states, unused here): bit 0 = radio i has transmitted,
                // i.e. its attenuation models exist (Position.set's keep-stale rules ask, ct_set_position_kernel)
                nb[DT + i] |= (i == d && n_data) ? 1u : 0u;
            }
            nb[2 * DT] |= 1u;                                             // the RRM has (the announcement)
#pragma unroll
            for (int w = 0; w < NWC; ++w) {
                const int b = 16 * w;
                uint4 o;
                o.x = nb[b + 0] | (nb[b + 1] << 8) | (nb[b + 2] << 16) | (nb[b + 3] << 24);
                o.y = nb[b + 4] | (nb[b + 5] << 8) | (nb[b + 6] << 16) | (nb[b + 7] << 24);
                o.z = nb[b + 8] | (nb[b + 9] << 8) | (nb[b + 10] << 16) | (nb[b + 11] << 24);
                o.w = nb[b + 12] | (nb[b + 13] << 8) | (nb[b + 14] << 16) | (nb[b + 15] << 24);
                if (COOP) out_qb[w] = o;
                else st_(st.qb, oq + 16u * w, o);
            }
        } else {
            GwTally ki = {0, 0, 0, 0, 0};
            for (int i = 0; i < D; ++i)
                if (i != d) st.qb[oq + (uint32_t)i] = (uint8_t)gw_len_after_ticks(st.qb[oq + (uint32_t)i], n_ticks, (uint32_t)cp->mult[i], ki);
            st.qb[oq + (uint32_t)d] = (uint8_t)len_d;
            if (n_data) st.qb[oq + (uint32_t)(D + d)] = 1;                // (talk bits, see above)
            st.qb[oq + (uint32_t)(2 * D)] = 1;
        }

        STAMP(5);
        // ---- all-pairs, D <= 6 (rows in registers) and the any-D path (in memory) -------------------------------
        if (ROWS) {
            double rn[2 * NH];
#pragma unroll
            for (int j = 0; j < 2 * NH; ++j) rn[j] = half_of(rxv[j >> 1], j & 1);
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                const double pa = PER_ENV ? half_of(pav[j >> 1], j & 1) : s_prx[RRM * R + j];
                const double pd = PER_ENV ? half_of(pdv[j >> 1], j & 1) : s_prx[d * R + j];
                const double a = heard(rn[j], pa, pd, n_data);
                if (j != d && !(a >= 0.0)) fl |= GW_FLAG_REFEXC;
                rn[j] = (j == d) ? rx_d : a;
            }
            rn[RRM] = rx_r;
            // (the residues settle on fixed points of the (+p, -p) pairs: most steps change nothing)
#pragma unroll
            for (int h = 0; h < NH; ++h)
                if (rn[2 * h] != rxv[h].x || rn[2 * h + 1] != rxv[h].y) st_(st.rxp, orx + 16u * h, make_double2(rn[2 * h], rn[2 * h + 1]));
        } else {
            if (!COOP) {
                if (rx_d != rx_d0) st_(st.rxp, ((size_t)e * RP + d) * 8u, rx_d);
                if (rx_r != rx_r0) st_(st.rxp, ((size_t)e * RP + RRM) * 8u, rx_r);
            }
            if (!COOP) {
                for (int j = 0; j < D; ++j) {
                    if (j == d) continue;
                    const double a0 = ld<double>(st.rxp, ((size_t)e * RP + j) * 8u);
                    const double a = heard(a0, link(RRM, j), link(d, j), n_data);
                    if (!(a >= 0.0)) fl |= GW_FLAG_REFEXC;
                    if (a != a0) st_(st.rxp, ((size_t)e * RP + j) * 8u, a);
                }
            }
        }
        if (!COOP) {
            if (ca_dirty) st_(st.bcache, ((size_t)e * 2 * D + 2 * d) * 16u, ca);
            if (cx_dirty) st_(st.bcache, ((size_t)e * 2 * D + 2 * d + 1) * 16u, cx);
        }

        // ---- interpreter feedback (counter_traffic.py:85-112, envs/core.py:142-153) -----------------------
        const int32_t latest = pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u));
        const int32_t abs_d = latest < 0 ? -latest : latest;
        int32_t r = last_abs - abs_d;
        last_abs = abs_d;
        r = r > 10 ? 10 : (r < -10 ? -10 : r);
        if (COOP) {
            out_commit = true;
            out_obs = (int32_t)(latest + c.counter_bound);
            out_rew = (float)r;
            out_tw = make_double2(t_end, wake);
            out_tk = make_uint4(tau, nbp, rvm, (uint32_t)last_abs | (dn << 31));
        } else {
            st_(obs, (size_t)e << 2, (int32_t)(latest + c.counter_bound));
            st_(reward, (size_t)e << 2, (float)r);
            st_(done, (size_t)e, (uint8_t)dn);
            st_(st.tw, o16, make_double2(t_end, wake));
            st_(st.tk, o16, make_uint4(tau, nbp, rvm, (uint32_t)last_abs | (dn << 31)));
        }
    }
    STAMP(6);
    if (!COOP && live) publish_env_counters(st.sa, N, e, k_pop, k_deliv, k_bad, fl, 1u);

    // ---- all-pairs, D = 8 / 16 / 32, second half: the listeners' updates (loads issued in front of the walk) ------------------
    if (COOP) {
        // A lone wave issues an instruction every ~8 cycles, so this is written for instruction count: ONE shuffle per env (the
        // walker's packet count), 32-bit offsets from the wave's base address, and the "reference would raise" bit -- never seen in
        // practice -- straight to its env's flag word by whichever listener lane found it.
        double* rx_wave = st.rxp + (size_t)e_wave * RP;
#pragma unroll
        for (int u = 0; u < CNP; ++u) {
            const uint32_t src = (uint32_t)(u * CEPP) + c_lane / CG;
            const int nsrc = __shfl(n_data, (int)src);
            if ((c_on >> u) & 1ull) {
                const double a = heard(c_a0[u], c_pa[u], c_pd[u], nsrc);
                if (a != c_a0[u]) st_(rx_wave, (size_t)c_off[u] * 8u, a);
                if (!(a >= 0.0))
                    __hip_atomic_fetch_or(st.sa + (size_t)3 * N + e_wave + src, (uint32_t)GW_FLAG_REFEXC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        STAMP(7);
        // ---- the walking lane's own stores, last (its rx[d], rx[RRM] are other words than the listeners' above) ----
        if (out_commit) {
            if (rx_d != rx_d0) st_(st.rxp, ((size_t)e * RP + d) * 8u, rx_d);
            if (rx_r != rx_r0) {
                st_(st.rxp, ((size_t)e * RP + RRM) * 8u, rx_r);
                if (RXR) st_(st.rxr, (size_t)e * 8u, rx_r);
            }
            if (ca_dirty) st_(st.bcache, ((size_t)e * 2 * D + 2 * d) * 16u, ca);
            if (cx_dirty) st_(st.bcache, ((size_t)e * 2 * D + 2 * d + 1) * 16u, cx);
#pragma unroll
            for (int w = 0; w < NWC; ++w) st_(st.qb, oq + 16u * w, out_qb[w]);
            st_(obs, (size_t)e << 2, out_obs);
            st_(reward, (size_t)e << 2, out_rew);
            st_(done, (size_t)e, (uint8_t)dn);
            st_(st.tw, o16, out_tw);
            st_(st.tk, o16, out_tk);
        }
        if (live) publish_env_counters(st.sa, N, e, k_pop, k_deliv, k_bad, fl, 1u);
        STAMP(8);
    }
}

// FsplAttenuation._update + dbmToMilliwatts with the device libm (devices/core.py:88-95, attenuation_models.py:28-36,
// simple_stack.py:111); a model created for a pair on one spot starts (and stays) at 0 dB
__device__ __forceinline__ double link_power(double dist, double extra_db, double tx_dbm, double twenty_log_f, bool same)
{
    double att = same ? 0.0 : 20 * log10(dist) + twenty_log_f - 147.55;
    if (extra_db != 0.0) att = (0.0 + att) + extra_db;                    // joined model: sum([fspl, custom])
    return pow(10.0, (tx_dbm - att) / 10);
}

// Position.set on radio `radio` (or, radio < 0, on every radio in index order: successive Position.set calls) of the envs
// selected by mask.  Each set notifies the attenuation models of the moved radio's pairs (devices/core.py:77-86 ->
// physical.py:380-386), which -- as in the reference --
//   * do not update while the pair is >= STANDBY_THRESHOLD = 3000 m apart (physical.py:371-386: the stale value stays),
//   * do not update when the pair now shares one spot (FsplAttenuation._update returns early, attenuation_models.py:31-33),
//   * do not exist before one of the two radios has transmitted (models are created at first use, physical.py:500-528;
//     GwState::talk): such a pair has nothing to keep and takes the positions of the moment.
// Nothing is on the air between two env.step() calls, so no reception is re-integrated (simple_stack.py:119-128).
__global__ void ct_set_position_kernel(GwState st, GwDevConst c, int radio, const double* __restrict__ xs,
                                       const double* __restrict__ ys, const double* __restrict__ all_pos,
                                       const uint8_t* __restrict__ mask)
{
    const uint32_t N = (uint32_t)st.N;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    if (mask && !mask[e]) return;
    const int D = c.D, R = D + 1, RP = gw_rp(R);
    double* pos = st.pos_env + (size_t)e * R * 2;
    double* prx = st.prx_env + (size_t)e * R * RP;
    uint64_t talk = 0;                                                    // bit r: radio r has transmitted
    if (st.talk) talk = st.talk[e];                                       // (explicit-queue mode keeps the mask in its own array)
    else for (int r = 0; r < R; ++r) talk |= (uint64_t)(st.qb[(size_t)e * st.RB + D + r] & 1u) << r;
    const int a_lo = radio >= 0 ? radio : 0, a_hi = radio >= 0 ? radio + 1 : R;
    bool moved_any = false;
    for (int a = a_lo; a < a_hi; ++a) {
        const double ax = radio >= 0 ? xs[e] : all_pos[((size_t)e * R + a) * 2 + 0];
        const double ay = radio >= 0 ? ys[e] : all_pos[((size_t)e * R + a) * 2 + 1];
        pos[a * 2 + 0] = ax;
        pos[a * 2 + 1] = ay;
        for (int b = 0; b < R; ++b) {
            if (b == a) continue;
            const double bx = pos[b * 2 + 0], by = pos[b * 2 + 1];
            const bool same = ax == bx && ay == by;
            const double dist = sqrt(pow(ax - bx, 2.0) + pow(ay - by, 2.0));
            const bool exists = (((talk >> a) | (talk >> b)) & 1ull) != 0;
            if (exists && (same || dist >= 3000.0)) continue;             // the stale attenuation stays
            const double p = link_power(dist, st.extra_tab[a * R + b], c.tx_power_dbm, c.twenty_log_f, same);
            prx[a * RP + b] = p;                                          // symmetric: attenuation depends on the pair only
            prx[b * RP + a] = p;
            moved_any = true;
        }
    }
    if (moved_any && st.bcache) {                                         // signal powers changed: the BER cache's keys no longer say enough
        const double nan = __longlong_as_double(0x7ff8000000000000ll);
        for (int i = 0; i < 2 * D; ++i) st.bcache[((size_t)e * 2 * D + i) * 2] = nan;
    }
}

// fresh handle in the live-PHY mode: every radio at thermal noise; per-env geometry starts as the handle's geometry
// with the host's link powers (bit-identical to the reference until a radio is moved)
__global__ void ct_init_dyn_kernel(GwState st, GwDevConst c, double thermal)
{
    const uint32_t N = (uint32_t)st.N;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int D = c.D, R = D + 1, RP = gw_rp(R);
    for (int r = 0; r < RP; ++r) st.rxp[(size_t)e * RP + r] = r < R ? thermal : 0.0;
    if (st.rxr) st.rxr[e] = thermal;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    if (st.bcache)
        for (int i = 0; i < 2 * D; ++i) { st.bcache[((size_t)e * 2 * D + i) * 2] = nan; st.bcache[((size_t)e * 2 * D + i) * 2 + 1] = 0.0; }
    if (st.talk) st.talk[e] = 0ull;
    if (st.prx_env) {
        for (int a = 0; a < R; ++a)
            for (int b = 0; b < RP; ++b) st.prx_env[((size_t)e * R + a) * RP + b] = b < R ? st.prx_tab[a * R + b] : 0.0;
        for (int i = 0; i < R * 2; ++i) st.pos_env[(size_t)e * R * 2 + i] = st.pos_tab[i];
    }
}

inline int ok_or_ehip() { return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP; }

template <int DT>
int launch_live(const GwState& st, const int32_t* device, const int32_t* duration, int32_t* obs, float* reward, uint8_t* done,
                hipStream_t stream)
{
    const unsigned grid = (unsigned)((st.N + 63) / 64);
    if (st.prx_env)
        hipLaunchKernelGGL((ct_step_live_kernel<DT, true>), dim3(grid), dim3(64), 0, stream, GW_LEAD_ARGS(st), obs, reward, done);
    else
        hipLaunchKernelGGL((ct_step_live_kernel<DT, false>), dim3(grid), dim3(64), 0, stream, GW_LEAD_ARGS(st), obs, reward, done);
    return ok_or_ehip();
}

} // namespace

int gw_launch_step_dyn(const GwState& st, const GwDevConst& cst, const int32_t* device, const int32_t* duration,
                       int32_t* obs, float* reward, uint8_t* done, void* stream)
{
    (void)cst;
    hipStream_t s = (hipStream_t)stream;
    switch (st.D) {
    case 2:  return launch_live<2>(st, device, duration, obs, reward, done, s);
    case 3:  return launch_live<3>(st, device, duration, obs, reward, done, s);
    case 4:  return launch_live<4>(st, device, duration, obs, reward, done, s);
    case 6:  return launch_live<6>(st, device, duration, obs, reward, done, s);
    case 8:  return launch_live<8>(st, device, duration, obs, reward, done, s);
    case 16: return launch_live<16>(st, device, duration, obs, reward, done, s);
    case 32: return launch_live<32>(st, device, duration, obs, reward, done, s);
    default: return launch_live<0>(st, device, duration, obs, reward, done, s);
    }
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
