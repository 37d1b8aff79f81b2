// ct_step.hip -- the GENERIC step kernel (GW_CFG_EXPLICIT_QUEUE): MAC queues as explicit rings of packet
// byte sizes, one per (env, sender), as the reference's deque holds them.  It makes no assumption about
// what the senders enqueue; the default kernel (ct_step_sfx.hip) exploits the structure of counter
// traffic instead and is ~12x faster.  Kept as the fallback for other traffic and as an independent
// second implementation the parity tests run against the same oracle.
//
// One launch advances all N environments by one env.step().  The reference walks
// ~90 SimPy events per step through Python objects (counter_traffic.py:146-158 ->
// simtools.py:77-88); here the event horizon of a step is enumerated directly per
// environment in f64, in the reference's exact operation order:
//
//   A.1  t_s = t_a + (slot - t_a % slot)                      simtools.py:44-53
//   A.2  announcement (13 B header + len(str(slots)) B payload); the addressed
//        sender decides header, then payload                  simple_stack.py:214-286,536-558
//   A.3  window at the addressed sender: pop + transmit while
//        (stop - now) > bits/dataRate                         simple_stack.py:397-434
//   A.4  the RRM decodes each data packet -> interpreter      networking/devices.py:163-168
//   A.5  t_end = t_r + (slots+1)*slot; counters tick every 1 ms (running f64 sum)
//        appending `mult` packets of 25+c bytes               counter_traffic.py:53-61
//   A.6  equal-time events: earlier-inserted first; process initialisation URGENT.
//
// No transcendental is evaluated on the device: link powers, BERs and the
// rx-power residue state machine come from host tables (gw_tables.cpp).
// Compile with -ffp-contract=off: every f64 result must be the IEEE result of the
// reference's individual operations.
#include "ct_common.hip.h"
#include "gw_runq.h"

using namespace gwk;

namespace {


// MAC queues: run-length deques, gw_runq.h -- one 16-byte record per (sender, env) holding the runs at both ends, the
// runs in between in a ring in HBM that a step of counter traffic does not touch.
__device__ __forceinline__ GwRunQ load_q(const GwRec& r) { return gw_runq_unpack(r); }
__device__ __forceinline__ GwRec store_q(const GwRunQ& q) { return gw_runq_pack(q); }

// DT > 0: compile-time sender count -- every sender's queue record is loaded up front into registers (the latency
// overlaps everything else); DT == 0: any sender count, loaded where needed.
// DYN: the live physical layer (f64 received power per radio in st.rxp, BER on the device, link powers shared or per env)
// instead of the noise-state bytes -- for layouts without a finite noise-state set; instantiated for DT == 0 only.
// SPLIT (DT > 0, table PHY): blocks of TWO waves over the same 64 envs.  Wave 0 walks the addressed sender's window (the
// step's critical path); wave 1, the helper, gives every OTHER sender's queue the step's counter ticks -- their number
// follows from the announcement's timing alone (all ticks up to t_end), not from the walk.  A lone wave per SIMD issues one
// instruction every ~8 cycles whatever it is (in-kernel stamps: the other senders' queues were 3 700 of a wave's 19 600
// cycles at D = 4), so taking instructions OFF the walker's stream is what shortens the launch; the helper's wave runs
// beside it on another SIMD of the CU.  The two waves write disjoint records (queue d / queues != d).
template <int DT, bool PER_ENV_STATS, bool DYN, bool SPLIT = false>
__global__ __launch_bounds__(256) void ct_step_kernel(GwState st,
                                                        const int32_t* __restrict__ device,
                                                        const int32_t* __restrict__ duration,
                                                        int32_t* __restrict__ obs,
                                                        float* __restrict__ reward,
                                                        uint8_t* __restrict__ done)
{
    static_assert(!SPLIT || (DT > 0 && !DYN), "the split form exists for compile-time sender counts on the table PHY");
    const int64_t N = st.N;
    const int64_t e = SPLIT ? (int64_t)blockIdx.x * 64 + (threadIdx.x & 63) : (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool helper = SPLIT && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 1;      // wave-uniform
    // the handle's constants BY VALUE through the constant address space (scalar loads the compiler may hoist and keep;
    // through the plain reference every use was a fresh global load); run-time indexed members go through the pointer
    const GwDevConst* cp = st.cst;
    const GwDevConst c = *(const GW_AS_CONST GwDevConst*)st.cst;
    const int D = DT > 0 ? DT : c.D;
    const int R = D + 1, S = GW_MAX_NSTATES, RRM = D;

    // the step tables the walk looks up per transmission (GwBlobLayout, built by gw_create) -> LDS: the window loop's
    // state -> BER chain is two dependent lookups per data packet, an L2 round trip each when read from global memory
    constexpr int DM = DT > 0 ? DT : GW_MAX_DEVICES;
    constexpr GwBlobLayout LM(DM);
    const GwBlobLayout L(D);
    __shared__ __attribute__((aligned(16))) uint8_t s_blob[LM.lds_total];
    for (int i = threadIdx.x; i < (L.lds_total >> 4); i += blockDim.x)
        *reinterpret_cast<uint4*>(s_blob + ((uint32_t)i << 4)) = *reinterpret_cast<const uint4*>(st.blob + ((uint32_t)i << 4));
    const double* s_ber = reinterpret_cast<const double*>(s_blob + L.ber);   // [0][d][s] d hears the RRM; [1][d][s] the RRM hears d
    const uint8_t* s_h1 = s_blob + L.h1;                                       // state of j after the announcement
    const uint8_t* s_rr = s_blob + L.r1;                                       // state of the RRM after a packet of d
    const uint8_t* g_h2 = st.blob + L.h2;                                      // j after the announcement and >= 1 packet of d (idem)
    const uint8_t* s_cls = s_blob + L.cls;

    STAMP(0);
    Tally k = {0, 0, 0, 0, 0};
    uint32_t k_bad = 0, k_steps = 0, fl_new = 0;

    // {multiplicity, ceil(2^20 / multiplicity), dest} per sender: looked up by the lane's action -- from LDS, not by a global
    // round trip that can only start once the action has arrived
    __shared__ int s_mult[DM], s_inv20[DM], s_dest[DM];
    for (int i = threadIdx.x; i < D; i += blockDim.x) { s_mult[i] = cp->mult[i]; s_inv20[i] = (int)cp->inv20[i]; s_dest[i] = cp->dest[i]; }

    // the env's packed records and per-sender queue records, issued before anything depends on them: {now, wake},
    // {counter, rvmask, last_abs | done << 31, flags}, the noise-state bytes of all radios
    constexpr int DR = DT > 0 ? DT : 1;
    constexpr int NXW = DT > 0 ? (DT + 1 + 15) / 16 : 1;          // 16-byte words of the noise-state record
    const int64_t el = e < N ? e : 0;
    GwRec qr[DR];
    if (DT > 0) {
#pragma unroll
        for (int i = 0; i < DR; ++i) qr[i] = st.qrec[(int64_t)i * N + el];
    }
    const double2 xw0 = reinterpret_cast<const double2*>(st.xw)[el];
    const uint4 xc0 = reinterpret_cast<const uint4*>(st.xc)[el];
    uint4 xsw[NXW];
#pragma unroll
    for (int w = 0; w < NXW; ++w) xsw[w] = DT > 0 ? reinterpret_cast<const uint4*>(st.xs + (size_t)el * st.XB)[w] : make_uint4(0u, 0u, 0u, 0u);
    const int d = device[el];
    int du = duration[el];
    __syncthreads();
    // every load issued above has LANDED before the walk branches (opaque uses: the compiler waits here).  Without this the
    // bad-action path reaches the kernel's tail with those loads formally outstanding, the tail reuses their registers, and
    // the wait the compiler then places at the join -- vmcnt counts loads and stores in one order -- makes the GOOD path
    // sit out its own record stores before the totals (in-kernel stamps: 2 700 cycles)
#define GW_LANDED(x) asm volatile("" : "+v"(x))
    if (DT > 0) {
#pragma unroll
        for (int i = 0; i < DR; ++i) { GW_LANDED(qr[i].x); GW_LANDED(qr[i].y); GW_LANDED(qr[i].z); GW_LANDED(qr[i].w); }
#pragma unroll
        for (int w = 0; w < NXW; ++w) { GW_LANDED(xsw[w].x); GW_LANDED(xsw[w].y); GW_LANDED(xsw[w].z); GW_LANDED(xsw[w].w); }
    }
    GW_LANDED(du);
#undef GW_LANDED
    STAMP(1);
    // noise state of radio j (run-time j): from the record's registers (DT > 0) or from memory
    auto xs_get = [&](int j) -> uint8_t {
        if (DT > 0) {
            uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
            for (int w = 0; w < NXW; ++w) {
                // (opaque register values first: selecting struct fields by a run-time index becomes a stack copy)
                uint32_t a0 = xsw[w].x, a1 = xsw[w].y, a2 = xsw[w].z, a3 = xsw[w].w;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                const bool h = (j >> 4) == w;
                w0 = h ? a0 : w0; w1 = h ? a1 : w1; w2 = h ? a2 : w2; w3 = h ? a3 : w3;
            }
            const uint32_t lo = (j & 4) ? w1 : w0, hi = (j & 4) ? w3 : w2;
            const uint32_t v = (j & 8) ? hi : lo;
            return (uint8_t)((v >> ((j & 3) * 8)) & 0xffu);
        }
        return st.xs[(size_t)e * st.XB + j];
    };

    // SPLIT: the helper also writes the radios' noise-state record (every radio heard the announcement and, if any, d's
    // data): the lookups are its own, what it needs from the walk -- whether data was sent, d's and the RRM's new states --
    // comes through LDS behind a workgroup barrier that BOTH waves pass exactly once, at their converged ends.  Table PHY with
    // idempotent states and no receive-mode peers only (both wave-uniform); otherwise the walker keeps the record.
    __shared__ uint32_t s_info[SPLIT ? 64 : 1];
    const bool xs_by_helper = SPLIT && c.idem_states != 0 && !c.peer_receive;
    if (SPLIT && helper) {
        Tally kh = {0, 0, 0, 0, 0};
        const bool valid_h = e < N && (unsigned)d < (unsigned)D && (unsigned)du < (unsigned)c.max_duration;
        uint8_t n1h[DR], n2h[DR];
        if (xs_by_helper && valid_h) {
#pragma unroll
            for (int j = 0; j < DR; ++j) {
                const uint8_t s0 = xs_get(j);
                n1h[j] = s_h1[j * S + s0];
                n2h[j] = g_h2[(j * D + d) * S + s0];
            }
        }
        if (valid_h) {
            const StepMath m(c);
            const double interval = c.counter_interval;
            const uint32_t bound = (uint32_t)c.counter_bound;
            const uint32_t base_bytes = (uint32_t)(c.mac_hdr + c.net_hdr);
            const int slots = du * c.duration_factor;                     // counter_traffic.py:149
            const int La = ndigits(slots) + (c.float_duration ? 2 : 0);
            const TxTimes an = tx_times(m, xw0.x, c.hdr_dur, m.over_rate((double)(La * 8)));
            const double t_end = an.t_e + (double)(slots + 1) * c.slot;   // simple_stack.py:557-558
            // every counter tick up to and including t_end (the walker counts the same running-sum sequence piecewise)
            double wake = xw0.y, delta = 0.0;
            uint32_t n_ticks = 0;
            {
                uint32_t nj = 0;
                double wj = wake;
                bool tiej = false, sane = false;
                const bool span_ok = c.fast_ticks && gw_tick_span_ok(wake, t_end, interval, &delta);
                gw_tick_jump_lo(wake, t_end, delta, c.inv_interval_lo, true, &nj, &wj, &tiej, &sane);
                if (span_ok && sane) {
                    n_ticks = nj;
                } else {
                    for (;;) {
                        const double w1 = wake + interval, w2 = w1 + interval, w3 = w2 + interval, w4 = w3 + interval;
                        const bool b0 = wake <= t_end, b1 = w1 <= t_end, b2 = w2 <= t_end, b3 = w3 <= t_end;
                        n_ticks += (uint32_t)b0 + (uint32_t)b1 + (uint32_t)b2 + (uint32_t)b3;   // monotone
                        wake = w4;
                        if (!b3) break;
                    }
                }
            }
            const uint32_t ctr0 = xc0.x;
#pragma unroll
            for (int i = 0; i < DR; ++i) {
                const uint32_t mult_i = (uint32_t)c.mult[i];
                if (i != d && n_ticks != 0u && mult_i != 0u) {
                    GwRunQ ri = load_q(qr[i]);
                    gw_runq_ticks(ri, n_ticks, ctr0, bound, base_bytes, st.runs + (((int64_t)e * D + i) << 7), mult_i, c.inv20[i], kh);
                    st.qrec[(int64_t)i * N + e] = store_q(ri);
                }
            }
            if (PER_ENV_STATS) {
                if (kh.app) atomicAdd(&st.pe_stats[2 * N + e], (unsigned long long)kh.app);
                if (kh.drop) atomicAdd(&st.pe_stats[4 * N + e], (unsigned long long)kh.drop);
            }
        }
        publish_totals(st.totals, kh, 0u, 0u, 0u);
        __syncthreads();                                   // the walker's word is in LDS
        if (xs_by_helper && valid_h) {
            const uint32_t info = s_info[threadIdx.x & 63];
            if (info & 1u) {                               // a good step: {1, data sent, d's state, the RRM's state}
                const bool data = (info >> 1) & 1u;
                uint32_t nbx[16 * NXW];
#pragma unroll
                for (int b = 0; b < 16 * NXW; ++b) {
                    const uint4& w = xsw[b >> 4];
                    const uint32_t word = ((b >> 2) & 3) == 0 ? w.x : (((b >> 2) & 3) == 1 ? w.y : (((b >> 2) & 3) == 2 ? w.z : w.w));
                    nbx[b] = (word >> ((b & 3) * 8)) & 0xffu;
                }
#pragma unroll
                for (int j = 0; j < DT; ++j) nbx[j] = j == d ? ((info >> 8) & 0xffu) : (data ? (uint32_t)n2h[j] : (uint32_t)n1h[j]);
                nbx[DT] = (info >> 16) & 0xffu;
#pragma unroll
                for (int w = 0; w < NXW; ++w) {
                    const int b = 16 * w;
                    uint4 o;
                    o.x = nbx[b + 0] | (nbx[b + 1] << 8) | (nbx[b + 2] << 16) | (nbx[b + 3] << 24);
                    o.y = nbx[b + 4] | (nbx[b + 5] << 8) | (nbx[b + 6] << 16) | (nbx[b + 7] << 24);
                    o.z = nbx[b + 8] | (nbx[b + 9] << 8) | (nbx[b + 10] << 16) | (nbx[b + 11] << 24);
                    o.w = nbx[b + 12] | (nbx[b + 13] << 8) | (nbx[b + 14] << 16) | (nbx[b + 15] << 24);
                    if (o.x != xsw[w].x || o.y != xsw[w].y || o.z != xsw[w].z || o.w != xsw[w].w)
                        reinterpret_cast<uint4*>(st.xs + (size_t)e * st.XB)[w] = o;
                }
            }
        }
        return;
    }

    // What the step leaves behind is held in registers and STORED AT THE VERY END, behind the wave's totals: the tail of the
    // kernel reuses registers, and a register that is the data of a store in flight is not free before that store has
    // completed (the compiler waits on vmcnt) -- with the stores first, the totals sat out a full store round trip.
    bool out_good = false, out_bad = false;
    int32_t out_obs = 0;
    float out_rew = 0.0f;
    uint8_t out_dn = 0;
    GwRec qout[DR];
    double2 out_xw = make_double2(0.0, 0.0);
    uint4 out_xc = make_uint4(0u, 0u, 0u, 0u);
    uint4 out_xs[NXW];
    uint32_t out_xs_dirty = 0u;
    Tally out_pe = {0, 0, 0, 0, 0};
    GwRec out_rec_d = {0u, 0u, 0u, 0u};
    const int out_d = d;
    uint32_t out_info = 0u;

    if (e < N) {
        uint32_t fl = xc0.w;
        uint32_t rvm = xc0.y;
        int32_t last_abs = (int32_t)(xc0.z & 0x7fffffffu);
        uint8_t dn = (uint8_t)(xc0.z >> 31);
        const int pv = c.payload_value;

        if ((unsigned)d >= (unsigned)D || (unsigned)du >= (unsigned)c.max_duration) {
            // counter_traffic.py:147 asserts; a batched step cannot raise per env: flag + skip
            fl |= GW_FLAG_BADACT;
            k_bad = 1;
            const int32_t latest = pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u));
            out_bad = true;
            out_obs = latest + c.counter_bound;
            out_dn = dn;
            out_xc = make_uint4(xc0.x, xc0.y, xc0.z, fl);
        } else {
            k_steps = 1;
            const StepMath m(c);
            const double slot = c.slot, br = c.bit_rate;
            const double hd = c.hdr_dur, hdr_bits = c.hdr_bits;
            const double interval = c.counter_interval;
            const uint32_t bound = (uint32_t)c.counter_bound;
            const uint32_t base_bytes = (uint32_t)(c.mac_hdr + c.net_hdr);
            const int mh = c.mac_hdr;

            const double t_a = xw0.x;
            double wake = xw0.y;
            const uint32_t ctr0 = xc0.x;
            const int slots = du * c.duration_factor;                     // counter_traffic.py:149

            STAMP(2);
            // ---- A.1 / A.2: announcement ---------------------------------------------
            const int La = ndigits(slots) + (c.float_duration ? 2 : 0);  // len(str(10000.0)) == len("10000") + 2
            const double pd_a = m.over_rate((double)(La * 8));
            const TxTimes an = tx_times(m, t_a, hd, pd_a);
            k.tx++;
            const bool per_env = DYN && st.prx_env != nullptr;
            auto lp = [&](int from, int to) -> double {           // link power from -> to, mW
                return per_env ? gw_link<true>(st, R, from, to, (uint32_t)e) : gw_link<false>(st, R, from, to, (uint32_t)e);
            };
            uint8_t s_d_old = 0, s_d = 0;
            const bool cls_valid = !DYN && t_a < c.cls_limit;
            const bool idem = !DYN && c.idem_states != 0;
            bool granted;
            if (DYN) {                                           // simple_stack.py:82 (+p), :166-167 (noise = received - signal), :154 (-p)
                const double p_a = lp(RRM, d);
                const double up = gw_rx(st, R, d, e) + p_a;
                const double noise = up - p_a;
                if (!(noise >= 0.0)) fl |= GW_FLAG_REFEXC;
                granted = receive(m, ber_bpsk_dev(p_a, noise, c.ten_log_br), an, br, hdr_bits, (double)(La * 8) * c.coded_factor, fl);
                gw_rx(st, R, d, e) = up + (-p_a);
            } else {
                s_d_old = xs_get(d);
                s_d = s_h1[d * S + s_d_old];
                granted = decode(m, s_cls[d * S + s_d], cls_valid, s_ber[d * S + s_d], an, br, hdr_bits,
                                 (double)(La * 8) * c.coded_factor, fl);
            }
            const double t_r = an.t_e;
            const double t_end = t_r + (double)(slots + 1) * slot;       // simple_stack.py:557-558

            STAMP(3);
            // ---- A.3: window at sender d -----------------------------------------------
            GwRec qr_d = {0u, 0u, 0u, 0u};
            if (DT > 0) {
#pragma unroll
                for (int i = 0; i < DR; ++i) {
                    // the words as opaque register values first: a select between array elements by a run-time index is
                    // folded by the compiler into dynamic addressing of a stack copy of the array
                    uint32_t w0 = qr[i].x, w1 = qr[i].y, w2 = qr[i].z, w3 = qr[i].w;
                    asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
                    const bool hit = i == d;
                    qr_d.x = hit ? w0 : qr_d.x; qr_d.y = hit ? w1 : qr_d.y; qr_d.z = hit ? w2 : qr_d.z; qr_d.w = hit ? w3 : qr_d.w;
                }
            } else {
                qr_d = st.qrec[(int64_t)d * N + e];
            }
            GwRunQ rd = load_q(qr_d);
            uint64_t* ring_d = st.runs + (((int64_t)e * D + d) << 7);
            const uint32_t mult_d = (uint32_t)s_mult[d];
            const uint32_t inv20_d = (uint32_t)s_inv20[d];
            int n_data = 0;
            uint32_t n_ticks = 0;                                         // counter ticks inside this step
            const uint8_t s_r_old = DYN ? (uint8_t)0 : xs_get(RRM);
            uint8_t s_r = s_r_old;
            const uint8_t s_r1 = DYN ? (uint8_t)0 : s_rr[d * S + s_r_old];        // the RRM after one packet of d
            const double ber_x1 = DYN ? 0.0 : s_ber[(D + d) * S + s_r1];
            const uint32_t cls_x1 = DYN ? 0u : s_cls[(D + d) * S + s_r1];
            // every other radio's noise state after the announcement (n1) and after the announcement AND >= 1 data packet of d
            // (n2): looked up NOW, consumed after the window -- a load issued behind the step's first stores would wait for
            // those stores to complete (vmcnt counts loads and stores in one order), a microsecond apiece
            uint8_t n1[DR], n2[DR];
            if (DT > 0 && !DYN && !xs_by_helper) {
#pragma unroll
                for (int j = 0; j < DR; ++j) {
                    const uint8_t s0 = xs_get(j);
                    n1[j] = s_h1[j * S + s0];
                    n2[j] = g_h2[(j * D + d) * S + s0];
                }
            }
            // live PHY: the RRM's (and a receive-mode peer's) received power, and the BER cached per noise value
            double rx_r = 0.0, rx_r0 = 0.0, p_x = 0.0, ber_xd = 0.0, nz_r_prev = -1.0;
            double rx_p = 0.0, rx_p0 = 0.0, p_p = 0.0, ber_pd = 0.0, nz_p_prev = -1.0;
            if (DYN) { rx_r = rx_r0 = gw_rx(st, R, RRM, e); p_x = lp(d, RRM); }
            // receive-mode MAC at the destination (simple_stack.py:443-448): it is idle during d's window (its own
            // window, the only thing that blocks its phyIn handler, ended a slot before the previous step did)
            const int dest_d = c.peer_receive ? s_dest[d] : d;
            const int j_peer = (c.peer_receive && dest_d != d) ? dest_d : -1;
            uint8_t s_p = 0, s_p_old = 0;
            uint32_t n_peer = 0;
            if (j_peer >= 0) {
                if (DYN) {
                    rx_p0 = gw_rx(st, R, j_peer, e);
                    const double pa = lp(RRM, j_peer);
                    rx_p = (rx_p0 + pa) + (-pa);                              // it heard the announcement too
                    p_p = lp(d, j_peer);
                } else {
                    s_p_old = xs_get(j_peer);
                    s_p = st.trans[((int64_t)j_peer * R + RRM) * S + s_p_old];    // it heard the announcement too
                }
            }

            // all counter ticks with wake < t (or <= t): counted in f64 four at a time (the running sum w += dt is the
            // reference's arithmetic, counter_traffic.py:61), applied to d's queue in one go
            auto ticks_to = [&](double t, bool inclusive) __attribute__((always_inline)) {
                uint32_t kk = 0;
                for (;;) {
                    const double w1 = wake + interval, w2 = w1 + interval, w3 = w2 + interval, w4 = w3 + interval;
                    const bool b0 = inclusive ? (wake <= t) : (wake < t);
                    const bool b1 = inclusive ? (w1 <= t) : (w1 < t);
                    const bool b2 = inclusive ? (w2 <= t) : (w2 < t);
                    const bool b3 = inclusive ? (w3 <= t) : (w3 < t);
                    if (inclusive && (wake == t || w1 == t || w2 == t || w3 == t)) fl |= GW_FLAG_TIE;
                    const uint32_t n = (uint32_t)b0 + (uint32_t)b1 + (uint32_t)b2 + (uint32_t)b3;   // monotone
                    wake = b3 ? w4 : (b2 ? w3 : (b1 ? w2 : (b0 ? w1 : wake)));
                    kk += n;
                    if (!b3) break;
                }
                gw_runq_ticks(rd, kk, ctr0 + n_ticks, bound, base_bytes, ring_d, mult_d, inv20_d, k);
                n_ticks += kk;
            };

            // the same in one jump where the step qualifies (gw_fastmath.h: preconditions once per step, then a dozen
            // instructions per call), else by the loop
            double delta = 0.0;
            const bool span_ok = c.fast_ticks && gw_tick_span_ok(wake, t_end, interval, &delta);
            auto ticks_upto = [&](double t, bool inclusive) __attribute__((always_inline)) {
                uint32_t nj = 0;
                double wj = wake;
                bool tiej = false, sane = false;
                gw_tick_jump_lo(wake, t, delta, c.inv_interval_lo, inclusive, &nj, &wj, &tiej, &sane);
                if (span_ok && sane) {
                    wake = wj;
                    if (tiej) fl |= GW_FLAG_TIE;
                    gw_runq_ticks(rd, nj, ctr0 + n_ticks, bound, base_bytes, ring_d, mult_d, inv20_d, k);
                    n_ticks += nj;
                } else {
                    ticks_to(t, inclusive);
                }
            };

            STAMP(4);
            if (granted) {
                const double total = (double)slots * slot;               // simple_stack.py:400
                const double stopw = t_r + total;                        // :401 (== timeout time :406)
                double cur = t_r;
                // ties at the window start: the MAC's process initialisation is URGENT, so it runs first
                ticks_upto(cur, false);
                // ---- the window loop, straight-line form (the default kernel's, ct_step_sfx.hip: a lone wave per SIMD issues
                //      an instruction every ~8 cycles whatever it is, so the per-packet decisions are settled for the whole
                //      step up front where they can be).  A lane takes it when tick jumps apply (span_ok), the exact fast
                //      forms of fmod and division hold up to t_end, the RRM's decode outcome is certain by class and its noise
                //      state idempotent, no receive-mode peer listens, and t_r >= 2 (t_end - t_r): then every packet's
                //      stop - t_s is exact (Sterbenz) and the completion event t_s + (stop - t_s) IS stop.  One exit
                //      condition, no branch inside but the run-queue's own (rarely taken) general paths; it ends when the
                //      window closes, the next packet does not fit, or the queue runs empty (the general loop below then
                //      waits for the tick).  GW_FLAG_CARRY can only be raised by a window's last packet: tested once.
                bool more = true;
                uint32_t pops = 0;
                {
                    const double span = t_end - t_r;
                    const bool straight = !DYN && span_ok && mult_d != 0u && idem && cls_valid && cls_x1 != (uint32_t)GW_CLS_COMPUTE &&
                                          m.fast_fmod && m.fast_div && t_end < m.fmod_limit && t_r >= span + span && j_peer < 0;
                    if (straight && rd.len != 0u) {
                        uint32_t chk = 0;
                        uint32_t s = rd.H.v0;
                        bool go = (stopw - cur) > gw_fast_div((double)(s * 8u), m.dr, m.rcp_dr);       // :418-420
                        while (go) {
                            const double pd = gw_fast_div((double)(((int)s - mh) * 8), m.dr, m.rcp_dr);
                            const double t_s = cur + (m.slot - gw_fast_fmod_lo(cur, m.slot, c.inv_slot_lo));
                            const double t_e = t_s + (hd + pd);
                            uint32_t nj = 0;
                            double wj = wake;
                            bool tiej = false, sane = false;
                            gw_tick_jump_lo(wake, t_e, delta, c.inv_interval_lo, true, &nj, &wj, &tiej, &sane);
                            chk |= (sane ? 0u : (uint32_t)GW_FLAG_INTERNAL) | (tiej ? (uint32_t)GW_FLAG_TIE : 0u);
                            // the pop (:425) and the ticks inside the transmission, in one queue operation (gw_runq.h)
                            gw_runq_pop1_ticks(rd, nj, ctr0 + n_ticks, bound, base_bytes, ring_d, mult_d, inv20_d, k);
                            n_ticks += nj;
                            wake = wj;
                            cur = t_e;
                            pops++;
                            s = rd.H.v0;
                            go = cur < stopw && rd.len != 0u && (stopw - cur) > gw_fast_div((double)(s * 8u), m.dr, m.rcp_dr);
                        }
                        more = cur < stopw && rd.len == 0u;               // an empty queue waits for a tick: general loop
                        fl |= chk | ((pops && !(cur < t_end)) ? (uint32_t)GW_FLAG_CARRY : 0u);
                    }
                }
                if (pops) {                                               // devices.py:163-168, counter_traffic.py:75-80
                    const bool okx = cls_x1 == (uint32_t)GW_CLS_OK;
                    k.pop += pops;
                    k.tx += pops;
                    n_data += (int)pops;
                    s_r = s_r1;
                    k.deliv += okx ? pops : 0u;
                    rvm |= okx ? (1u << d) : 0u;
                    dn = (okx && pv == c.counter_bound) ? (uint8_t)1 : dn;
                }
                if (more)
                for (;;) {
                    if (rd.len == 0u) {                                   // :409-416
                        // (a silent sender, mult 0, never signals packet-added: the MAC waits for the timeout)
                        if (mult_d > 0u && wake < stopw) {
                            cur = wake;
                            wake = wake + interval;
                            gw_runq_ticks(rd, 1u, ctr0 + n_ticks, bound, base_bytes, ring_d, mult_d, inv20_d, k);
                            n_ticks++;
                        } else break;
                    }
                    STAMP(12);
                    const uint32_t s = rd.H.v0;                           // the head run's first packet
                    const double need = m.over_rate((double)(s * 8u));    // messages.py:67-75
                    if (!((stopw - cur) > need)) break;                   // :418-420 idle until the window ends
                    gw_runq_pop_front(rd, 1u, ring_d, mult_d, inv20_d, base_bytes + bound);   // :425
                    k.pop++;
                    const int pay = (int)s - mh;
                    const TxTimes x = tx_times(m, cur, hd, m.over_rate((double)(pay * 8)));
                    k.tx++;
                    n_data++;
                    STAMP(13);
                    double ber_x = ber_x1;
                    uint32_t cls_x = cls_x1;
                    if (DYN) {
                        const double up = rx_r + p_x;
                        const double noise = up - p_x;
                        if (!(noise >= 0.0)) fl |= GW_FLAG_REFEXC;
                        if (noise != nz_r_prev) { ber_xd = ber_bpsk_dev(p_x, noise, c.ten_log_br); nz_r_prev = noise; }
                        ber_x = ber_xd;
                        cls_x = GW_CLS_COMPUTE;
                        rx_r = up + (-p_x);
                    } else if (idem) {
                        s_r = s_r1;
                    } else {                                              // not seen with f64 link powers; full tables in HBM
                        s_r = st.trans[((int64_t)RRM * R + d) * S + s_r];
                        ber_x = st.ber[((int64_t)RRM * R + d) * S + s_r];
                        cls_x = st.cls[((int64_t)RRM * R + d) * S + s_r];
                    }
                    const bool ok = decode(m, cls_x, cls_valid, ber_x, x, br, hdr_bits, (double)(pay * 8) * c.coded_factor, fl);
                    if (ok) {                                             // devices.py:163-168, counter_traffic.py:75-80
                        k.deliv++;
                        rvm |= (1u << d);
                        if (pv == c.counter_bound) dn = 1;
                    }
                    if (j_peer >= 0) {
                        double ber_p;
                        if (DYN) {
                            const double up = rx_p + p_p;
                            const double noise = up - p_p;
                            if (!(noise >= 0.0)) fl |= GW_FLAG_REFEXC;
                            if (noise != nz_p_prev) { ber_pd = ber_bpsk_dev(p_p, noise, c.ten_log_br); nz_p_prev = noise; }
                            ber_p = ber_pd;
                            rx_p = up + (-p_p);
                        } else {
                            s_p = st.trans[((int64_t)j_peer * R + d) * S + s_p];
                            ber_p = st.ber[((int64_t)j_peer * R + d) * S + s_p];
                        }
                        if (receive(m, ber_p, x, br, hdr_bits, (double)(pay * 8) * c.coded_factor, fl)) n_peer++;
                    }
                    if (!(x.t_e < t_end)) fl |= GW_FLAG_CARRY;
                    STAMP(14);
                    // ticks are older events than the MAC's resume at t_e: they go first
                    ticks_upto(x.t_e, true);
                    STAMP(15);
                    cur = x.t_e;
                    if (!(cur < stopw)) break;                            // window timeout already processed
                }
            }

            STAMP(5);
            // ---- A.5: remaining ticks up to the end of the step ----------------------------
            ticks_upto(t_end, true);
            // the same n_ticks ticks d's walk just counted reach every other sender's queue (all senders tick together); the
            // records are STORED at the very end of the step, behind every load (see above)
            STAMP(6);
            const GwRec rec_d = store_q(rd);
            if (DT > 0) {
#pragma unroll
                for (int i = 0; i < DR; ++i) {
                    GwRec ro = qr[i];
                    const uint32_t mult_i = (uint32_t)c.mult[i];
                    if (!SPLIT && i != d && n_ticks != 0u && mult_i != 0u) {
                        GwRunQ ri = load_q(qr[i]);
                        gw_runq_ticks(ri, n_ticks, ctr0, bound, base_bytes, st.runs + (((int64_t)e * D + i) << 7), mult_i, c.inv20[i], k);
                        ro = store_q(ri);
                    }
                    const bool hit = i == d;
                    qout[i].x = hit ? rec_d.x : ro.x; qout[i].y = hit ? rec_d.y : ro.y;
                    qout[i].z = hit ? rec_d.z : ro.z; qout[i].w = hit ? rec_d.w : ro.w;
                }
            } else {
                for (int i = 0; i < D; ++i) {
                    if (i == d) continue;
                    const uint32_t mult_i = (uint32_t)s_mult[i];
                    if (n_ticks != 0u && mult_i != 0u) {
                        GwRunQ ri = load_q(st.qrec[(int64_t)i * N + e]);
                        gw_runq_ticks(ri, n_ticks, ctr0, bound, base_bytes, st.runs + (((int64_t)e * D + i) << 7), mult_i, (uint32_t)s_inv20[i], k);
                        st.qrec[(int64_t)i * N + e] = store_q(ri);
                    }
                }
                st.qrec[(int64_t)d * N + e] = rec_d;
            }
            STAMP(7);
            uint32_t ctr_new = ctr0 + n_ticks;                            // `if counter < bound: counter += 1` per tick
            ctr_new = (ctr0 >= bound) ? ctr0 : (ctr_new < bound ? ctr_new : bound);

            // ---- rx-power state of every radio (simple_stack.py:130-157) ---------------------
            if (DYN) {
                if (st.talk) st.talk[e] |= (1ull << RRM) | (n_data ? (1ull << d) : 0ull);   // whose attenuation models exist now
                if (rx_r != rx_r0) gw_rx(st, R, RRM, e) = rx_r;
                if (j_peer >= 0) {
                    if (rx_p != rx_p0) gw_rx(st, R, j_peer, e) = rx_p;
                    if (n_peer) st.peer_rx[(int64_t)j_peer * N + e] += n_peer;
                }
                for (int j = 0; j < D; ++j) {
                    if (j == d || j == j_peer) continue;
                    const double a0 = gw_rx(st, R, j, e);
                    const double pa = lp(RRM, j);
                    double a = (a0 + pa) + (-pa);
                    if (n_data) {
                        const double pd = lp(d, j);
                        for (int n = 0; n < n_data; ++n) {
                            const double b = (a + pd) + (-pd);
                            if (b == a) break;                            // a fixed point of the (+p, -p) pair stays one
                            a = b;
                        }
                    }
                    if (!(a >= 0.0)) fl |= GW_FLAG_REFEXC;
                    if (a != a0) gw_rx(st, R, j, e) = a;
                }
            } else {
                if (j_peer >= 0 && n_peer) st.peer_rx[(int64_t)j_peer * N + e] += n_peer;
                // every other radio heard the announcement and, if any, d's data; d heard only the announcement
                auto after = [&](int j, uint8_t s0) -> uint8_t {
                    if (j == d) return s_d;
                    if (j == j_peer) return s_p;
                    if (idem) {
                        // (two typed loads and a select of the VALUES: a select between the LDS pointer and the global one is
                        //  a generic pointer, whose flat load waits for every outstanding store of the wave)
                        const uint8_t a = DT > 0 ? n2[DT > 0 ? j : 0] : g_h2[(j * D + d) * S + s0];
                        const uint8_t b = DT > 0 ? n1[DT > 0 ? j : 0] : s_h1[j * S + s0];
                        return n_data ? a : b;
                    }
                    uint8_t sj = st.trans[((int64_t)j * R + RRM) * S + s0];
                    for (int n = 0; n < n_data; ++n) sj = st.trans[((int64_t)j * R + d) * S + sj];
                    return sj;
                };
                if (xs_by_helper) {
                    out_info = 1u | (n_data ? 2u : 0u) | ((uint32_t)s_d << 8) | ((uint32_t)s_r << 16);
                } else if (DT > 0) {
                    uint32_t nbx[16 * NXW];
#pragma unroll
                    for (int b = 0; b < 16 * NXW; ++b) {
                        const uint4& w = xsw[b >> 4];
                        const uint32_t word = ((b >> 2) & 3) == 0 ? w.x : (((b >> 2) & 3) == 1 ? w.y : (((b >> 2) & 3) == 2 ? w.z : w.w));
                        nbx[b] = (word >> ((b & 3) * 8)) & 0xffu;
                    }
#pragma unroll
                    for (int j = 0; j < DT; ++j) nbx[j] = after(j, (uint8_t)nbx[j]);
                    nbx[DT] = s_r;
#pragma unroll
                    for (int w = 0; w < NXW; ++w) {
                        const int b = 16 * w;
                        uint4 o;
                        o.x = nbx[b + 0] | (nbx[b + 1] << 8) | (nbx[b + 2] << 16) | (nbx[b + 3] << 24);
                        o.y = nbx[b + 4] | (nbx[b + 5] << 8) | (nbx[b + 6] << 16) | (nbx[b + 7] << 24);
                        o.z = nbx[b + 8] | (nbx[b + 9] << 8) | (nbx[b + 10] << 16) | (nbx[b + 11] << 24);
                        o.w = nbx[b + 12] | (nbx[b + 13] << 8) | (nbx[b + 14] << 16) | (nbx[b + 15] << 24);
                        out_xs[w] = o;
                        if (o.x != xsw[w].x || o.y != xsw[w].y || o.z != xsw[w].z || o.w != xsw[w].w) out_xs_dirty |= 1u << w;
                    }
                } else {
                    uint8_t* xr = st.xs + (size_t)e * st.XB;
                    for (int j = 0; j < D; ++j) {
                        const uint8_t s0 = xr[j];
                        const uint8_t s1 = after(j, s0);
                        if (s1 != s0) xr[j] = s1;
                    }
                    if (s_r != s_r_old) xr[RRM] = s_r;
                }
            }

            STAMP(8);
            // ---- interpreter feedback (counter_traffic.py:85-112, envs/core.py:142-153) -------
            const int32_t latest = pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u));
            const int32_t abs_d = latest < 0 ? -latest : latest;
            int32_t r = last_abs - abs_d;
            last_abs = abs_d;
            r = r > 10 ? 10 : (r < -10 ? -10 : r);
            out_good = true;
            out_obs = latest + c.counter_bound;
            out_rew = (float)r;
            out_dn = dn;
            out_xw = make_double2(t_end, wake);
            out_xc = make_uint4(ctr_new, rvm, (uint32_t)last_abs | ((uint32_t)dn << 31), fl);
            out_pe = k;
            out_rec_d = rec_d;
        }
        fl_new = fl;
    }
    if (SPLIT) {                                           // (converged: every lane of both waves gets here exactly once)
        s_info[threadIdx.x & 63] = out_info;
        __syncthreads();
    }
    STAMP(9);
    publish_totals(st.totals, k, k_steps, k_bad, fl_new);
    STAMP(10);
    // ---- the step's stores, last of all ----------------------------------------------------------------------------
    if (out_good || out_bad) {
        obs[e] = out_obs;
        reward[e] = out_rew;
        done[e] = out_dn;
    }
    if (out_bad && out_xc.w != xc0.w) st.xc[(size_t)e * 4 + 3] = out_xc.w;
    if (out_good) {
        if (DT > 0) {
#pragma unroll
            for (int w = 0; w < NXW; ++w)
                if ((out_xs_dirty >> w) & 1u) reinterpret_cast<uint4*>(st.xs + (size_t)e * st.XB)[w] = out_xs[w];
            if (SPLIT) {                                   // the addressed sender's record only: the others are the helper's
                st.qrec[(int64_t)out_d * N + e] = out_rec_d;
            } else {
#pragma unroll
                for (int i = 0; i < DR; ++i) st.qrec[(int64_t)i * N + e] = qout[i];
            }
        }
        reinterpret_cast<double2*>(st.xw)[e] = out_xw;
        reinterpret_cast<uint4*>(st.xc)[e] = out_xc;
        if (PER_ENV_STATS) {
            st.pe_stats[0 * N + e] += out_pe.tx;
            st.pe_stats[1 * N + e] += out_pe.deliv;
            st.pe_stats[3 * N + e] += out_pe.pop;
            if (SPLIT) {                                   // (the helper adds the other senders' share to these two)
                if (out_pe.app) atomicAdd(&st.pe_stats[2 * N + e], (unsigned long long)out_pe.app);
                if (out_pe.drop) atomicAdd(&st.pe_stats[4 * N + e], (unsigned long long)out_pe.drop);
            } else {
                st.pe_stats[2 * N + e] += out_pe.app;
                st.pe_stats[4 * N + e] += out_pe.drop;
            }
        }
    }
    STAMP(11);
}

// fresh env: counters 1 (counter_traffic.py:48), first tick at t=0, all radios at thermal noise
__global__ void ct_init_kernel(GwState st)
{
    const int64_t N = st.N;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int D = st.cst->D, R = st.cst->R;
    reinterpret_cast<double2*>(st.xw)[e] = make_double2(st.cst->start_time, st.cst->no_traffic ? (double)INFINITY : st.cst->start_time);
    reinterpret_cast<uint4*>(st.xc)[e] = make_uint4(1u, 0u, 0u, 0u);      // counter 1 (counter_traffic.py:48), nothing received, no flags
    if (st.qrec) for (int i = 0; i < D; ++i) st.qrec[(int64_t)i * N + e] = GwRec{0u, 0u, 0u, 0u};
    (void)R;
    for (int b = 0; b < st.XB; ++b) st.xs[(size_t)e * st.XB + b] = 0;
    if (st.pe_stats) for (int s = 0; s < 5; ++s) st.pe_stats[(int64_t)s * N + e] = 0ull;
    if (st.peer_rx) for (int i = 0; i < D; ++i) st.peer_rx[(int64_t)i * N + e] = 0u;
}

// SimpleNetworkDevice.send -> SimpleMac.networkInHandler: one packet appended per env (devices.py:84-86,
// simple_stack.py:463-471)
__global__ void ct_enqueue_kernel(GwState st, int sender, const int32_t* __restrict__ payload_bytes)
{
    const int64_t N = st.N;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int32_t pb = payload_bytes[e];
    if (pb < 0) return;
    const GwDevConst& c = *st.cst;
    GwRunQ q = load_q(st.qrec[(int64_t)sender * N + e]);
    uint64_t* ring = st.runs + (((int64_t)e * c.D + sender) << 7);
    uint32_t dropped = 0u;
    if (q.len == (uint32_t)GW_QUEUE_CAP) {                // deque(maxlen=100): drop the oldest
        gw_runq_pop_front(q, 1u, ring, (uint32_t)c.mult[sender], c.inv20[sender], (uint32_t)(c.mac_hdr + c.net_hdr + c.counter_bound));
        dropped = 1u;
    }
    gw_runq_append_literal(q, (uint32_t)(c.mac_hdr + c.net_hdr + pb), ring);
    st.qrec[(int64_t)sender * N + e] = store_q(q);
    if (st.pe_stats) { st.pe_stats[2 * N + e] += 1u; st.pe_stats[4 * N + e] += dropped; }
}

// counter_traffic.py:135-144 + :69-73 -- counters and interpreter only; time is NOT rewound
__global__ void ct_reset_kernel(GwState st, const uint8_t* __restrict__ mask, int32_t* __restrict__ obs)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= st.N) return;
    uint4 xc = reinterpret_cast<const uint4*>(st.xc)[e];
    if (!mask || mask[e]) {
        xc.x = 0u; xc.y = 0u; xc.z = 0u;                   // counter, receivedValues, lastAbs / done; the sticky flags stay
        reinterpret_cast<uint4*>(st.xc)[e] = xc;
    }
    if (obs) {
        const uint32_t rvm = xc.y;
        obs[e] = st.cst->payload_value * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u)) + st.cst->counter_bound;
    }
}

__global__ void ct_received_kernel(GwState st, int32_t* __restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int D = st.cst->D;
    if (idx >= st.N * D) return;
    const int64_t e = idx / D;
    const int i = (int)(idx - e * D);
    out[idx] = ((st.xc[(size_t)e * 4 + 1] >> i) & 1u) ? st.cst->payload_value : 0;
}

inline int check_launch()
{
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}

} // namespace

int gw_launch_init(const GwState& st, void* stream)
{
    const unsigned grid = (unsigned)((st.N + 255) / 256);
    hipLaunchKernelGGL(ct_init_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, st);
    return check_launch();
}

int gw_launch_reset(const GwState& st, const uint8_t* mask, int32_t* obs, void* stream)
{
    const unsigned grid = (unsigned)((st.N + 255) / 256);
    hipLaunchKernelGGL(ct_reset_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, st, mask, obs);
    return check_launch();
}

int gw_launch_step(const GwState& st, const int32_t* device, const int32_t* duration,
                   int32_t* obs, float* reward, uint8_t* done, void* stream)
{
    const unsigned blk = (unsigned)st.block;
    const unsigned grid = (unsigned)((st.N + blk - 1) / blk);
    static const bool no_split = getenv("GW_NO_SPLIT") != nullptr;                      // A/B switch
    const bool split = blk == 64u && !no_split;
#define GW_LAUNCH_SPLIT(DT_)                                                                                     \
    do {                                                                                                        \
        if (st.pe_stats)                                                                                        \
            hipLaunchKernelGGL((ct_step_kernel<DT_, true, false, true>), dim3(grid), dim3(128), 0, (hipStream_t)stream, \
                               st, device, duration, obs, reward, done);                                        \
        else                                                                                                    \
            hipLaunchKernelGGL((ct_step_kernel<DT_, false, false, true>), dim3(grid), dim3(128), 0, (hipStream_t)stream, \
                               st, device, duration, obs, reward, done);                                        \
    } while (0)
#define GW_LAUNCH_GENERIC(DT_)                                                                                   \
    do {                                                                                                        \
        if (split && (DT_) > 0) { constexpr int dts_ = ((DT_) > 0) ? (DT_) : 2; GW_LAUNCH_SPLIT(dts_); break; }      \
        if (st.pe_stats)                                                                                        \
            hipLaunchKernelGGL((ct_step_kernel<DT_, true, false>), dim3(grid), dim3(blk), 0, (hipStream_t)stream, \
                               st, device, duration, obs, reward, done);                                        \
        else                                                                                                    \
            hipLaunchKernelGGL((ct_step_kernel<DT_, false, false>), dim3(grid), dim3(blk), 0, (hipStream_t)stream, \
                               st, device, duration, obs, reward, done);                                        \
    } while (0)
    if (st.rxp) {                                        // live PHY (open noise-state set / per-env geometry)
#define GW_LAUNCH_DYN(DT_)                                                                                       \
    do {                                                                                                        \
        if (st.pe_stats)                                                                                        \
            hipLaunchKernelGGL((ct_step_kernel<DT_, true, true>), dim3(grid), dim3(blk), 0, (hipStream_t)stream, st, device, duration, obs, reward, done); \
        else                                                                                                    \
            hipLaunchKernelGGL((ct_step_kernel<DT_, false, true>), dim3(grid), dim3(blk), 0, (hipStream_t)stream, st, device, duration, obs, reward, done); \
    } while (0)
        switch (st.D) {                                  // the queue records in registers for the usual sender counts
        case 4:  GW_LAUNCH_DYN(4); break;
        case 16: GW_LAUNCH_DYN(16); break;
        default: GW_LAUNCH_DYN(0); break;
        }
#undef GW_LAUNCH_DYN
        return check_launch();
    }
    switch (st.D) {
    case 2:  GW_LAUNCH_GENERIC(2); break;
    case 3:  GW_LAUNCH_GENERIC(3); break;
    case 4:  GW_LAUNCH_GENERIC(4); break;
    case 8:  GW_LAUNCH_GENERIC(8); break;
    case 16: GW_LAUNCH_GENERIC(16); break;
    default: GW_LAUNCH_GENERIC(0); break;
    }
#undef GW_LAUNCH_GENERIC
#undef GW_LAUNCH_SPLIT
    return check_launch();
}

int gw_launch_enqueue(const GwState& st, int sender, const int32_t* payload_bytes, void* stream)
{
    const unsigned grid = (unsigned)((st.N + 255) / 256);
    hipLaunchKernelGGL(ct_enqueue_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, st, sender, payload_bytes);
    return check_launch();
}

int gw_launch_received(const GwState& st, int32_t* out, void* stream)
{
    const int64_t total = st.N * (int64_t)st.D;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(ct_received_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, st, out);
    return check_launch();
}
