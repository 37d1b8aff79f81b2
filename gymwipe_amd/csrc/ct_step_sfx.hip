// ct_step_sfx.hip -- the default step kernel (+ its init / reset / readout kernels).
//
// MAC queues in the "suffix" encoding of gw_queue.h: one length byte per sender, a tick counter and
// the reset breakpoints per env.  A counter tick is `len = min(len + mult, 100)`, a pop is `len -= 1`,
// the head packet's size is arithmetic on the tick index.  The step therefore touches no queue memory:
// it loads four 16-byte records per env, walks the event horizon of the step in registers, and stores
// the records back.  Per-env HBM layout (all offsets 32-bit):
//     tw[e] = {now, next tick}                      2 x f64
//     tk[e] = {tau, nbp, rvmask, last_abs | done<<31}   4 x u32: what a step changes, stored WHOLE (a 4-byte and an 8-byte
//                                                       store into two records went out as partial lines under write-through:
//                                                       1.46x the bytes the encoding needs, profiles/r2_final_sq)
//     ip[e] = {newest breakpoint, 2nd newest breakpoint}   written by reset / init only
//     qb[e] = bytes: len[0..D), rx-power state[0..D], pad to 16
//
// Step walk, reference file:line:
//   A.1  t_s = t_a + (slot - t_a % slot)                                   simtools.py:44-53
//   A.2  announcement heard by the addressed sender (header, payload)      simple_stack.py:214-286,536-558
//   A.3  window: pop + transmit while (stop - now) > bits/dataRate         simple_stack.py:397-434
//   A.4  RRM decodes each data packet -> interpreter                       networking/devices.py:163-168
//   A.5  t_end = t_r + (slots+1)*slot; ticks every 1 ms (running f64 sum)   counter_traffic.py:53-61
//   A.6  equal-time events: earlier-inserted first; process initialisation URGENT.
#include "ct_common.hip.h"
#include "gw_queue.h"

using namespace gwk;

// (in-kernel stamps of the diagnostic build: STAMP in ct_common.hip.h)

namespace {

template <class T>
__device__ __forceinline__ T ld(const void* base, uint32_t byte_off)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ void st_(void* base, uint32_t byte_off, const T& v)
{
    gwk::gw_store_wt(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off), v);
}

// byte `idx` (0..15) of a 16-byte record held in four registers, idx not known at compile time.  The words are
// taken BY VALUE: selecting between the fields of a struct object is folded by the compiler into dynamic addressing
// of a stack copy of it.
__device__ __forceinline__ uint32_t byte_of(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t idx)
{
    const uint32_t lo = (idx & 4u) ? y : x;
    const uint32_t hi = (idx & 4u) ? w : z;
    const uint32_t v = (idx & 8u) ? hi : lo;
    return (v >> ((idx & 3u) * 8u)) & 0xffu;
}
__device__ __forceinline__ uint32_t word_of(const uint4& w, int i)          // i compile-time after unrolling
{
    return i == 0 ? w.x : (i == 1 ? w.y : (i == 2 ? w.z : w.w));
}


// Step tables (GwStripeLayout, gw_internal.h; built by the host at gw_create: gw_api.cpp; state-major, so that only the
// stripes of the noise states this handle's layout has are staged).  In LDS, because the step's first decisions wait on two
// dependent lookups in them:
//   mi    u32[D][2]   {mult, ceil(65536/mult)} per sender
//   per state s:  ber0 f64[D]  sender d hearing the RRM in NEW state s      ber1 f64[D]  the RRM hearing sender d
//                 h1 u8[D]     state of sender j after the announcement (trans[j][RRM][s])
//                 r1 u8[D]     state of the RRM after one packet of sender d (trans[RRM][d][s])
//                 cls0 / cls1 u8[D]  decode certainty, same indexing as ber0 / ber1
// Read straight from HBM/L2 (issued before the window loop, consumed after it, so the latency is hidden):
//   h2    u8[S][D][D] state of sender j after the announcement AND >= 1 data packet of d (valid when hearing the
//                     same talker twice changes nothing more, which gw_create verifies: idem_states)

// Keep a wave-uniform constant in registers from here on: without this the compiler re-loads kernel
// arguments (s_load + s_waitcnt) at a dozen points of the step, each exposing the scalar-cache latency;
// pinned right after the state loads are issued, all of it hides under the HBM latency of those loads.
#define PIN_V(x) asm volatile("" : "+v"(x))
#define PIN_S(x) asm volatile("" : "+s"(x))

// DT > 0: compile-time device count, the qb byte record is held in registers (NWC 16-byte words);
// DT == 0: any device count, the record's bytes are read and written in memory.
// FEEDBACK: write the CounterTraffic interpreter's (obs, reward, done); the fused pendulum step (below) replaces them by
// the plant's.  now_out: the env's clock after the step (unchanged for a bad action), live_out: e < N.
// MODE 0: every fast form behind its run-time flag; 1: all validated at gw_create (FAST); 2: FAST and no env can reach the
// fast forms' validity limits during this launch (host-side bound on the simulated time).
// HALF: 32 envs per 64-lane wave (lanes 0..31): twice the waves for a batch too small to give every SIMD one (the
// pendulum step at 32 768 envs); the upper lanes idle through the walk and help in the plant's matrix-core rounds.
template <int DT, bool FEEDBACK, int MODE, bool HALF = false>
__device__ __forceinline__ void ct_step_sfx_body(const GwState& st, const GwDevConst& c,
                                                 const int32_t* __restrict__ device,
                                                 const int32_t* __restrict__ duration,
                                                 int32_t* __restrict__ obs,
                                                 float* __restrict__ reward,
                                                 uint8_t* __restrict__ done,
                                                 uint8_t* __restrict__ fb,          // one-byte feedback row (gw_step_fb) or null
                                                 int n_stage,                       // 16-byte chunks of the tables to stage
                                                 double& now_out, bool& live_out)
{
    constexpr bool PACKED = DT > 0;                      // the byte record is held in registers
    constexpr bool FAST = MODE >= 1, NOLIM = MODE == 2;
    constexpr int NWC = DT > 0 ? (2 * DT + 1 + 15) / 16 : 1;   // 16-byte words of the record
    const uint32_t N = (uint32_t)st.N;
    const uint32_t e = HALF ? blockIdx.x * 32u + (threadIdx.x & 31u) : blockIdx.x * 64u + threadIdx.x;   // 64-thread blocks
    const int D = DT > 0 ? DT : c.D;
    const int R = D + 1, RRM = D;
    constexpr int S = GW_MAX_NSTATES;
    const uint32_t RB = PACKED ? 16u * NWC : (uint32_t)st.RB;

    // ---- step tables -> LDS.  Their global loads are issued FIRST, the per-env state loads right behind
    //      them, and only then are the tables written to LDS, so that both HBM/L2 latencies overlap
    //      (vmcnt is in-order: waiting for the older table loads does not wait for the younger state loads).
    STAMP(0);
    constexpr int DM = DT > 0 ? DT : GW_MAX_DEVICES;
    constexpr GwStripeLayout LM(DM);
    const GwStripeLayout L(D);
    constexpr int MAX_CH = LM.staged_chunks(GW_MAX_NSTATES);
    const int n_ch = n_stage < MAX_CH ? n_stage : MAX_CH;    // 16-byte chunks: mi + the stripes of the states this handle has
    const int tid = threadIdx.x;
    constexpr int nthr = 64;                                             // the launchers' block size
    constexpr int PER_FULL = (MAX_CH + 63) / 64;                         // chunks per thread, all 16 states
    constexpr int PER = (DT > 0 && PER_FULL <= 11) ? PER_FULL : 1;
    constexpr bool one_pass = DT > 0 && PER_FULL <= 11;
    // (one pass: the LDS image is padded to whole rounds of 64 chunks and every lane loads and writes its chunk of every round
    //  unconditionally -- lanes past the staged prefix re-read chunk 0 (one broadcast request) into the padding; skipping
    //  such rounds with a wave-uniform test measured no better -- so that the staging is straight-line
    //  code: with a guard per chunk it was a chain of exec-mask regions, and the scalar loads of the remaining kernel
    //  arguments ended up behind the first wait for the tables instead of in front of it)
    __shared__ __attribute__((aligned(16))) uint8_t s_blob[one_pass ? PER * 64 * 16 : MAX_CH * 16];
    uint4 r_ch[PER];
    if (one_pass) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int ch = tid + i * nthr;
            r_ch[i] = ld<uint4>(st.blob, (uint32_t)(ch < n_ch ? ch : 0) << 4);
        }
    }

    Tally k = {0, 0, 0, 0, 0};
    uint32_t k_bad = 0;

    // ---- per-env state loads (before anything is stored) ---------------------------------------
    // Unconditional, from a clamped env index (lanes past N read env 0 and store nothing): as plain SSA values the
    // 16-byte records stay in registers.  Loaded under `if (live)` into pre-initialised variables they became stack
    // objects, and for some device counts the compiler then folded the byte selects below into dynamic addressing
    // of such an object and moved it to LDS -- D = 3 and D = 6 ran three times slower than D = 4.
    const bool live = e < N && (!HALF || threadIdx.x < 32u);
    const uint32_t el = live ? e : 0u;
    const uint32_t o16 = e << 4, o16l = el << 4;
    const uint32_t oq = e * RB, oql = el * RB;
    int d = device[el];
    int du = duration[el];
    uint4 ip = ld<uint4>(st.ip, o16l);
    const double2 tw = ld<double2>(st.tw, o16l);
    const uint4 tk = ld<uint4>(st.tk, o16l);
    uint4 qw[NWC];
#pragma unroll
    for (int w = 0; w < NWC; ++w) qw[w] = PACKED ? ld<uint4>(st.qb, oql + 16u * w) : make_uint4(0u, 0u, 0u, 0u);
    now_out = tw.x;
    live_out = live;
    if (one_pass) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int ch = tid + i * nthr;
            *reinterpret_cast<uint4*>(s_blob + ((uint32_t)ch << 4)) = r_ch[i];
        }
    } else {
        for (int i = tid; i < n_ch; i += nthr) *reinterpret_cast<uint4*>(s_blob + ((uint32_t)i << 4)) = ld<uint4>(st.blob, (uint32_t)i << 4);
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);

    // ---- constants -> registers, after the tables are in LDS: the wave has waited for the table loads anyway, the arguments
    //      arrived meanwhile, and no wait for them stands between the wave's start and its first vector loads --------------------------------
    StepMathT<FAST, NOLIM> m(c);
    double slot = c.slot, br = c.bit_rate, hd = c.hdr_dur, hdr_bits = c.hdr_bits, interval = c.counter_interval;
    double coded_factor = c.coded_factor, cls_limit = c.cls_limit, inv_interval_lo = c.inv_interval_lo, inv_slot_lo = c.inv_slot_lo;
    int pv = c.payload_value, cbound = c.counter_bound, max_duration = c.max_duration, dfactor = c.duration_factor;
    int mh = c.mac_hdr, base_b = c.mac_hdr + c.net_hdr, idem_i = c.idem_states, fast_ticks = c.fast_ticks;
    __builtin_amdgcn_sched_barrier(0);
    // (as SCALAR registers: a vector instruction takes one scalar operand, which is all these expressions need; pinned as
    //  vector registers each double cost two v_mov at the top of every wave -- 30 instructions of a wave that issues one per
    //  4-8 cycles)
    PIN_S(m.slot); PIN_S(m.inv_slot); PIN_S(m.fmod_limit); PIN_S(m.dr); PIN_S(m.rcp_dr); PIN_S(m.max_ber);
    if (!FAST) { PIN_S(m.fast_fmod); PIN_S(m.fast_div); PIN_S(m.fast_decide); }
    PIN_S(slot); PIN_S(br); PIN_S(hd); PIN_S(hdr_bits); PIN_S(interval); PIN_S(coded_factor); PIN_S(cls_limit); PIN_S(inv_interval_lo); PIN_S(inv_slot_lo);
    PIN_S(pv); PIN_S(cbound); PIN_S(max_duration); PIN_S(dfactor); PIN_S(mh); PIN_S(base_b); PIN_S(idem_i); PIN_S(fast_ticks);
    if (FEEDBACK) { PIN_S(obs); PIN_S(reward); PIN_S(done); PIN_S(fb); }   // the output pointers too (the kernel's only arguments that are not preloaded)
    __builtin_amdgcn_sched_barrier(0);

    // (the action is used unconditionally here, so that its loads stay at the top with the others: with every use inside
    //  `if (live)` the compiler sank them into that block, behind the waits above)
    PIN_V(d); PIN_V(du);
    PIN_V(ip.x); PIN_V(ip.y); PIN_V(ip.z); PIN_V(ip.w);   // (the breakpoints are first used in the window: keep their load up here too)
    const uint2* s_mi = reinterpret_cast<const uint2*>(s_blob + L.mi);
    const uint8_t* s_st = s_blob + L.s0;                // stripe of state s at s_st + s * L.stripe
    const uint8_t* g_h2 = st.blob + L.h2;               // HBM/L2: [s][j][d]
    const uint32_t STR = (uint32_t)L.stripe, DD = (uint32_t)D;
    auto t_ber0 = [&](uint32_t dd, uint32_t ss) { return *reinterpret_cast<const double*>(s_st + ss * STR + (uint32_t)L.ber0 + dd * 8u); };
    auto t_ber1 = [&](uint32_t dd, uint32_t ss) { return *reinterpret_cast<const double*>(s_st + ss * STR + (uint32_t)L.ber1 + dd * 8u); };
    auto t_h1 = [&](uint32_t j, uint32_t ss) { return (uint32_t)s_st[ss * STR + (uint32_t)L.h1 + j]; };
    auto t_r1 = [&](uint32_t dd, uint32_t ss) { return (uint32_t)s_st[ss * STR + (uint32_t)L.r1 + dd]; };
    auto t_cls0 = [&](uint32_t dd, uint32_t ss) { return (uint32_t)s_st[ss * STR + (uint32_t)L.cls0 + dd]; };
    auto t_cls1 = [&](uint32_t dd, uint32_t ss) { return (uint32_t)s_st[ss * STR + (uint32_t)L.cls1 + dd]; };
    auto t_h2 = [&](uint32_t j, uint32_t dd, uint32_t ss) { return (uint32_t)g_h2[(ss * DD + j) * DD + dd]; };

    if (live) {
        uint32_t rvm = tk.z;
#ifdef GW_STAMPS
        asm volatile("" : "+v"(rvm));         // diagnostic build: touch ip so that the stamp below sits after the state loads have landed
#endif
        STAMP(3);
        int32_t last_abs = (int32_t)(tk.w & 0x7fffffffu);
        uint32_t dn = tk.w >> 31;
        uint32_t fl = 0;

        if ((unsigned)d >= (unsigned)D || (unsigned)du >= (unsigned)max_duration) {
            // counter_traffic.py:147 asserts; a batched step cannot raise per env: flag + skip
            fl = GW_FLAG_BADACT;
            k_bad = 1;
            const int32_t latest = pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u));
            if (FEEDBACK) {
                obs[e] = latest + cbound;
                reward[e] = 0.0f;
                done[e] = (uint8_t)dn;
                if (fb) fb[e] = (uint8_t)((uint32_t)(((latest > 0) - (latest < 0)) + 1) | (10u << 2) | (dn << 7));
            }
        } else {
            const uint32_t bound = (uint32_t)cbound;
            const uint32_t base_bytes = (uint32_t)base_b;
            const bool idem = FAST || idem_i != 0;

            uint32_t len_d, s_d_old, s_r_old;
            if (PACKED) {
                if (NWC == 1) {
                    // The record's words as opaque register values first: selecting between struct fields by a run-time
                    // index is folded by the compiler into dynamic addressing of a stack copy of the record, which it then
                    // moves to LDS -- whenever the index range straddles a word (D = 3, 6: three times slower than D = 4).
                    uint32_t q0 = qw[0].x, q1 = qw[0].y, q2 = qw[0].z, q3 = qw[0].w;
                    asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
                    len_d = byte_of(q0, q1, q2, q3, (uint32_t)d);
                    s_d_old = byte_of(q0, q1, q2, q3, (uint32_t)(DT + d));
                } else {                                  // multi-word record: pick the word, then the byte, from registers
                    uint32_t lx = 0, ly = 0, lz = 0, lw = 0, sx = 0, sy = 0, sz = 0, sw = 0;
#pragma unroll
                    for (int w = 0; w < NWC; ++w) {
                        uint32_t q0 = qw[w].x, q1 = qw[w].y, q2 = qw[w].z, q3 = qw[w].w;
                        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));      // opaque: see the one-word case
                        const bool hl = ((uint32_t)d >> 4) == (uint32_t)w, hs = ((uint32_t)(DT + d) >> 4) == (uint32_t)w;
                        lx = hl ? q0 : lx; ly = hl ? q1 : ly; lz = hl ? q2 : lz; lw = hl ? q3 : lw;
                        sx = hs ? q0 : sx; sy = hs ? q1 : sy; sz = hs ? q2 : sz; sw = hs ? q3 : sw;
                    }
                    len_d = byte_of(lx, ly, lz, lw, (uint32_t)d & 15u);
                    s_d_old = byte_of(sx, sy, sz, sw, (uint32_t)(DT + d) & 15u);
                }
                s_r_old = (word_of(qw[(2 * DT) >> 4], ((2 * DT) >> 2) & 3) >> (((2 * DT) & 3) * 8)) & 0xffu;
            } else {
                len_d = st.qb[oq + (uint32_t)d];
                s_d_old = st.qb[oq + (uint32_t)(D + d)];
                s_r_old = st.qb[oq + (uint32_t)(2 * D)];
            }
            const double t_a = tw.x;
            double wake = tw.y;
            const uint32_t tau0 = tk.x, nbp = tk.y;
            GwBp bpc, bpp;
            bpc.t0 = ip.x; bpc.c0 = ip.y;
            bpp.t0 = ip.z; bpp.c0 = ip.w;
            const GwBp* hist = st.bph + ((size_t)e << 7);
            // what the addressed sender / the RRM become after hearing the RRM / sender d once
            const uint32_t s_d = t_h1((uint32_t)d, s_d_old);
            const uint32_t s_r1 = t_r1((uint32_t)d, s_r_old);
            const uint2 mi = s_mi[d];
            // ... and every other sender j: after the announcement (n1), and after >= 1 data packet of d too (n2)
            uint32_t nb[PACKED ? 16 * NWC : 1];
            uint32_t n1[PACKED ? DT : 1], n2[PACKED ? DT : 1], mlt[PACKED ? DT : 1];
            if (PACKED) {
#pragma unroll
                for (int b = 0; b < 16 * NWC; ++b) nb[b] = (word_of(qw[b >> 4], (b >> 2) & 3) >> ((b & 3) * 8)) & 0xffu;
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    n1[i] = t_h1((uint32_t)i, nb[DT + i]);
                    n2[i] = t_h2((uint32_t)i, (uint32_t)d, nb[DT + i]);
                    mlt[i] = s_mi[i].x;
                }
            }
            const double ber_a = t_ber0((uint32_t)d, s_d);
            const uint32_t cls_a = t_cls0((uint32_t)d, s_d);
            const double ber_x1 = t_ber1((uint32_t)d, s_r1);
            const uint32_t cls_x1 = t_cls1((uint32_t)d, s_r1);
            uint32_t s_r = s_r_old;
            const uint32_t mult_d = mi.x;
            const uint32_t inv16_d = mi.y;
            const bool cls_valid = NOLIM || t_a < cls_limit;

            STAMP(4);
            const int slots = du * dfactor;                               // counter_traffic.py:149

            // ---- A.1 / A.2: announcement ------------------------------------------------------
            const int Ld = ndigits(slots);
            const TxTimes an = tx_times(m, t_a, hd, m.over_rate((double)(Ld * 8)));
            k.tx++;
            STAMP(5);
            const bool granted = decode(m, cls_a, cls_valid, ber_a, an, br, hdr_bits,
                                        (double)(Ld * 8) * coded_factor, fl);
            const double t_r = an.t_e;
            const double t_end = t_r + (double)(slots + 1) * slot;       // simple_stack.py:557-558

            STAMP(6);
            // ---- A.3: window at sender d ------------------------------------------------------
            uint32_t tau = tau0;
            int n_data = 0;
            Tally kd = {0, 0, 0, 0, 0};                                   // appends/drops of sender d only

            // all counter ticks with wake < t (or <= t): counted in f64 four at a time (the running
            // sum w += dt is the reference's arithmetic), applied to d's queue length in one go
            auto ticks_to = [&](double t, bool inclusive) {
                uint32_t kk = 0;
                for (;;) {
                    const double w1 = wake + interval, w2 = w1 + interval, w3 = w2 + interval, w4 = w3 + interval;
                    const bool b0 = inclusive ? (wake <= t) : (wake < t);
                    const bool b1 = inclusive ? (w1 <= t) : (w1 < t);
                    const bool b2 = inclusive ? (w2 <= t) : (w2 < t);
                    const bool b3 = inclusive ? (w3 <= t) : (w3 < t);
                    // the tick times increase, so only the LAST counted one can fall exactly on t
                    const double last = b3 ? w3 : (b2 ? w2 : (b1 ? w1 : wake));
                    if (inclusive && b0 && last == t) fl |= GW_FLAG_TIE;
                    const uint32_t n = (uint32_t)b0 + (uint32_t)b1 + (uint32_t)b2 + (uint32_t)b3;   // monotone
                    wake = b3 ? w4 : (b2 ? w3 : (b1 ? w2 : (b0 ? w1 : wake)));
                    kk += n;
                    if (!b3) break;
                }
                tau += kk;
                len_d = gw_len_after_ticks(len_d, kk, mult_d, kd);
            };

            // the same in one jump (gw_fastmath.h: a floor division corrected by the exact FMA residual; exact, validated at
            // gw_create).  The jump's time-independent preconditions are established ONCE for the step (gw_tick_span_ok over
            // [wake, t_end]: one binade, constant increment, exact differences), so that a call is a dozen instructions; a lane
            // whose step does not qualify (the first 62 ms of an env, a binade end within reach, a rounding-tie interval) counts
            // with the running-sum loop above.  A data packet of the first steps after a reset lasts 2-4 ms = 2-4 ticks.
            double delta = 0.0;
            const bool span_ok = (FAST || fast_ticks) && gw_tick_span_ok(wake, t_end, interval, &delta);
            auto ticks_upto = [&](double t, bool inclusive) {
                uint32_t nj = 0;
                double wj = wake;
                bool tiej = false, sane = false;
                gw_tick_jump_lo(wake, t, delta, inv_interval_lo, inclusive, &nj, &wj, &tiej, &sane);
                if (span_ok && sane) {
                    wake = wj;
                    tau += nj;
                    if (tiej) fl |= GW_FLAG_TIE;
                    len_d = gw_len_after_ticks(len_d, nj, mult_d, kd);
                } else {
                    ticks_to(t, inclusive);
                }
            };

            if (granted) {
                const double total = (double)slots * slot;               // simple_stack.py:400
                const double stopw = t_r + total;                        // :401 (== timeout time :406)
                double cur = t_r;
                // ties at the window start: the MAC's process initialisation is URGENT, so it runs first
                ticks_upto(cur, false);
                // ---- the window loop, straight-line form (MI355X: one wave per SIMD issues one instruction per 4-8 cycles
                //      whatever it is, and the wave lasts as long as its busiest lane's 8-9 packets; in-kernel stamps put the
                //      general loop below at 945 + 974 cycles per packet of that lane, exec-mask bookkeeping around its
                //      per-packet decisions included).  A lane takes the straight line when every one of those decisions is known
                //      in advance for the whole step:
                //        * tick jumps apply (span_ok) and the exact fast forms of fmod and division hold up to t_end;
                //        * the RRM's decode outcome is certain by class, its noise state idempotent;
                //        * t_r >= 2 (t_end - t_r): every packet's stop - t_s is exact (Sterbenz), so the completion event
                //          t_s + (stop - t_s) IS stop -- no select, no `not t.completed` case (GW_FLAG_REFEXC);
                //      and while the queue neither runs empty nor needs the breakpoint ring (> 2 resets inside its span).  The
                //      loop is software-pipelined -- an iteration transmits the packet whose fit was established by the
                //      previous one, then sizes the next head -- has ONE exit condition and no branch inside; everything it
                //      does is unconditional, because a lane that leaves never runs another iteration.  GW_FLAG_CARRY can only
                //      be raised by a window's last packet and is tested once behind the loop; the tick jump's self-check is
                //      accumulated and raises GW_FLAG_INTERNAL (never expected: gw_fastmath.h).
                bool more = true;
                const double span = t_end - t_r;
                const bool straight = span_ok && mult_d != 0u && idem && cls_valid && cls_x1 != (uint32_t)GW_CLS_COMPUTE &&
                                      (FAST || (m.fast_fmod && m.fast_div)) && (NOLIM || t_end < m.fmod_limit) &&
                                      t_r >= span + span;
                uint32_t pops = 0;
                if (straight && len_d != 0u) {                            // (an empty queue waits for a tick: general loop)
                    auto head = [&](uint32_t len, uint32_t tk_now, bool& deep) {
                        const uint32_t age = __umul24(len + mult_d - 1u, inv16_d) >> 16;      // gw_ceil_div
                        const uint32_t ht = tk_now - age;                 // tick of the head packet
                        const bool older = ht < bpc.t0;
                        deep = older && ht < bpp.t0;                      // > 2 resets inside the queue's span
                        return base_bytes + gw_min_u32((older ? bpp.c0 : bpc.c0) + (ht - (older ? bpp.t0 : bpc.t0)), bound);
                    };
                    // (loop-carried VALUES only: a bool that lives across a divergent loop is a lane mask the compiler
                    //  re-merges with three scalar instructions per iteration)
                    bool deep = false;
                    uint32_t chk = 0;
                    uint32_t s = head(len_d, tau, deep);
                    bool go = !deep && (stopw - cur) > gw_fast_div((double)(s * 8u), m.dr, m.rcp_dr);   // :418-420
                    while (go) {
                        // ---- transmit the head packet: pop (:425), slot alignment (simple_stack.py:204), durations
                        //      (physical.py:244-279); the completion event fires at stop (see above)
                        const double pd = gw_fast_div((double)(((int)s - mh) * 8), m.dr, m.rcp_dr);
                        const double t_s = cur + (m.slot - gw_fast_fmod_lo(cur, m.slot, inv_slot_lo));
                        const double t_e = t_s + (hd + pd);
                        // ---- counter ticks up to and including t_e (older events than the MAC's resume)
                        uint32_t nj = 0;
                        double wj = wake;
                        bool tiej = false, sane = false;
                        gw_tick_jump_lo(wake, t_e, delta, inv_interval_lo, true, &nj, &wj, &tiej, &sane);
                        chk |= (sane ? 0u : (uint32_t)GW_FLAG_INTERNAL) | (tiej ? (uint32_t)GW_FLAG_TIE : 0u);
                        len_d = gw_min_u32(len_d - 1u + __umul24(nj, mult_d), (uint32_t)GW_QUEUE_CAP);
                        tau += nj;
                        wake = wj;
                        cur = t_e;
                        pops++;
                        // ---- the next head, and whether the loop goes on: window still open (else its timeout has been
                        //      processed), a packet there, no breakpoint-ring lookup, and it fits (messages.py:67-75, :418-420)
                        bool deep_n = false;
                        s = head(len_d, tau, deep_n);
                        go = cur < stopw && len_d != 0u && !deep_n && (stopw - cur) > gw_fast_div((double)(s * 8u), m.dr, m.rcp_dr);
                    }
                    // the general loop takes over where the window is still open and the straight line ended on something it
                    // does not model (queue ran empty, breakpoint ring)
                    (void)head(len_d, tau, deep);
                    more = cur < stopw && (len_d == 0u || deep);
                    fl |= chk | ((pops && !(cur < t_end)) ? (uint32_t)GW_FLAG_CARRY : 0u);
                }
                if (pops) {                                               // devices.py:163-168, counter_traffic.py:75-80
                    const bool okx = cls_x1 == (uint32_t)GW_CLS_OK;
                    k.pop += pops;
                    n_data += (int)pops;
                    s_r = s_r1;
                    k.deliv += okx ? pops : 0u;
                    rvm |= okx ? (1u << d) : 0u;
                    dn = (okx && pv == cbound) ? 1u : dn;
                }
                if (more)
                for (;;) {
                    if (len_d == 0) {                                     // :409-416
                        // (a silent sender, mult 0, never signals packet-added: the MAC waits for the timeout)
                        if (mult_d != 0u && wake < stopw) {
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
                    uint32_t cls_x = cls_x1;
                    if (idem) {
                        s_r = s_r1;
                    } else {                                              // not seen with f64 link powers; full tables in HBM
                        s_r = st.trans[(uint32_t)((RRM * R + d) * S) + s_r];
                        ber_x = st.ber[(uint32_t)((RRM * R + d) * S) + s_r];
                        cls_x = st.cls[(uint32_t)((RRM * R + d) * S) + s_r];
                    }
                    const bool ok = decode(m, cls_x, cls_valid, ber_x, x, br, hdr_bits,
                                           (double)(pay * 8) * coded_factor, fl);
                    // devices.py:163-168, counter_traffic.py:75-80 (as selects)
                    k.deliv += ok ? 1u : 0u;
                    rvm |= ok ? (1u << d) : 0u;
                    dn = (ok && pv == cbound) ? 1u : dn;
                    fl |= !(x.t_e < t_end) ? (uint32_t)GW_FLAG_CARRY : 0u;
                    ticks_upto(x.t_e, true);                              // ticks are older events than the MAC's resume
                    cur = x.t_e;
                    if (!(cur < stopw)) break;                            // window timeout already processed
                }
            }

            STAMP(7);
            // ---- A.5: remaining ticks up to the end of the step: up to 21 of them, counted in one jump
            //      (gw_fastmath.h; exact, validated at gw_create) or, where the jump declines, by the loop --------
            ticks_upto(t_end, true);
            STAMP(8);
            const uint32_t n_ticks = tau - tau0;
            k.app += kd.app;
            k.drop += kd.drop;

            // ---- every other sender saw the same n_ticks ticks; every other radio heard the
            //      announcement and, if any, d's data (simple_stack.py:130-157) ------------------
            auto heard_slow = [&](int j, uint32_t s0) {                   // exact count of hearings, tables in HBM
                uint32_t s = st.trans[(uint32_t)((j * R + RRM) * S) + s0];
                for (int n = 0; n < n_data; ++n) s = st.trans[(uint32_t)((j * R + d) * S) + s];
                return s;
            };
            if (PACKED) {
                const uint32_t heard_data = n_data ? 0xffffffffu : 0u;
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    Tally ki = {0, 0, 0, 0, 0};
                    const uint32_t li = gw_len_after_ticks(nb[i], n_ticks, mlt[i], ki);
                    // a bit-mask blend, not `n_data ? n2[i] : n1[i]`: with the flags compile-time constants the compiler turns the
                    // latter into ONE select between the two ARRAYS, which sends both to scratch memory
                    uint32_t si = (n2[i] & heard_data) | (n1[i] & ~heard_data);
                    if (!idem) si = heard_slow(i, nb[DT + i]);
                    if (i != d) { k.app += ki.app; k.drop += ki.drop; }   // d's own ticks were counted in the window
                    nb[i] = (i == d) ? len_d : li;
                    nb[DT + i] = (i == d) ? s_d : si;
                }
                nb[2 * DT] = s_r;
                uint4 o[NWC];
#pragma unroll
                for (int w = 0; w < NWC; ++w) {
                    const int b = 16 * w;
                    o[w].x = nb[b + 0] | (nb[b + 1] << 8) | (nb[b + 2] << 16) | (nb[b + 3] << 24);
                    o[w].y = nb[b + 4] | (nb[b + 5] << 8) | (nb[b + 6] << 16) | (nb[b + 7] << 24);
                    o[w].z = nb[b + 8] | (nb[b + 9] << 8) | (nb[b + 10] << 16) | (nb[b + 11] << 24);
                    o[w].w = nb[b + 12] | (nb[b + 13] << 8) | (nb[b + 14] << 16) | (nb[b + 15] << 24);
                }
                STAMP(9);
#pragma unroll
                for (int w = 0; w < NWC; ++w) st_(st.qb, oq + 16u * w, o[w]);
            } else {
                for (int i = 0; i < D; ++i) {
                    if (i == d) continue;
                    const uint32_t l0 = st.qb[oq + (uint32_t)i];
                    const uint32_t s0 = st.qb[oq + (uint32_t)(D + i)];
                    const uint32_t s1 = !idem ? heard_slow(i, s0)
                                              : (n_data ? t_h2((uint32_t)i, (uint32_t)d, s0) : t_h1((uint32_t)i, s0));
                    st.qb[oq + (uint32_t)i] = (uint8_t)gw_len_after_ticks(l0, n_ticks, s_mi[i].x, k);
                    if (s1 != s0) st.qb[oq + (uint32_t)(D + i)] = (uint8_t)s1;
                }
                st.qb[oq + (uint32_t)d] = (uint8_t)len_d;
                if (s_d != s_d_old) st.qb[oq + (uint32_t)(D + d)] = (uint8_t)s_d;
                if (s_r != s_r_old) st.qb[oq + (uint32_t)(2 * D)] = (uint8_t)s_r;
            }

            STAMP(10);
            // ---- interpreter feedback (counter_traffic.py:85-112, envs/core.py:142-153) -----------
            const int32_t latest = pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u));
            const int32_t abs_d = latest < 0 ? -latest : latest;
            int32_t r = last_abs - abs_d;
            last_abs = abs_d;
            r = r > 10 ? 10 : (r < -10 ? -10 : r);
            if (FEEDBACK) {
                st_(obs, e << 2, (int32_t)(latest + cbound));
                st_(reward, e << 2, (float)r);
                st_(done, e, (uint8_t)dn);
                // the same feedback in the exchange format of feedback_pack.hip: bits 0-1 sign(obs - bound) + 1, 2-6 reward + 10, 7 done
                if (fb) st_(fb, e, (uint8_t)((uint32_t)(((latest > 0) - (latest < 0)) + 1) | ((uint32_t)(r + 10) << 2) | (dn << 7)));
            }
            now_out = t_end;

            st_(st.tw, o16, make_double2(t_end, wake));
            st_(st.tk, o16, make_uint4(tau, nbp, rvm, (uint32_t)last_abs | (dn << 31)));   // whole records only
        }
        STAMP(11);
        // ---- per-env event counters (popped, delivered, bad, flags: the rest is derived from the state) ----
        publish_env_counters(st.sa, N, e, k.pop, k.deliv, k_bad, fl, 1u);
    }
    STAMP(12);
}

// (kernel arguments and the header in front of the `ip` records: GW_LEAD_PARAMS, hdr_state, hdr_const in ct_common.hip.h)
template <int DT, int MODE>
__global__ __launch_bounds__(64) void ct_step_sfx_kernel(GW_LEAD_PARAMS, int32_t* __restrict__ obs, float* __restrict__ reward,
                                                        uint8_t* __restrict__ done, uint8_t* __restrict__ fb)
{
    const int n_dev = (int)(dev_stage & 0xffu);
    const GwState st = hdr_state<DT>(ip, tw, tk, qb, n_envs, n_dev);
    const GwDevConst c = hdr_const<DT>(ip, n_dev);
    double now_new;
    bool live;
    ct_step_sfx_body<DT, true, MODE>(st, c, device, duration, obs, reward, done, fb, (int)(dev_stage >> 8), now_new, live);
}

// ---- BASELINE config 4: env.step() of the pendulum env in ONE launch ------------------------------------------------
// = the band-assignment step of the env's network (sensor + silent controller, D = 2) + OdePlant.updateState to the
// env's new clock (plants/core.py:38-49; builder-defined linear plant x <- A x + B u on the f64 matrix cores, see
// plant_mfma.hip for the operand mapping) + InvertedPendulumInterpreter's reading of the plant
// (envs/inverted_pendulum.py:27-57).  The step kernel is thread-per-env, the MFMA tile is 16 envs x 4 state
// components: the wave's 64 plant states go through an LDS transpose and four rounds of 16 envs each.
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int MODE, bool HALF>
__global__ __launch_bounds__(64) void pend_step_kernel(GW_LEAD_PARAMS, GwPlantDev p,
                                                       int32_t* __restrict__ obs, float* __restrict__ reward,
                                                       double* __restrict__ angle_deg)
{
    const int n_dev = (int)(dev_stage & 0xffu);
    const GwState st = hdr_state<2>(ip, tw, tk, qb, n_envs, n_dev);
    const GwDevConst c = hdr_const<2>(ip, n_dev);
    constexpr int EPW = HALF ? 32 : 64;                  // envs per wave
    constexpr int NR = EPW / 16;                         // matrix-core rounds of 16 envs each
    __shared__ double s_x[EPW][5];                       // [env of the wave][component], padded: conflict-free both ways
    const int lane = threadIdx.x;
    const int64_t e = (int64_t)blockIdx.x * EPW + (lane & (EPW - 1));
    const bool mine = e < p.N && lane < EPW;
    const int64_t el = e < p.N ? e : 0;
    // the plant's state, loaded before the step so that its latency hides behind the network walk
    const double2 x01 = *reinterpret_cast<const double2*>(p.x + el * 4);
    const double2 x23 = *reinterpret_cast<const double2*>(p.x + el * 4 + 2);
    const double u = p.u[el];
    const double tl = p.t_last[el];
    const unsigned long long nsub0 = p.nsub[el];
    __shared__ double s_pop[(GW_PLANT_KMAX / 4) * 64];    // MFMA A operand of every candidate group, by lane
#pragma unroll
    for (int grp = 0; grp < GW_PLANT_KMAX / 4; ++grp) s_pop[grp * 64 + lane] = p.Pop[grp * 64 + lane];
    __shared__ double s_q[(GW_PLANT_KMAX + 1) * 4];      // Q_k[g]: the accumulated input vector of k substeps
    for (int i = lane; i < (GW_PLANT_KMAX + 1) * 4; i += 64) s_q[i] = p.Qtab[i];

    double now_new;
    bool live;
    ct_step_sfx_body<2, false, MODE, HALF>(st, c, device, duration, nullptr, nullptr, nullptr, nullptr, (int)(dev_stage >> 8), now_new, live);
#ifdef GW_EXP_NO_EPILOGUE
    if (now_new >= 0.0) return;
#endif

    // substeps to take: n = round((now - last) / dt), nothing if time did not advance
    int n = 0;
    if (mine && live && now_new > tl) n = (int)llrint((now_new - tl) * p.inv_dt);
    STAMP(13);
    if (lane < EPW) { s_x[lane][0] = x01.x; s_x[lane][1] = x01.y; s_x[lane][2] = x23.x; s_x[lane][3] = x23.y; }
    __syncthreads();
    const int g = lane >> 4, col = lane & 15;
    // NR rounds of 16 envs, interleaved: the rounds' MFMA chains are independent, so they issue back to back
    double xg[NR], uq[NR];
    int nq[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int src = 16 * q + col;
        xg[q] = s_x[src][g];
        uq[q] = __shfl(u, src);
        nq[q] = __shfl(n, src);
    }
    int nmax = n;                                        // wave-wide maximum of the substep counts
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(nmax, o); nmax = t > nmax ? t : nmax; }
    nmax = __builtin_amdgcn_readfirstlane(nmax);         // the same in every lane: a scalar, so that the group loop below has a
                                                         // scalar trip count (as a vector value it became exec masking around MFMAs)
    STAMP(14);
#ifdef GW_EXP_NO_MFMA
    nmax = 0;
#endif
    while (nmax > 0) {                                   // one pass unless an env needs more than GW_PLANT_KMAX substeps
        const int cmax = nmax > GW_PLANT_KMAX ? GW_PLANT_KMAX : nmax;
        int chunk[NR], mygrp[NR];
        v4f64 acc[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            chunk[q] = nq[q] > GW_PLANT_KMAX ? GW_PLANT_KMAX : nq[q];
            mygrp[q] = (chunk[q] - 1) >> 2;              // the candidate group holding this env's substep count (-1: none)
            acc[q] = v4f64{0.0, 0.0, 0.0, 0.0};
        }
        // Every group's MFMA accumulates into the same registers, but an env's state enters only the MFMA of ITS group
        // (B operand zero elsewhere: exact), so the accumulator ends up holding the four candidates of that one group and a
        // single select per round remains.  (Selecting per group compiled to 3 nested exec-mask regions per group and
        // round: 2.4 us of branching around 0.8 us of MFMA.)
        const int ngrp = (cmax + 3) >> 2;                // scalar trip count; NOT unrolled: a per-group `if` made every
#pragma unroll 1                                         // group a merge point that moved all accumulators AGPR -> VGPR -> AGPR
        for (int grp = 0; grp < ngrp; ++grp) {           // candidates k0..k0+3, k0 = 4*grp + 1
            const double a_p = s_pop[grp * 64 + lane];
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const double b = (mygrp[q] == grp) ? xg[q] : 0.0;
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_p, b, acc[q], 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {                   // my candidate, then + Q_k u: rank one, one fused multiply-add per lane
            const int r = (chunk[q] - 1) & 3;
            const double lo = (r & 1) ? acc[q].y : acc[q].x, hi = (r & 1) ? acc[q].w : acc[q].z;
            const double pick = (r & 2) ? hi : lo;
            xg[q] = chunk[q] > 0 ? fma(s_q[chunk[q] * 4 + g], uq[q], pick) : xg[q];
            nq[q] -= chunk[q];
        }
        nmax -= cmax;
    }
    STAMP(15);
#pragma unroll
    for (int q = 0; q < NR; ++q) s_x[16 * q + col][g] = xg[q];
    __syncthreads();
    if (mine) {
        const double a0 = s_x[lane][0], a1 = s_x[lane][1], a2 = s_x[lane][2], a3 = s_x[lane][3];
        if (n > 0) {
            gwk::gw_store_wt(reinterpret_cast<double2*>(p.x + e * 4), make_double2(a0, a1));
            gwk::gw_store_wt(reinterpret_cast<double2*>(p.x + e * 4 + 2), make_double2(a2, a3));
            gwk::gw_store_wt(p.t_last + e, now_new);
            gwk::gw_store_wt(p.nsub + e, nsub0 + (unsigned long long)n);
        }
        const double deg = a2 * (180.0 / 3.141592653589793);        // envs/inverted_pendulum.py:27-57
        if (obs) gwk::gw_store_wt(obs + e, (int32_t)deg);
        if (reward) gwk::gw_store_wt(reward + e, (float)fabs(180.0 - deg));
        if (angle_deg) gwk::gw_store_wt(angle_deg + e, deg);
    }
}

// fresh env: counters 1 (counter_traffic.py:48) == breakpoint (tick 0, value 1); first tick at t = 0;
// queues empty; every radio at thermal noise (state 0)
__global__ void ct_init_sfx_kernel(GwState st)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint32_t)st.N) return;
    const uint32_t o16 = e << 4;
    st_(st.tw, o16, make_double2(st.cst->start_time, st.cst->start_time));
    st_(st.tk, o16, make_uint4(0u, 1u, 0u, 0u));          // tau 0, one breakpoint, nothing received
    st_(st.ip, o16, make_uint4(0u, 1u, 0u, 0u));          // newest breakpoint (tick 0, value 1)
    for (int b = 0; b < st.RB; ++b) st.qb[e * (uint32_t)st.RB + b] = 0;
    GwBp b0; b0.t0 = 0u; b0.c0 = 1u;
    st.bph[(size_t)e << 7] = b0;
    for (int w = 0; w < GW_SA_WORDS; ++w) st.sa[(size_t)w * st.N + e] = 0u;
    if (e == 0u) { st.sa[(size_t)4 * st.N] = 0u; st.sa[(size_t)4 * st.N + 1] = 0u; }
}

// counter_traffic.py:135-144 + :69-73 -- counters and interpreter only; time is NOT rewound.
// In the suffix encoding "counters <- 0" is a new breakpoint (tau, 0).
__global__ void ct_reset_sfx_kernel(GwState st, const uint8_t* __restrict__ mask, int32_t* __restrict__ obs)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint32_t)st.N) return;
    const uint32_t o16 = e << 4;
    uint32_t rvm;
    if (!mask || mask[e]) {
        uint4 tk = ld<uint4>(st.tk, o16);
        uint4 bp = ld<uint4>(st.ip, o16);          // {newest (t0, c0), second newest (t0, c0)}
        GwBp cur; cur.t0 = bp.x; cur.c0 = 0u;
        if (bp.x == tk.x) {                        // no tick since the newest breakpoint: overwrite it
            st.bph[((size_t)e << 7) + ((tk.y - 1u) & GW_RING_MASK)] = cur;
            bp.y = 0u;
        } else {
            bp.z = bp.x; bp.w = bp.y;              // old newest becomes second newest
            cur.t0 = tk.x;
            st.bph[((size_t)e << 7) + (tk.y & GW_RING_MASK)] = cur;
            tk.y += 1u; bp.x = cur.t0; bp.y = 0u;
        }
        tk.z = 0u; tk.w = 0u;                      // interpreter.reset(): receivedValues, lastAbs, done
        st_(st.tk, o16, tk);
        st_(st.ip, o16, bp);
        rvm = 0u;
    } else {
        rvm = ld<uint4>(st.tk, o16).z;
    }
    if (obs) obs[e] = st.cst->payload_value * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u)) + st.cst->counter_bound;
}

__global__ void ct_received_sfx_kernel(GwState st, int32_t* __restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int D = st.D;
    if (idx >= st.N * D) return;
    const uint32_t e = (uint32_t)(idx / D);
    const int i = (int)(idx - (int64_t)e * D);
    const uint32_t rvm = ld<uint4>(st.tk, e << 4).z;
    out[idx] = ((rvm >> i) & 1u) ? st.cst->payload_value : 0;
}

__global__ void ct_delivered_sfx_kernel(GwState st, uint32_t* __restrict__ out)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (uint32_t)st.N) out[e] = st.sa[(size_t)2 * e + 1];
}

// the sticky per-env GW_FLAG_* bits (and, in the explicit-queue mode, their per-wave OR) back to zero
__global__ void ct_clear_flags_kernel(GwState st)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < st.N) {
        if (st.sa) st.sa[(size_t)3 * st.N + e] = 0u;
        if (st.xc) st.xc[(size_t)e * 4 + 3] = 0u;
    }
    if (st.totals && e < st.n_slots) st.totals[(size_t)e * GW_T_COUNT + GW_T_FLAGS] = 0ull;
}

// The step kernel's arguments as the one block the runtime copies (same order and natural alignment as the kernel's
// parameter list: 88 bytes).  Launched through hipModuleLaunchKernel with this block instead of `<<< >>>`: the triple-chevron
// path looks the kernel up by its host address and copies twelve arguments one by one, 0.36 us more per launch on the host
// (tools/launch_floor.hip: 3.31 -> 2.95 us) -- and the host's enqueue rate is what limits the light phases of a rollout.
struct StepArgs {
    uint32_t* ip; double* tw; uint32_t* tk; uint8_t* qb; const int32_t* device; const int32_t* duration;
    uint32_t n_envs, dev_stage;
    int32_t* obs; float* reward; uint8_t* done; uint8_t* fb;
};
static_assert(sizeof(StepArgs) == 88, "StepArgs must mirror ct_step_sfx_kernel's parameter list");

template <int DT, int MODE>
hipFunction_t step_function()
{
    static hipFunction_t fn[16] = {};                  // per device: a module's functions belong to the device it was loaded on
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!fn[dev]) {
        hipFunction_t f = nullptr;
        if (hipGetFuncBySymbol(&f, reinterpret_cast<const void*>(&ct_step_sfx_kernel<DT, MODE>)) != hipSuccess) { (void)hipGetLastError(); f = nullptr; }
        fn[dev] = f;
    }
    return fn[dev];
}

template <int DT, int MODE>
void launch_mode(const GwState& st, unsigned grid, const int32_t* device, const int32_t* duration,
                 int32_t* obs, float* reward, uint8_t* done, uint8_t* fb, hipStream_t stream)
{
    const uint32_t dev_stage = (uint32_t)st.D | ((uint32_t)st.stage_chunks << 8);
    if (hipFunction_t f = step_function<DT, MODE>()) {
        StepArgs a = {st.ip, st.tw, st.tk, st.qb, device, duration, (uint32_t)st.N, dev_stage, obs, reward, done, fb};
        size_t size = sizeof a;
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        (void)hipModuleLaunchKernel(f, grid, 1, 1, 64, 1, 1, 0, stream, nullptr, extra);
        return;
    }
    hipLaunchKernelGGL((ct_step_sfx_kernel<DT, MODE>), dim3(grid), dim3(64), 0, stream, st.ip, st.tw, st.tk, st.qb, device, duration,
                       (uint32_t)st.N, dev_stage, obs, reward, done, fb);
}

template <int DT>
int launch(const GwState& st, const GwDevConst& cst, const int32_t* device, const int32_t* duration,
           int32_t* obs, float* reward, uint8_t* done, uint8_t* fb, void* stream, bool below_limits)
{
    const unsigned grid = (unsigned)((st.N + 63) / 64);  // the kernel's compile-time block size is 64
    // every exact fast form validated for this handle (gw_create): the instantiation without their fallbacks -- and, when
    // the host can rule out that any env reaches their validity limits in this launch, without the per-lane limit tests
    const bool fast = cst.fast_fmod && cst.fast_div && cst.fast_decide && cst.idem_states && cst.fast_ticks;
    switch (fast ? (below_limits ? 2 : 1) : 0) {
    case 2:  launch_mode<DT, 2>(st, grid, device, duration, obs, reward, done, fb, (hipStream_t)stream); break;
    case 1:  launch_mode<DT, 1>(st, grid, device, duration, obs, reward, done, fb, (hipStream_t)stream); break;
    default: launch_mode<DT, 0>(st, grid, device, duration, obs, reward, done, fb, (hipStream_t)stream); break;
    }
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}

inline int ok_or_ehip() { return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP; }

} // namespace

int gw_launch_step_sfx(const GwState& st, const GwDevConst& cst, const int32_t* device, const int32_t* duration,
                       int32_t* obs, float* reward, uint8_t* done, uint8_t* fb, void* stream, bool below_limits)
{
    switch (st.D) {
    case 2:  return launch<2>(st, cst, device, duration, obs, reward, done, fb, stream, below_limits);
    case 3:  return launch<3>(st, cst, device, duration, obs, reward, done, fb, stream, below_limits);
    case 4:  return launch<4>(st, cst, device, duration, obs, reward, done, fb, stream, below_limits);
    case 5:  return launch<5>(st, cst, device, duration, obs, reward, done, fb, stream, below_limits);
    case 6:  return launch<6>(st, cst, device, duration, obs, reward, done, fb, stream, below_limits);
    case 7:  return launch<7>(st, cst, device, duration, obs, reward, done, fb, stream, below_limits);
    case 8:  return launch<8>(st, cst, device, duration, obs, reward, done, fb, stream, below_limits);
    case 16: return launch<16>(st, cst, device, duration, obs, reward, done, fb, stream, below_limits);
    case 32: return launch<32>(st, cst, device, duration, obs, reward, done, fb, stream, below_limits);
    default: return launch<0>(st, cst, device, duration, obs, reward, done, fb, stream, below_limits);
    }
}

int gw_launch_init_sfx(const GwState& st, void* stream)
{
    const unsigned grid = (unsigned)((st.N + 255) / 256);
    hipLaunchKernelGGL(ct_init_sfx_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, st);
    return ok_or_ehip();
}

int gw_launch_reset_sfx(const GwState& st, const uint8_t* mask, int32_t* obs, void* stream)
{
    const unsigned grid = (unsigned)((st.N + 255) / 256);
    hipLaunchKernelGGL(ct_reset_sfx_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, st, mask, obs);
    return ok_or_ehip();
}

int gw_launch_delivered_sfx(const GwState& st, uint32_t* out, void* stream)
{
    hipLaunchKernelGGL(ct_delivered_sfx_kernel, dim3((unsigned)((st.N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, st, out);
    return ok_or_ehip();
}

int gw_launch_received_sfx(const GwState& st, int32_t* out, void* stream)
{
    const int64_t total = st.N * (int64_t)st.D;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(ct_received_sfx_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, st, out);
    return ok_or_ehip();
}

int gw_launch_pend_step(const GwState& st, const GwDevConst& cst, const GwPlantDev& p, const int32_t* device, const int32_t* duration,
                        int32_t* obs, float* reward, double* angle_deg, void* stream, bool below_limits)
{
    // 32 envs per wave while that still leaves SIMDs without a wave (1024 SIMDs: up to 32 768 envs), else 64
    const bool half = st.N <= 32 * 1024;
    const unsigned grid = (unsigned)((st.N + (half ? 31 : 63)) / (half ? 32 : 64));
    const bool fast = cst.fast_fmod && cst.fast_div && cst.fast_decide && cst.idem_states && cst.fast_ticks;
    const int mode = fast ? (below_limits ? 2 : 1) : 0;
#define GW_PEND(MODE_, HALF_) hipLaunchKernelGGL((pend_step_kernel<MODE_, HALF_>), dim3(grid), dim3(64), 0, (hipStream_t)stream, GW_LEAD_ARGS(st), p, obs, reward, angle_deg)
    if (half) { if (mode == 2) GW_PEND(2, true); else if (mode == 1) GW_PEND(1, true); else GW_PEND(0, true); }
    else      { if (mode == 2) GW_PEND(2, false); else if (mode == 1) GW_PEND(1, false); else GW_PEND(0, false); }
#undef GW_PEND
    return ok_or_ehip();
}

int gw_launch_clear_flags(const GwState& st, void* stream)
{
    const int64_t n = st.N > st.n_slots ? st.N : st.n_slots;
    hipLaunchKernelGGL(ct_clear_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, st);
    return ok_or_ehip();
}
