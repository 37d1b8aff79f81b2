// grid_phy.hip -- N replicas of the reference's PHY grid benchmark (tests/test_benchmark.py:20-91),
// the workload with concurrent transmissions and a real all-pairs interference sum.
//
// One lane per radio, one GROUP of GW lanes per replica (GW = the power of two >= n: 4, 8, 16, 32 or 64), 64 / GW replicas
// per wave: at n = 4 sixteen replicas share a wave and a replica's reductions are two shuffle stages instead of six
// (round 1 and 2 gave every replica a whole wave: 60 idle lanes at n = 4).  Every lane keeps ITS device's pending events
// (at most one of each kind) in registers; the next event of a replica is the group-wide minimum over
// (time, priority, insertion id) -- SimPy's heap order -- found with a width-GW shuffle butterfly.  Each group pops its
// own event, so the handler switch diverges across groups: one pass of the loop runs every handler some group needs and
// retires one event PER GROUP.  A handler runs with the group's lanes in parallel wherever the model loops over radios:
//     NOTIFY  (a transmission starts)  every other radio adds its received power  simple_stack.py:130-144
//     END     (it completes)           ... and subtracts it again                 :146-157
//     power change while receiving     count bit errors, re-evaluate the BER      :161-188,:223-231
// Insertion ids are handed out in the order the reference creates its events (lane order inside a
// loop over radios, via ballot + popcount), which is what decides who transmits first when several
// radios leave "wait until my reception ends" (:199-200) in the same instant -- the common case here.
//
// Events (one slot per kind and device), priority URGENT(0) before NORMAL(1) at equal times:
//   TICK   sender process wakes: SEND command to the PHY, next tick in 10 ms   tests/test_benchmark.py:33-48
//   HINIT  PHY handler starts (URGENT): wait for reception end or for the slot  simple_stack.py:192-204
//   RXFIN  the reception this handler waited for has ended                      :200, :267
//   SLOT   slot boundary: create the transmission                               physical.py:576-608
//   NOTIFY zero-delay "new transmission" notification                           physical.py:601-607
//   RXINIT a radio's receive process starts (URGENT)                            simple_stack.py:214-221
//   HDR    header end: receivers decide the header                              :238-248
//   END    transmission end: sender resumes, powers drop, payload decisions     :209-212, :146-157, :251-267
//   RXPROC / HPROC  a receive / handler process object completes (admission flags, queued commands)
//                                                                               simtools.py:322-392
#include <hip/hip_runtime.h>
#include <math.h>
#include "gw_internal.h"

namespace {

enum { EV_TICK = 0, EV_HINIT, EV_SLOT, EV_RXFIN, EV_HDR, EV_END, EV_NOTIFY, EV_RXINIT, EV_RXPROC, EV_HPROC, EV_MOVE, EV_COUNT };
static_assert(EV_COUNT == GW_GRID_EVENTS, "event slots");
constexpr uint32_t kNormal = 0x80000000u;       // key = priority bit | insertion id
constexpr double kInf = __builtin_inf();

// physical.py:25-58,82-98,208-212 with the device libm
__device__ __forceinline__ double ber_bpsk(double sig_mw, double noise_mw, double ten_log_br, double sqrt2pi)
{
    const double s = 10 * log10(sig_mw);
    const double n = 10 * log10(noise_mw);
    if (s <= n) return 0.5;
    const double ratio_db = s - n - ten_log_br;
    const double ratio = pow(10.0, ratio_db / 10);
    const double x = sqrt(2 * ratio);
    const double e = 2.718281828459045;
    return (1 - pow(e, -1.4 * x)) * pow(e, -(pow(x, 2.0) / 2)) / (1.135 * sqrt2pi * x);
}

// counter-based generator (the test tree evaluates the same function to reproduce a replica's walk)
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double grid_uniform(unsigned long long seed, unsigned long long replica, unsigned device,
                                               unsigned k, unsigned which)
{
    unsigned long long h = splitmix64(seed * 0x100000001B3ull + replica);
    h = splitmix64(h ^ (unsigned long long)(device * 0x9E3779B1ull + k));
    h = splitmix64(h ^ which);
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// FsplAttenuation._update + dbmToMilliwatts with the device libm (attenuation_models.py:28-36, physical.py:98)
__device__ __forceinline__ double fspl_power(double ax, double ay, double bx, double by, double tx_dbm, double twenty_log_f)
{
    const double dist = sqrt(pow(ax - bx, 2.0) + pow(ay - by, 2.0));
    const double att = 20 * log10(dist) + twenty_log_f - 147.55;
    return pow(10.0, (tx_dbm - att) / 10);
}

__device__ __forceinline__ int lanes_below(unsigned long long mask, int lane)
{
    return __popcll(mask & ((1ull << lane) - 1ull));
}

template <bool MOBILE, int GW>
__global__ __launch_bounds__(64) void grid_run_kernel(GwGridDev g, double seconds)
{
    constexpr int G = 64 / GW;                           // replicas per wave
    const int wl = threadIdx.x;                          // lane of the wave
    const int grp = wl / GW;
    const int lane = wl % GW;                            // this lane's radio within its replica
    const int gsh = grp * GW;                            // the group's first lane (ballot masks are shifted down by it)
    const unsigned long long gmask = GW == 64 ? ~0ull : ((1ull << GW) - 1ull);
    const int n = g.n;
    const int64_t env = (int64_t)blockIdx.x * G + grp;
    const bool have = env < g.N;                         // this group has a replica
    const bool me = have && lane < n;                    // this lane is a radio

    // ---- link tables -> LDS, one slice per group: static prx[from][to] (mW); mobile: the power each radio STORED for each
    // active transmission (simple_stack.py:78,136), plus the current positions
    extern __shared__ double s_all[];
    double* s_prx = s_all + (size_t)grp * (n * n + 2 * n);
    double* s_px = s_prx + n * n;
    double* s_py = s_px + n;
    if (have) {
        if (MOBILE) { for (int i = lane; i < n * n; i += GW) s_prx[i] = g.txp[env * n * n + i]; }
        else        { for (int i = lane; i < n * n; i += GW) s_prx[i] = g.prx[i]; }
    }

    // ---- state -> registers -----------------------------------------------------------------------------
    GwGridLane L;
    if (me) L = g.lanes[env * n + lane];
    else {
        for (int k = 0; k < EV_COUNT; ++k) { L.ev_t[k] = kInf; L.ev_k[k] = 0xffffffffu; }
        L.rx_power = 0; L.flags = 0; L.px = 0; L.py = 0; L.tx_on = 0;
        L.started = L.handler_running = L.transmitting = L.receiving = L.waiting_rx = L.rx_running = L.rx_phase = 0;
        L.queued = 0; L.rx_src = 0; L.rxi_src = 0;
    }
    if (MOBILE && me) { s_px[lane] = L.px; s_py[lane] = L.py; }
    __syncthreads();
    GwGridEnv E;                                         // group-uniform: now, eid, counters
    if (have) E = g.envs[env];
    else { E.now = 0.0; E.eid = 0; E.events = 0; E.n_tx = 0; E.first_run = 0; }
    double now = E.now;
    uint32_t eid = E.eid;
    uint32_t n_events = E.events, n_tx = E.n_tx;
    // SimMan.runSimulation(seconds): an URGENT stop event at now + seconds, created now (simtools.py:85-88)
    const double t_stop = now + seconds;
    const double t_stop_at = now + (t_stop - now);
    uint32_t k_stop;                                     // URGENT: priority bit clear
    if (E.first_run) {                                   // see grid_init_kernel
        k_stop = (uint32_t)(MOBILE ? 2 * n : n);
        eid = (uint32_t)(MOBILE ? 4 * n + 1 : 2 * n + 1);
        E.first_run = 0;
    }
    else k_stop = eid++;
    bool running = have;                                 // this group's replica has not reached its stop event yet

    const double slot = g.slot, interval = g.send_interval, br = g.bit_rate;
    const double hdr_bits = g.hdr_bits, pay_bits = g.pay_bits, hd = g.hdr_dur, pd = g.pay_dur;

    auto push = [&](int kind, double t, uint32_t key) { L.ev_t[kind] = t; L.ev_k[kind] = key; };
    // power change at a radio that is receiving: count errors from the (never advanced) segment start,
    // then re-evaluate the BER unless the received transmission is over (simple_stack.py:223-231)
    auto power_changed = [&]() {
        L.err_sum += L.ber * (now - L.t_seg) * br;
        if (!(now >= L.rx_stop)) {
            const double sig = s_prx[L.rx_src * n + lane];
            const double noise = L.rx_power - sig;
            if (!(noise >= 0)) L.flags |= GW_FLAG_REFEXC;            // the reference asserts here
            L.ber = ber_bpsk(sig, noise, g.ten_log_br, g.sqrt2pi);
        }
    };

    for (;;) {
        // ---- pop: minimum over (time, key) of every pending event of the group's lanes --------------------------
        double bt = kInf; uint32_t bk = 0xffffffffu; int bkind = 0;
#pragma unroll
        for (int k = 0; k < EV_COUNT; ++k) {
            const bool better = L.ev_t[k] < bt || (L.ev_t[k] == bt && L.ev_k[k] < bk);
            if (better) { bt = L.ev_t[k]; bk = L.ev_k[k]; bkind = k; }
        }
        int bwho = lane;
#pragma unroll
        for (int off = GW / 2; off > 0; off >>= 1) {
            const double ot = __shfl_xor(bt, off, GW);
            const uint32_t ok = __shfl_xor(bk, off, GW);
            const int okind = __shfl_xor(bkind, off, GW);
            const int owho = __shfl_xor(bwho, off, GW);
            if (ot < bt || (ot == bt && ok < bk)) { bt = ot; bk = ok; bkind = okind; bwho = owho; }
        }
        // the stop event: URGENT, created at the start of this run
        if (running && !(bt < t_stop_at || (bt == t_stop_at && bk < k_stop))) { now = t_stop_at; running = false; }
        if (running) {
            now = bt;
            n_events++;
            if (n_events - E.events > g.max_events) { if (me) L.flags |= GW_FLAG_CARRY; running = false; }   // every group reaches this: no hang
        }
        if (!__any(running)) break;                       // (wave-uniform: every lane leaves together)
        const int dev = bwho;                             // group-uniform
        const bool mine = me && lane == dev && running;
        if (mine) {
#pragma unroll
            for (int k = 0; k < EV_COUNT; ++k) if (k == bkind) { L.ev_t[k] = kInf; L.ev_k[k] = 0xffffffffu; }
        }
        if (!running) bkind = EV_COUNT;                   // a finished group runs no handler while the others go on

        switch (bkind) {
        case EV_TICK: {
            int inc = 0;
            if (mine) {
                if (L.started) {                          // tests/test_benchmark.py:36-48
                    L.n_sent++;
                    if (L.handler_running) L.queued++;
                    else { L.handler_running = 1; push(EV_HINIT, now, eid); inc = 1; }   // URGENT
                }
                L.started = 1;
                push(EV_TICK, now + interval, kNormal | (eid + inc));
                inc += 1;
            }
            eid += __shfl(inc, dev, GW);
            break;
        }
        case EV_HINIT: {
            int inc = 0;
            if (mine) {
                if (L.receiving) L.waiting_rx = 1;        // yield nReceivingFinished.event: nothing is scheduled
                else { L.transmitting = 1; push(EV_SLOT, now + (slot - fmod(now, slot)), kNormal | eid); inc = 1; }
            }
            eid += __shfl(inc, dev, GW);
            break;
        }
        case EV_RXFIN: {
            if (mine) { L.transmitting = 1; push(EV_SLOT, now + (slot - fmod(now, slot)), kNormal | eid); }
            eid += 1;
            break;
        }
        case EV_SLOT: {                                   // FrequencyBand.transmit + Transmission.__init__
            if (mine) {
                const double dur = hd + pd;
                L.tx_stop = now + dur;
                const double th = now + hd;
                push(EV_HDR, th > now ? now + (th - now) : now + 0.0, kNormal | eid);
                push(EV_END, L.tx_stop > now ? now + (L.tx_stop - now) : now + 0.0, kNormal | (eid + 1));
                push(EV_NOTIFY, now, kNormal | (eid + 2));
            }
            eid += 3;
            n_tx++;
            break;
        }
        case EV_NOTIFY: {
            const double stop_dev = __shfl(L.tx_stop, dev, GW);
            if (mine) L.tx_on = 1;
            if (me && lane != dev) {                      // _onNewTransmission at every other radio
                double p;
                if (MOBILE) {
                    p = fspl_power(L.px, L.py, s_px[dev], s_py[dev], g.tx_power_dbm, g.twenty_log_f);
                    s_prx[dev * n + lane] = p;            // _transmissionToReceivedPower[t]
                } else p = s_prx[dev * n + lane];
                L.rx_power = L.rx_power + p;
                if (L.receiving) power_changed();
            }
            // receive-process admission, blocking and not queued, in radio order (the sender too)
            const bool start = me && !L.rx_running;
            const unsigned long long m = (__ballot(start) >> gsh) & gmask;   // this group's lanes
            if (start) {
                L.rx_running = 1;
                L.rxi_src = dev;
                L.rxi_stop = stop_dev;
                push(EV_RXINIT, now, eid + lanes_below(m, lane));        // URGENT
            }
            eid += __popcll(m);
            break;
        }
        case EV_RXINIT: {
            int inc = 0;
            if (mine) {
                if (!L.transmitting) {                    // simple_stack.py:216-235
                    L.receiving = 1;
                    L.rx_src = L.rxi_src;
                    L.rx_stop = L.rxi_stop;
                    L.rx_phase = 0;
                    L.err_sum = 0.0;
                    L.t_seg = now;
                    const double sig = s_prx[L.rx_src * n + lane];
                    const double noise = L.rx_power - sig;
                    if (!(noise >= 0)) L.flags |= GW_FLAG_REFEXC;
                    L.ber = ber_bpsk(sig, noise, g.ten_log_br, g.sqrt2pi);
                } else {
                    push(EV_RXPROC, now, kNormal | eid);  // the generator ends at once
                    inc = 1;
                }
            }
            eid += __shfl(inc, dev, GW);
            break;
        }
        case EV_HDR: {                                    // receivers of `dev` in the header phase, radio order
            const bool rx = me && L.receiving && L.rx_src == dev && L.rx_phase == 0;
            bool fail = false;
            if (rx) {
                L.err_sum += L.ber * (now - L.t_seg) * br;
                if (4.0 * rint(L.err_sum) <= hdr_bits) {  // round(err)/bits <= 0.25, bits integral
                    L.hdr_ok++;
                    L.rx_phase = 1;
                    L.err_sum = 0.0;
                    L.t_seg = now;
                    const double sig = s_prx[dev * n + lane];
                    const double noise = L.rx_power - sig;
                    if (!(noise >= 0)) L.flags |= GW_FLAG_REFEXC;
                    L.ber = ber_bpsk(sig, noise, g.ten_log_br, g.sqrt2pi);
                } else {
                    L.hdr_fail++;
                    fail = true;
                }
            }
            // a failing receiver: [RXFIN if its handler waits] then RXPROC, lane by lane
            const int pushes = fail ? (L.waiting_rx ? 2 : 1) : 0;
            int before = pushes;                          // exclusive prefix sum over lanes
#pragma unroll
            for (int off = 1; off < GW; off <<= 1) { const int v = __shfl_up(before, off, GW); if (lane >= off) before += v; }
            const int total = __shfl(before, GW - 1, GW);
            before -= pushes;
            if (fail) {
                L.receiving = 0; L.err_sum = 0.0; L.ber = 0.0; L.t_seg = now;
                uint32_t e0 = eid + before;
                if (L.waiting_rx) { L.waiting_rx = 0; push(EV_RXFIN, now, kNormal | e0); e0++; }
                push(EV_RXPROC, now, kNormal | e0);
            }
            eid += total;
            break;
        }
        case EV_END: {
            // (1) the sender's handler resumes: cmd done event, then its process event
            if (mine) { L.transmitting = 0; L.tx_on = 0; push(EV_HPROC, now, kNormal | (eid + 1)); }
            eid += 2;
            // (2) every other radio: the power goes away; receivers re-integrate
            if (me && lane != dev) {
                const double p = s_prx[dev * n + lane];
                L.rx_power = L.rx_power + (-p);
                if (L.receiving) {
                    if (L.rx_src == dev && !(now >= L.rx_stop)) L.flags |= GW_FLAG_REFEXC;   // KeyError in the reference
                    power_changed();
                }
            }
            // (3) payload decisions of the radios that received this transmission, radio order
            const bool rx = me && L.receiving && L.rx_src == dev && L.rx_phase == 1;
            if (rx) {
                L.err_sum += L.ber * (now - L.t_seg) * br;               // counted a second time (:252)
                if (4.0 * rint(L.err_sum) <= pay_bits) L.pay_ok++; else L.pay_fail++;
            }
            const int pushes = rx ? (L.waiting_rx ? 2 : 1) : 0;
            int before = pushes;
#pragma unroll
            for (int off = 1; off < GW; off <<= 1) { const int v = __shfl_up(before, off, GW); if (lane >= off) before += v; }
            const int total = __shfl(before, GW - 1, GW);
            before -= pushes;
            if (rx) {
                L.receiving = 0; L.err_sum = 0.0; L.ber = 0.0; L.t_seg = now;
                uint32_t e0 = eid + before;
                if (L.waiting_rx) { L.waiting_rx = 0; push(EV_RXFIN, now, kNormal | e0); e0++; }
                push(EV_RXPROC, now, kNormal | e0);
            }
            eid += total;
            break;
        }
        case EV_RXPROC: {
            if (mine) L.rx_running = 0;
            break;
        }
        case EV_MOVE: {                                   // mover process: tests/test_benchmark.py:73-85
            if (MOBILE) {
                if (mine) {
                    const double xo = -g.move_span + (g.move_span - (-g.move_span)) * grid_uniform(g.seed, (unsigned long long)env, (unsigned)lane, L.move_k, 1u);
                    const double yo = -g.move_span + (g.move_span - (-g.move_span)) * grid_uniform(g.seed, (unsigned long long)env, (unsigned)lane, L.move_k, 2u);
                    L.px = L.px + xo;
                    L.py = L.py + yo;
                    L.move_k++;
                    s_px[lane] = L.px; s_py[lane] = L.py;
                    push(EV_MOVE, now + g.move_interval, kNormal | eid);
                }
                eid += 1;
                // (one wave per workgroup, LDS operations of a wave complete in order: the positions just written are what the
                //  group's other lanes read below; a workgroup barrier has no place inside a handler only some groups run)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const unsigned long long on_air = (__ballot(me && L.tx_on) >> gsh) & gmask;
                // (A) the mover's own transmission, heard by everyone else: _onAttenuationChange (simple_stack.py:119-128)
                if ((on_air >> dev) & 1ull) {
                    if (me && lane != dev) {
                        const double dx = L.px - s_px[dev], dy = L.py - s_py[dev];
                        if (sqrt(dx * dx + dy * dy) < 3000.0) {              // STANDBY_THRESHOLD, physical.py:371-386
                            const double np_ = fspl_power(L.px, L.py, s_px[dev], s_py[dev], g.tx_power_dbm, g.twenty_log_f);
                            const double delta = np_ - s_prx[dev * n + lane];
                            s_prx[dev * n + lane] = np_;
                            L.rx_power = L.rx_power + delta;
                            if (L.receiving && delta != 0) power_changed();
                        }
                    }
                }
                // (B) every other transmission on the air, as heard by the mover
                if (mine) {
                    unsigned long long rest = on_air & ~(1ull << dev);
                    while (rest) {
                        const int i = __ffsll((long long)rest) - 1;
                        rest &= rest - 1;
                        const double dx = L.px - s_px[i], dy = L.py - s_py[i];
                        if (sqrt(dx * dx + dy * dy) < 3000.0) {
                            const double np_ = fspl_power(s_px[i], s_py[i], L.px, L.py, g.tx_power_dbm, g.twenty_log_f);
                            const double delta = np_ - s_prx[i * n + lane];
                            s_prx[i * n + lane] = np_;
                            L.rx_power = L.rx_power + delta;
                            if (L.receiving && delta != 0) power_changed();
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            break;
        }
        case EV_HPROC: {                                  // queued SEND commands (simtools.py:369-381)
            int inc = 0;
            if (mine) {
                if (L.queued) { L.queued--; push(EV_HINIT, now, eid); inc = 1; }        // URGENT
                else L.handler_running = 0;
            }
            eid += __shfl(inc, dev, GW);
            break;
        }
        }
    }

    // ---- registers -> state -----------------------------------------------------------------------------
    __syncthreads();
    if (MOBILE && have) { for (int i = lane; i < n * n; i += GW) g.txp[env * n * n + i] = s_prx[i]; }
    if (me) g.lanes[env * n + lane] = L;
    if (have && lane == 0) {
        E.now = now; E.eid = eid; E.events = n_events; E.n_tx = n_tx;
        g.envs[env] = E;
    }
}

// Position.set() from the host: the (A)/(B) part of the MOVE handler at the replica's current time
__global__ __launch_bounds__(64) void grid_set_position_kernel(GwGridDev g, int dev, const double* __restrict__ xs, const double* __restrict__ ys)
{
    const int lane = threadIdx.x;
    const int n = g.n;
    const int64_t env = blockIdx.x;
    const bool me = lane < n;
    extern __shared__ double s_prx[];
    double* s_px = s_prx + n * n;
    double* s_py = s_px + n;
    for (int i = lane; i < n * n; i += 64) s_prx[i] = g.txp[env * n * n + i];
    GwGridLane L;
    if (me) L = g.lanes[env * n + lane];
    else { memset(&L, 0, sizeof L); }
    const double now = g.envs[env].now;
    if (me && lane == dev) { L.px = xs[env]; L.py = ys[env]; }
    if (me) { s_px[lane] = L.px; s_py[lane] = L.py; }
    __syncthreads();
    const double br = g.bit_rate;
    auto power_changed = [&]() {
        L.err_sum += L.ber * (now - L.t_seg) * br;
        if (!(now >= L.rx_stop)) {
            const double sig = s_prx[L.rx_src * n + lane];
            const double noise = L.rx_power - sig;
            if (!(noise >= 0)) L.flags |= GW_FLAG_REFEXC;
            L.ber = ber_bpsk(sig, noise, g.ten_log_br, g.sqrt2pi);
        }
    };
    const unsigned long long on_air = __ballot(me && L.tx_on);
    if (((on_air >> dev) & 1ull) && me && lane != dev) {
        const double dx = L.px - s_px[dev], dy = L.py - s_py[dev];
        if (sqrt(dx * dx + dy * dy) < 3000.0) {
            const double np_ = fspl_power(L.px, L.py, s_px[dev], s_py[dev], g.tx_power_dbm, g.twenty_log_f);
            const double delta = np_ - s_prx[dev * n + lane];
            s_prx[dev * n + lane] = np_;
            L.rx_power = L.rx_power + delta;
            if (L.receiving && delta != 0) power_changed();
        }
    }
    if (me && lane == dev) {
        unsigned long long rest = on_air & ~(1ull << dev);
        while (rest) {
            const int i = __ffsll((long long)rest) - 1;
            rest &= rest - 1;
            const double dx = L.px - s_px[i], dy = L.py - s_py[i];
            if (sqrt(dx * dx + dy * dy) < 3000.0) {
                const double np_ = fspl_power(s_px[i], s_py[i], L.px, L.py, g.tx_power_dbm, g.twenty_log_f);
                const double delta = np_ - s_prx[i * n + lane];
                s_prx[i * n + lane] = np_;
                L.rx_power = L.rx_power + delta;
                if (L.receiving && delta != 0) power_changed();
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < n * n; i += 64) g.txp[env * n * n + i] = s_prx[i];
    if (me) g.lanes[env * n + lane] = L;
}

__global__ void grid_init_kernel(GwGridDev g, const double* __restrict__ delays, const double* __restrict__ pos, double thermal)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.N * g.n) return;
    const int64_t env = idx / g.n;
    const int dev = (int)(idx - env * g.n);
    GwGridLane L;
    memset(&L, 0, sizeof L);
    for (int k = 0; k < EV_COUNT; ++k) { L.ev_t[k] = kInf; L.ev_k[k] = 0xffffffffu; }
    // insertion ids at start-up: process inits of device 0..n-1 take 0..n-1; run() is called before
    // they are processed, so the first stop event is id n; each init then yields timeout(initial_delay):
    // ids n+1 .. 2n.
    // mobile: sender inits 0..n-1, mover inits n..2n-1, stop 2n, delay timeouts 2n+1..3n, first moves 3n+1..4n
    L.ev_t[EV_TICK] = 0.0 + delays[idx];
    L.ev_k[EV_TICK] = kNormal | (uint32_t)((g.mobile ? 2 * g.n : g.n) + 1 + dev);
    if (g.mobile) {
        L.ev_t[EV_MOVE] = 0.0 + grid_uniform(g.seed, (unsigned long long)env, (unsigned)dev, 0u, 0u) * g.move_interval;
        L.ev_k[EV_MOVE] = kNormal | (uint32_t)(3 * g.n + 1 + dev);
    }
    L.px = pos[dev * 2]; L.py = pos[dev * 2 + 1];
    L.rx_power = thermal;
    g.lanes[idx] = L;
    if (dev == 0) {
        GwGridEnv E;
        memset(&E, 0, sizeof E);
        E.now = 0.0;
        E.eid = (uint32_t)((g.mobile ? 4 : 2) * g.n + 1);
        E.first_run = 1;                                 // the first run's stop event has id n (2n when mobile)
        g.envs[env] = E;
    }
}

} // namespace

int gw_grid_launch_run(const GwGridDev& g, double seconds, void* stream)
{
    // lanes per replica: the power of two that holds its radios; 64 / GW replicas share a wave
    int gw = g.n <= 4 ? 4 : (g.n <= 8 ? 8 : (g.n <= 16 ? 16 : (g.n <= 32 ? 32 : 64)));
    // ... unless that leaves SIMDs without a wave (1 024 of them): a small batch spreads out first (4 096 replicas of 4 radios:
    // groups of 16 lanes, four replicas per wave, 1 024 waves -- sixteen per wave would be 256 waves on a quarter of the chip)
    while (gw < 64 && (g.N * gw + 63) / 64 < 1024) gw *= 2;
    const int per_wave = 64 / gw;
    const size_t lds = (size_t)per_wave * ((size_t)g.n * g.n + 2 * (size_t)g.n) * sizeof(double);
    const unsigned grid = (unsigned)((g.N + per_wave - 1) / per_wave);
    hipStream_t s = (hipStream_t)stream;
#define GW_GRID(M_, W_) hipLaunchKernelGGL((grid_run_kernel<M_, W_>), dim3(grid), dim3(64), lds, s, g, seconds)
    if (g.mobile) { switch (gw) { case 4: GW_GRID(true, 4); break; case 8: GW_GRID(true, 8); break; case 16: GW_GRID(true, 16); break;
                                  case 32: GW_GRID(true, 32); break; default: GW_GRID(true, 64); break; } }
    else          { switch (gw) { case 4: GW_GRID(false, 4); break; case 8: GW_GRID(false, 8); break; case 16: GW_GRID(false, 16); break;
                                  case 32: GW_GRID(false, 32); break; default: GW_GRID(false, 64); break; } }
#undef GW_GRID
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}

int gw_grid_launch_set_position(const GwGridDev& g, int dev, const double* xs, const double* ys, void* stream)
{
    const size_t lds = ((size_t)g.n * g.n + 2 * (size_t)g.n) * sizeof(double);
    hipLaunchKernelGGL(grid_set_position_kernel, dim3((unsigned)g.N), dim3(64), lds, (hipStream_t)stream, g, dev, xs, ys);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}

int gw_grid_launch_init(const GwGridDev& g, const double* delays_dev, const double* pos_dev, double thermal, void* stream)
{
    const int64_t total = g.N * g.n;
    hipLaunchKernelGGL(grid_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, delays_dev, pos_dev, thermal);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}
