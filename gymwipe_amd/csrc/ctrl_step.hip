// ctrl_step.hip -- SURVEY 8f rank 2, second half: the pendulum env's control loop CLOSED over the receive-mode MACs.
//
// BUILDER-DEFINED (the reference never runs this loop: nobody sets `receiving`, and its env cannot be constructed);
// the specification this kernel is held to, bit for bit, is the event-driven ControlLoopModel of the test
// infrastructure.  Three network devices -- sensor (0), controller (1), actuator (2, not assignable) -- and the RRM:
//   * every counter tick the sensor queues the plant's angle for the controller, then the plant advances one
//     substep x <- A x + B u (plants/sliding_pendulum.py:116-135, plants/core.py:38-49);
//   * every `period` ticks from tick `start` on the controller queues -angle_deg for the actuator unless its angle is
//     0 (control/inverted_pendulum.py:52-69 with the shipped gains kp = 1, ki = kd = 0);
//   * a data packet decoded by its destination's receive-mode MAC is handed up: the controller takes degrees(value)
//     as its angle (:39-41), the actuator takes value as the motor velocity (sliding_pendulum.py:154-155);
//   * band assignment, announcement, window loop, slot alignment, BER integration, decode: exactly the step of
//     ct_step.hip, whose helpers are shared.
// One lane per environment; both MAC queues are explicit rings of payload values (packet sizes are constant); a step's
// appends are staged in LDS and flushed once, the addressed queue's head is prefetched.
// Event order at equal times follows the insertion rule: ticks before t first, then the delivery at t (its
// completion event was queued when the transmission started, more than a tick interval ago), then a tick at t.
#include "ct_common.hip.h"

using namespace gwk;

namespace {

constexpr int CR = 4, CRRM = 3;                             // radios (3 network devices + RRM), index of the RRM
constexpr int S = GW_MAX_NSTATES;

struct Lane {                                               // everything indexed by a run-time device id goes through
    double now, wake, x[4], u, ang;                         // selects below: a dynamically indexed member array would
    uint32_t ktick, got1, got2, ntx, ncmd, nsub, fl;        // send the whole struct to scratch memory
    uint32_t rxs[CR];
};

__device__ __forceinline__ uint32_t pick(const uint32_t (&a)[CR], int i)
{
    return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}

// Queue handling of one step.  The values a step appends (the sensor's sample of every tick: up to 22; the controller's
// command every `period` ticks) are staged in LDS, one column per lane, and written to the rings once, at the end of the
// step; the first PRE entries at the addressed queue's head are loaded before the walk.  A pop therefore never waits for
// memory inside the window loop (round 1: one dependent 8-byte load per transmission, one scattered 8-byte store per tick).
//   queue = the last `len` values of its append stream; this step's appends are the last `app` of it:
//   the head is a value of this step iff len <= app (then it is append number app - len).
constexpr int NEW0 = 24, NEW1 = 24, PRE = 8;

struct Q {                                                 // one queue during a step (scalars only: no dynamic indexing)
    uint32_t head, len, app, head0, tail0;
};

__device__ __forceinline__ void q_push(Q& q, double* s_col, double v)      // deque(maxlen=100): drop the oldest when full
{
    if (q.len == GW_QUEUE_CAP) { q.head = (q.head + 1u) & GW_RING_MASK; q.len--; }
    s_col[q.app << 6] = v;                                 // column of this lane: element j at [j * 64]
    q.app++;
    q.len++;
}

__global__ __launch_bounds__(64) void ctrl_step_kernel(GwCtrlDev c, const int32_t* __restrict__ device,
                                                       const int32_t* __restrict__ duration, int32_t* __restrict__ obs,
                                                       float* __restrict__ reward, double* __restrict__ angle_deg)
{
    __shared__ double s_new0[NEW0 * 64], s_new1[NEW1 * 64];
    // the noise-state transitions and the BER per (listener, talker, state): looked up for every radio at every transmission
    // (a window of the sensor's queue is dozens of them) -- from LDS, not a dependent global round trip apiece
    __shared__ uint8_t s_trans[CR * CR * S];
    __shared__ double s_ber[CR * CR * S];
    for (int i = threadIdx.x; i < CR * CR * S; i += blockDim.x) { s_trans[i] = c.trans[i]; s_ber[i] = c.ber[i]; }
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.N) return;
    double* col0 = s_new0 + (threadIdx.x & 63);
    double* col1 = s_new1 + (threadIdx.x & 63);
    // (the handle's constants by value through the constant address space: scalar loads the compiler may hoist and keep)
    const GwDevConst k = *(const GW_AS_CONST GwDevConst*)c.cst;
    Lane L;
    L.now = c.now[e]; L.wake = c.wake[e]; L.u = c.u[e]; L.ang = c.ang[e]; L.ktick = c.ktick[e];
    for (int i = 0; i < 4; ++i) L.x[i] = c.x[e * 4 + i];
    Q q0, q1;
    { const uint32_t hl = c.qhl[e]; q0.head = hl & 0xffu; q0.len = hl >> 8; }
    { const uint32_t hl = c.qhl[c.N + e]; q1.head = hl & 0xffu; q1.len = hl >> 8; }
    q0.app = q1.app = 0u;
    q0.head0 = q0.head; q1.head0 = q1.head;
    q0.tail0 = (q0.head + q0.len) & GW_RING_MASK; q1.tail0 = (q1.head + q1.len) & GW_RING_MASK;
#pragma unroll
    for (int j = 0; j < CR; ++j) L.rxs[j] = c.rxs[j * c.N + e];
    L.got1 = c.got[e * 2]; L.got2 = c.got[e * 2 + 1];
    L.ntx = c.ntx[e]; L.ncmd = c.ncmd[e]; L.nsub = c.nsub[e]; L.fl = c.flags[e];
    double* ring0 = c.pay + ((size_t)e * 2 + 0) * GW_RING_PHYS;
    double* ring1 = c.pay + ((size_t)e * 2 + 1) * GW_RING_PHYS;

    const int d = device[e], du = duration[e];
    const double deg_per_rad = 180.0 / 3.141592653589793;
    if ((unsigned)d >= 2u || (unsigned)du >= (unsigned)k.max_duration) {
        L.fl |= GW_FLAG_BADACT;                              // envs/inverted_pendulum.py:102 asserts; flagged and skipped here
    } else {
        const StepMath m(k);
        const double slot = k.slot, br = k.bit_rate, hd = k.hdr_dur, hdr_bits = k.hdr_bits, interval = k.counter_interval;
        const int mh = k.mac_hdr;
        double* ringd = d == 0 ? ring0 : ring1;
        // the first PRE values at the head of the addressed queue, issued now: they land during the announcement
        double pre[PRE];
        {
            const uint32_t h0 = d == 0 ? q0.head : q1.head;
#pragma unroll
            for (int i = 0; i < PRE; ++i) pre[i] = ringd[(h0 + (uint32_t)i) & GW_RING_MASK];
        }
        // the control tick test `(tick - start) % period == 0` as a running phase: one modulo per step, not per tick
        uint32_t phase = L.ktick >= c.start ? (L.ktick - c.start) % c.period : 0u;

        // one counter tick: the sensor's process was created first, so it runs first at every tick time
        auto tick_all = [&]() __attribute__((always_inline)) {
            q_push(q0, col0, L.x[2]);                                           // sensor: sample, queue, ...
            double nx[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {                                      // ... then the plant advances one substep
                double acc = c.A[i * 4 + 0] * L.x[0];
                acc = acc + c.A[i * 4 + 1] * L.x[1];
                acc = acc + c.A[i * 4 + 2] * L.x[2];
                acc = acc + c.A[i * 4 + 3] * L.x[3];
                nx[i] = acc + c.B[i] * L.u;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) L.x[i] = nx[i];
            L.nsub++;
            const uint32_t kk = L.ktick;                                        // controller
            if (kk >= c.start) {
                if (phase == 0u && L.ang != 0.0) {
                    q_push(q1, col1, -L.ang);
                    L.ncmd++;
                }
                phase = phase + 1u == c.period ? 0u : phase + 1u;
            }
            L.ktick = kk + 1u;
            L.wake = L.wake + interval;                                         // running sum, as everywhere
        };
        auto ticks_until = [&](double t, bool inclusive) __attribute__((always_inline)) {
            while (inclusive ? (L.wake <= t) : (L.wake < t)) {
                if (L.wake == t) L.fl |= GW_FLAG_TIE;
                tick_all();
            }
        };
        // every radio but `from` hears a transmission: its noise state moves (simple_stack.py:130-157)
        auto all_hear = [&](int from) {
#pragma unroll
            for (int j = 0; j < CR; ++j)
                if (j != from) L.rxs[j] = s_trans[(j * CR + from) * S + L.rxs[j]];   // j is a compile-time index here
        };

        const double t_a = L.now;
        const int slots = du * k.duration_factor;
        const int Ld = ndigits(slots);
        const TxTimes an = tx_times(m, t_a, hd, m.over_rate((double)(Ld * 8)));
        L.ntx++;
        all_hear(CRRM);
        const bool granted = receive(m, s_ber[(d * CR + CRRM) * S + pick(L.rxs, d)], an, br, hdr_bits,
                                     (double)(Ld * 8) * k.coded_factor, L.fl);
        const double t_r = an.t_e;
        const double t_end = t_r + (double)(slots + 1) * slot;
        if (granted) {
            const double stopw = t_r + (double)slots * slot;
            double cur = t_r;
            ticks_until(cur, false);                                            // the MAC's initialisation is URGENT: it goes first
            const int dst = d + 1;                                              // sensor -> controller, controller -> actuator
            const uint32_t sz = (uint32_t)(k.mac_hdr + k.net_hdr) + (d == 0 ? 2u : 1u);
            const double need = m.over_rate((double)(sz * 8u));
            const double pd = m.over_rate((double)(((int)sz - mh) * 8));
            for (;;) {
                bool closed = false;
                while ((d == 0 ? q0.len : q1.len) == 0u) {                      // wait for packet-added or the window timeout
                    if (L.wake < stopw) { cur = L.wake; tick_all(); }
                    else { closed = true; break; }
                }
                if (closed) break;
                if (!((stopw - cur) > need)) break;
                // the head value (a tick may have dropped older entries meanwhile: head/len say where the head is now)
                Q& qd = d == 0 ? q0 : q1;
                double v;
                if (qd.len <= qd.app) {                                         // appended in this step: from the LDS column
                    v = (d == 0 ? col0 : col1)[(qd.app - qd.len) << 6];
                } else {
                    const uint32_t i = (qd.head - qd.head0) & GW_RING_MASK;     // how many older entries have gone already
                    if (i < (uint32_t)PRE) {
                        v = pre[0];
#pragma unroll
                        for (int j = 1; j < PRE; ++j) v = (i == (uint32_t)j) ? pre[j] : v;
                    } else {
                        v = ringd[qd.head];
                    }
                }
                qd.head = (qd.head + 1u) & GW_RING_MASK;
                qd.len--;
                const TxTimes x = tx_times(m, cur, hd, pd);
                L.ntx++;
                all_hear(d);
                const bool ok = receive(m, s_ber[(dst * CR + d) * S + pick(L.rxs, dst)], x, br, hdr_bits,
                                        (double)(((int)sz - mh) * 8) * k.coded_factor, L.fl);
                if (!(x.t_e < t_end)) L.fl |= GW_FLAG_CARRY;
                ticks_until(x.t_e, false);                                      // ticks during the transmission
                if (ok) {                                                       // ... then the delivery at t_e ...
                    if (d == 0) { L.ang = v * deg_per_rad; L.got1++; }          // degrees(value), control/inverted_pendulum.py:41
                    else { L.u = v; L.got2++; }
                }
                ticks_until(x.t_e, true);                                       // ... then a tick falling exactly on it
                cur = x.t_e;
                if (!(cur < stopw)) break;
            }
        }
        ticks_until(t_end, true);
        L.now = t_end;
    }
    const double deg = L.x[2] * deg_per_rad;                                    // InvertedPendulumInterpreter, envs/inverted_pendulum.py:27-57
    obs[e] = (int32_t)deg;
    reward[e] = (float)fabs(180.0 - deg);
    if (angle_deg) angle_deg[e] = deg;

    // this step's appends -> the rings: value j of queue q sits in slot (tail0 + j) & mask
    for (uint32_t j = 0; j < q0.app; ++j) ring0[(q0.tail0 + j) & GW_RING_MASK] = col0[j << 6];
    for (uint32_t j = 0; j < q1.app; ++j) ring1[(q1.tail0 + j) & GW_RING_MASK] = col1[j << 6];

    c.now[e] = L.now; c.wake[e] = L.wake; c.u[e] = L.u; c.ang[e] = L.ang; c.ktick[e] = L.ktick;
    for (int i = 0; i < 4; ++i) c.x[e * 4 + i] = L.x[i];
    c.qhl[e] = (uint16_t)(q0.head | (q0.len << 8));
    c.qhl[c.N + e] = (uint16_t)(q1.head | (q1.len << 8));
#pragma unroll
    for (int j = 0; j < CR; ++j) c.rxs[j * c.N + e] = (uint8_t)L.rxs[j];
    c.got[e * 2] = L.got1; c.got[e * 2 + 1] = L.got2;
    c.ntx[e] = L.ntx; c.ncmd[e] = L.ncmd; c.nsub[e] = L.nsub; c.flags[e] = L.fl;
}

__global__ void ctrl_init_kernel(GwCtrlDev c, double x0, double x1, double x2, double x3, double u0)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.N) return;
    c.now[e] = 0.0; c.wake[e] = 0.0; c.u[e] = u0; c.ang[e] = 0.0; c.ktick[e] = 0u;
    c.x[e * 4 + 0] = x0; c.x[e * 4 + 1] = x1; c.x[e * 4 + 2] = x2; c.x[e * 4 + 3] = x3;
    c.qhl[e] = 0; c.qhl[c.N + e] = 0;
    for (int j = 0; j < CR; ++j) c.rxs[j * c.N + e] = 0;
    c.got[e * 2] = 0u; c.got[e * 2 + 1] = 0u;
    c.ntx[e] = 0u; c.ncmd[e] = 0u; c.nsub[e] = 0u; c.flags[e] = 0u;
}

} // namespace

int gw_ctrl_launch_step(const GwCtrlDev& c, const int32_t* device, const int32_t* duration, int32_t* obs, float* reward,
                        double* angle_deg, void* stream)
{
    hipLaunchKernelGGL(ctrl_step_kernel, dim3((unsigned)((c.N + 63) / 64)), dim3(64), 0, (hipStream_t)stream, c, device, duration,
                       obs, reward, angle_deg);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}

int gw_ctrl_launch_init(const GwCtrlDev& c, const double* x0, double u0, void* stream)
{
    hipLaunchKernelGGL(ctrl_init_kernel, dim3((unsigned)((c.N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c, x0[0], x0[1], x0[2],
                       x0[3], u0);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}
