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
// One lane per environment; both MAC queues are explicit rings of payload values (packet sizes are constant).
// Event order at equal times follows the insertion rule: ticks before t first, then the delivery at t (its
// completion event was queued when the transmission started, more than a tick interval ago), then a tick at t.
#include "ct_common.hip.h"

using namespace gwk;

namespace {

constexpr int CR = 4, CRRM = 3;                             // radios (3 network devices + RRM), index of the RRM
constexpr int S = GW_MAX_NSTATES;

struct Lane {                                               // everything indexed by a run-time device id goes through
    double now, wake, x[4], u, ang;                         // selects below: a dynamically indexed member array would
    uint32_t ktick, head[2], len[2], got1, got2, ntx, ncmd, nsub, fl;   // send the whole struct to scratch memory
    uint32_t rxs[CR];
};

__device__ __forceinline__ uint32_t pick(const uint32_t (&a)[CR], int i)
{
    return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}

__device__ __forceinline__ void ring_push(double* ring, uint32_t& head, uint32_t& len, double v)
{
    if (len == GW_QUEUE_CAP) { head = (head + 1u) & GW_RING_MASK; len--; }   // deque(maxlen=100): drop the oldest
    ring[(head + len) & GW_RING_MASK] = v;
    len++;
}

__global__ __launch_bounds__(64) void ctrl_step_kernel(GwCtrlDev c, const int32_t* __restrict__ device,
                                                       const int32_t* __restrict__ duration, int32_t* __restrict__ obs,
                                                       float* __restrict__ reward, double* __restrict__ angle_deg)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.N) return;
    const GwDevConst& k = *c.cst;
    Lane L;
    L.now = c.now[e]; L.wake = c.wake[e]; L.u = c.u[e]; L.ang = c.ang[e]; L.ktick = c.ktick[e];
    for (int i = 0; i < 4; ++i) L.x[i] = c.x[e * 4 + i];
#pragma unroll
    for (int q = 0; q < 2; ++q) { const uint32_t hl = c.qhl[q * c.N + e]; L.head[q] = hl & 0xffu; L.len[q] = hl >> 8; }
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

        // one counter tick: the sensor's process was created first, so it runs first at every tick time
        auto tick_all = [&]() {
            ring_push(ring0, L.head[0], L.len[0], L.x[2]);                     // sensor: sample, queue, ...
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
            if (kk >= c.start && (kk - c.start) % c.period == 0u && L.ang != 0.0) {
                ring_push(ring1, L.head[1], L.len[1], -L.ang);
                L.ncmd++;
            }
            L.ktick = kk + 1u;
            L.wake = L.wake + interval;                                         // running sum, as everywhere
        };
        auto ticks_until = [&](double t, bool inclusive) {
            while (inclusive ? (L.wake <= t) : (L.wake < t)) {
                if (L.wake == t) L.fl |= GW_FLAG_TIE;
                tick_all();
            }
        };
        // every radio but `from` hears a transmission: its noise state moves (simple_stack.py:130-157)
        auto all_hear = [&](int from) {
#pragma unroll
            for (int j = 0; j < CR; ++j)
                if (j != from) L.rxs[j] = c.trans[((size_t)j * CR + from) * S + L.rxs[j]];   // j is a compile-time index here
        };

        const double t_a = L.now;
        const int slots = du * k.duration_factor;
        const int Ld = ndigits(slots);
        const TxTimes an = tx_times(m, t_a, hd, m.over_rate((double)(Ld * 8)));
        L.ntx++;
        all_hear(CRRM);
        const bool granted = receive(m, c.ber[((size_t)d * CR + CRRM) * S + pick(L.rxs, d)], an, br, hdr_bits,
                                     (double)(Ld * 8) * k.coded_factor, L.fl);
        const double t_r = an.t_e;
        const double t_end = t_r + (double)(slots + 1) * slot;
        if (granted) {
            const double stopw = t_r + (double)slots * slot;
            double cur = t_r;
            ticks_until(cur, false);                                            // the MAC's initialisation is URGENT: it goes first
            const int dst = d + 1;                                              // sensor -> controller, controller -> actuator
            double* ring = d == 0 ? ring0 : ring1;
            const uint32_t sz = (uint32_t)(k.mac_hdr + k.net_hdr) + (d == 0 ? 2u : 1u);
            uint32_t hd_q = d == 0 ? L.head[0] : L.head[1];                    // d's queue in scalars for the loop
            for (;;) {
                bool closed = false;
                while ((d == 0 ? L.len[0] : L.len[1]) == 0u) {                  // wait for packet-added or the window timeout
                    if (L.wake < stopw) { cur = L.wake; tick_all(); }
                    else { closed = true; break; }
                }
                if (closed) break;
                const double need = m.over_rate((double)(sz * 8u));
                if (!((stopw - cur) > need)) break;
                hd_q = d == 0 ? L.head[0] : L.head[1];                          // (a tick may have dropped the oldest entry)
                const double v = ring[hd_q];
                hd_q = (hd_q + 1u) & GW_RING_MASK;
                if (d == 0) { L.head[0] = hd_q; L.len[0]--; } else { L.head[1] = hd_q; L.len[1]--; }
                const TxTimes x = tx_times(m, cur, hd, m.over_rate((double)(((int)sz - mh) * 8)));
                L.ntx++;
                all_hear(d);
                const bool ok = receive(m, c.ber[((size_t)dst * CR + d) * S + pick(L.rxs, dst)], x, br, hdr_bits,
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

    c.now[e] = L.now; c.wake[e] = L.wake; c.u[e] = L.u; c.ang[e] = L.ang; c.ktick[e] = L.ktick;
    for (int i = 0; i < 4; ++i) c.x[e * 4 + i] = L.x[i];
#pragma unroll
    for (int q = 0; q < 2; ++q) c.qhl[q * c.N + e] = (uint16_t)(L.head[q] | (L.len[q] << 8));
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
