// ct_common.hip.h -- device helpers shared by the step kernels (explicit-ring and run-length).
// Every f64 expression follows the reference's operation order; compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include "gw_internal.h"
#include "gw_queue.h"
#include "gw_fastmath.h"

namespace gwk {

typedef GwTally Tally;

struct TxTimes { double t_s, t_h, t_e, stop; };

// The three f64 operations of the step that are expensive on the device, each with an exact fast
// form that gw_create validated for this configuration (gw_fastmath.h) and the plain form otherwise.
struct StepMath {
    double slot, inv_slot, fmod_limit, dr, rcp_dr, max_ber;
    int fast_fmod, fast_div, fast_decide;

    __device__ __forceinline__ explicit StepMath(const GwDevConst& c)
        : slot(c.slot), inv_slot(c.inv_slot), fmod_limit(c.fmod_limit), dr(c.data_rate),
          rcp_dr(c.rcp_data_rate), max_ber(c.max_ber), fast_fmod(c.fast_fmod), fast_div(c.fast_div),
          fast_decide(c.fast_decide) {}

    // t % slot                                                       simtools.py:53
    __device__ __forceinline__ double slot_rem(double t) const
    {
        return (fast_fmod && t < fmod_limit) ? gw_fast_fmod(t, slot, inv_slot) : fmod(t, slot);
    }
    // bits / dataRate                                                 physical.py:244-247, messages.py:67-75
    __device__ __forceinline__ double over_rate(double bits) const
    {
        return fast_div ? gw_fast_div(bits, dr, rcp_dr) : bits / dr;
    }
    // round(errSum)/totalBits <= maxCorrectableBer (banker's rounding) simple_stack.py:269-286
    __device__ __forceinline__ bool decodes(double err, double bits) const
    {
        return fast_decide ? (4.0 * rint(err) <= bits) : ((rint(err) / bits) <= max_ber);
    }
};

// simple_stack.py:204 (next slot; a FULL slot when already aligned) +
// physical.py:244-279 (durations) + simtools.py:112-116 (events fire at now + (t - now))
__device__ __forceinline__ TxTimes tx_times(const StepMath& m, double cur, double hd, double pd)
{
    TxTimes x;
    x.t_s = cur + (m.slot - m.slot_rem(cur));
    const double dur = hd + pd;
    x.stop = x.t_s + dur;
    const double th = x.t_s + hd;
    x.t_h = (th > x.t_s) ? x.t_s + (th - x.t_s) : x.t_s + 0.0;
    x.t_e = (x.stop > x.t_s) ? x.t_s + (x.stop - x.t_s) : x.t_s + 0.0;
    return x;
}

// simple_stack.py:214-286 with nothing else on the air: header decision at t_h, then the
// payload error sum counted twice from the same segment start (:180-188,:223-231,:252).
__device__ __forceinline__ bool receive(const StepMath& m, double ber, const TxTimes& x, double bit_rate,
                                        double hdr_bits, double pay_bits, uint32_t& flags)
{
    double err = 0.0 + ber * (x.t_h - x.t_s) * bit_rate;
    if (!m.decodes(err, hdr_bits)) return false;
    const double seg = ber * (x.t_e - x.t_h) * bit_rate;
    if (!(x.t_e >= x.stop)) flags |= GW_FLAG_REFEXC;      // `not t.completed` -> KeyError in the reference
    err = (0.0 + seg) + seg;
    return m.decodes(err, pay_bits);
}

__device__ __forceinline__ int ndigits(int v)             // messages.py:51-52 len(str(value))
{
    int n = 1;
    while (v >= 10) { v /= 10; ++n; }
    return n;
}


// reductions over the ACTIVE width of a wave (blocks narrower than 64 leave the upper lanes unborn)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v, int width = 64)
{
    for (int off = width >> 1; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v, int width = 64)
{
    for (int off = width >> 1; off > 0; off >>= 1) v |= __shfl_down(v, off, 64);
    return v;
}


// Event totals.  Each wave owns one 64-byte slot (GW_T_COUNT u64 words); gw_stats_read sums the
// slots on the host.  Measured on MI355X: eight atomics per wave on ONE shared line serialised the
// whole launch (~90 us); eight 64-lane reductions + a load/add/store of the private slot still cost
// 43% of the wave's cycles.  So: counters are packed into three lane words (per-lane values are
// small), reduced with three shuffles trees, and added with fire-and-forget atomics on the wave's
// own line (no contention, no load round trip).
__device__ __forceinline__ void publish_totals(unsigned long long* totals, const Tally& k, uint32_t k_steps,
                                               uint32_t k_bad, uint32_t fl_new)
{
    const int w = blockDim.x < 64 ? (int)blockDim.x : 64;
    // per lane: tx, deliv, pop <= ~64 -> 10-bit fields (wave sums < 2^16 each in 16-bit lanes of a u64)
    const unsigned long long a = (unsigned long long)k.tx | ((unsigned long long)k.deliv << 16) |
                                 ((unsigned long long)k.pop << 32) | ((unsigned long long)(k_steps | (k_bad << 8)) << 48);
    const unsigned long long b = (unsigned long long)k.app | ((unsigned long long)k.drop << 32);
    unsigned long long ra = a, rb = b;
    uint32_t rf = fl_new;
    for (int off = w >> 1; off > 0; off >>= 1) {
        ra += __shfl_down(ra, off, 64);
        rb += __shfl_down(rb, off, 64);
        rf |= __shfl_down(rf, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        const size_t waves_per_block = (blockDim.x + 63) >> 6;
        const size_t wave = (size_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6);
        unsigned long long* t = totals + wave * GW_T_COUNT;
        const unsigned long long sb = ra >> 48;
        atomicAdd(&t[GW_T_STEPS], sb & 0xffull);
        atomicAdd(&t[GW_T_TX], ra & 0xffffull);
        atomicAdd(&t[GW_T_APP], rb & 0xffffffffull);
        if ((ra >> 16) & 0xffffull) atomicAdd(&t[GW_T_DELIV], (ra >> 16) & 0xffffull);
        if ((ra >> 32) & 0xffffull) atomicAdd(&t[GW_T_POP], (ra >> 32) & 0xffffull);
        if (rb >> 32) atomicAdd(&t[GW_T_DROP], rb >> 32);
        if (sb >> 8) atomicAdd(&t[GW_T_BAD], sb >> 8);
        if (rf) atomicOr(&t[GW_T_FLAGS], (unsigned long long)rf);
    }
}

} // namespace gwk
