// ct_common.hip.h -- device helpers shared by the step kernels (explicit-ring, suffix, fused rollout).
// Every f64 expression follows the reference's operation order; compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include "gw_internal.h"
#include "gw_queue.h"
#include "gw_fastmath.h"

// In-kernel stamps: only in the diagnostic build (make STAMPS=1 -> libgymwipe_amd_stamps.so); the
// product library contains no stamp code.  Values go to a buffer nothing else reads.
#ifdef GW_STAMPS
#define STAMP(i)                                                                              \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long _t;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if ((threadIdx.x & 63) == 0)                                                          \
            st.stamps[(((size_t)blockIdx.x * ((blockDim.x + 63) >> 6)) + (threadIdx.x >> 6)) * 16 + (i)] = _t; \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

namespace gwk {

typedef GwTally Tally;

struct TxTimes { double t_s, t_h, t_e, stop; };

// The three f64 operations of the step that are expensive on the device, each with an exact fast
// form that gw_create validated for this configuration (gw_fastmath.h) and the plain form otherwise.
// FAST = every fast form was validated for this handle (the usual case): the flags are compile-time constants and the
// plain forms' code and the wave-uniform branches around them disappear from the instruction stream.
// NOLIM (with FAST) = the host guarantees that every env's clock stays below the fast forms' validity limits for the
// whole launch (gw_api.cpp keeps an upper bound of the simulated time): no per-lane limit tests either.
template <bool FAST, bool NOLIM = false>
struct StepMathT {
    double slot, inv_slot, fmod_limit, dr, rcp_dr, max_ber;
    int fast_fmod, fast_div, fast_decide;

    __device__ __forceinline__ explicit StepMathT(const GwDevConst& c)
        : slot(c.slot), inv_slot(c.inv_slot), fmod_limit(c.fmod_limit), dr(c.data_rate),
          rcp_dr(c.rcp_data_rate), max_ber(c.max_ber), fast_fmod(FAST ? 1 : c.fast_fmod), fast_div(FAST ? 1 : c.fast_div),
          fast_decide(FAST ? 1 : c.fast_decide) {}

    // t % slot                                                       simtools.py:53
    __device__ __forceinline__ double slot_rem(double t) const
    {
        if (FAST && NOLIM) return gw_fast_fmod(t, slot, inv_slot);
        return ((FAST || fast_fmod) && t < fmod_limit) ? gw_fast_fmod(t, slot, inv_slot) : fmod(t, slot);
    }
    // bits / dataRate                                                 physical.py:244-247, messages.py:67-75
    __device__ __forceinline__ double over_rate(double bits) const
    {
        return (FAST || fast_div) ? gw_fast_div(bits, dr, rcp_dr) : bits / dr;
    }
    // round(errSum)/totalBits <= maxCorrectableBer (banker's rounding) simple_stack.py:269-286
    __device__ __forceinline__ bool decodes(double err, double bits) const
    {
        return (FAST || fast_decide) ? (4.0 * rint(err) <= bits) : ((rint(err) / bits) <= max_ber);
    }
};
typedef StepMathT<false> StepMath;

// simple_stack.py:204 (next slot; a FULL slot when already aligned) +
// physical.py:244-279 (durations) + simtools.py:112-116 (events fire at now + (t - now))
template <class M>
__device__ __forceinline__ TxTimes tx_times(const M& m, double cur, double hd, double pd)
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
template <class M>
__device__ __forceinline__ bool receive(const M& m, double ber, const TxTimes& x, double bit_rate,
                                        double hdr_bits, double pay_bits, uint32_t& flags)
{
    double err = 0.0 + ber * (x.t_h - x.t_s) * bit_rate;
    if (!m.decodes(err, hdr_bits)) return false;
    const double seg = ber * (x.t_e - x.t_h) * bit_rate;
    if (!(x.t_e >= x.stop)) flags |= GW_FLAG_REFEXC;      // `not t.completed` -> KeyError in the reference
    err = (0.0 + seg) + seg;
    return m.decodes(err, pay_bits);
}

// Decode outcome of one reception: certain by class (gw_tables.cpp: decode_class), or the exact arithmetic
template <class M>
__device__ __forceinline__ bool decode(const M& m, uint32_t cls, bool cls_valid, double ber, const TxTimes& x,
                                       double br, double hdr_bits, double pay_bits, uint32_t& fl)
{
    if (!(x.t_e >= x.stop)) fl |= GW_FLAG_REFEXC;        // `not t.completed` -> KeyError in the reference
    if (cls_valid && cls != GW_CLS_COMPUTE) return cls == GW_CLS_OK;
    uint32_t dummy = 0;
    return receive(m, ber, x, br, hdr_bits, pay_bits, dummy);
}

// Write-through (sc1) store for per-env state and outputs.  A launch ends with the write-back of whatever its waves left
// dirty in the L2s, and nothing of the next launch starts before that.  Waves finish at very different times (the slowest
// lane gates a wave), so data written through as each wave ends is already on its way while the slow waves still run
// (tools/launch_floor.hip: 7.22 -> 6.87 us per launch for 4 MB written; the default step kernel: 6.7 -> 6.3 us per step).
template <class T>
__device__ __forceinline__ void gw_store_wt(T* ptr, const T& v)
{
#ifdef GW_EXP_PLAIN_STORES
    *ptr = v;
#else
    if constexpr (sizeof(T) == 16) {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        u32x4_ w; __builtin_memcpy(&w, &v, 16);
        // (s_nop: a store of more than 8 bytes reads its data registers for two more cycles, and the compiler cannot see
        // into the asm to keep the next write of them away, as it does for its own stores)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(w) : "memory");
    } else if constexpr (sizeof(T) == 8) {
        unsigned long long w; __builtin_memcpy(&w, &v, 8);
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(ptr), "v"(w) : "memory");
    } else if constexpr (sizeof(T) == 4) {
        unsigned w; __builtin_memcpy(&w, &v, 4);
        asm volatile("global_store_dword %0, %1, off sc1" ::"v"(ptr), "v"(w) : "memory");
    } else {
        static_assert(sizeof(T) == 1, "gw_store_wt: 1, 4, 8 or 16 bytes");
        unsigned w = (unsigned)*reinterpret_cast<const uint8_t*>(&v);
        asm volatile("global_store_byte %0, %1, off sc1" ::"v"(ptr), "v"(w) : "memory");
    }
#endif
}

// Per-env event counters of the default-mode kernels: fire-and-forget atomics, issued only by lanes that have something to
// add (a step without data pops nothing, flags are rare).  No load, no dependent store: nothing of it is on a wave's
// critical path, and a quiet step moves no counter bytes at all.
__device__ __forceinline__ void publish_env_counters(uint32_t* sa, uint32_t N, uint32_t e, uint32_t pop, uint32_t deliv,
                                                     uint32_t bad, uint32_t fl, uint32_t steps)
{
    if (pop)                                             // (delivered <= popped) one 64-bit atomic bumps both
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(sa) + e, (unsigned long long)pop | ((unsigned long long)deliv << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bad) __hip_atomic_fetch_add(sa + (size_t)2 * N + e, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (fl)  __hip_atomic_fetch_or(sa + (size_t)3 * N + e, fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e == 0u)                                         // the handle's env.step() count: one lane per launch
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(sa + (size_t)4 * N), (unsigned long long)steps,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- live-PHY helpers (ct_step_dyn.hip and the generic kernel's live-PHY instantiation) ----
// physical.py:25-58 (Eb/N0, Q approximation), :82-98 (dBm helpers), :208-212 (BPSK) with the device libm
__device__ __forceinline__ double ber_bpsk_dev(double sig_mw, double noise_mw, double ten_log_br)
{
    const double s = 10 * log10(sig_mw);
    const double n = 10 * log10(noise_mw);
    if (s <= n) return 0.5;
    const double ratio_db = s - n - ten_log_br;
    const double ratio = pow(10.0, ratio_db / 10);
    const double x = sqrt(2 * ratio);
    const double e = 2.718281828459045;
    const double sqrt2pi = 2.5066282746310002;
    return (1 - pow(e, -1.4 * x)) * pow(e, -(pow(x, 2.0) / 2)) / (1.135 * sqrt2pi * x);
}

// link power from -> to (mW) of env e: the env's own matrix (row = talker) or the handle's shared [from][to]
template <bool PER_ENV>
__device__ __forceinline__ double gw_link(const GwState& st, int R, int from, int to, uint32_t e)
{
    if (PER_ENV) return st.prx_env[((size_t)e * R + from) * gw_rp(R) + to];
    return st.prx_tab[from * R + to];
}
// phy._receivedPower of radio j of env e (rows of gw_rp(R) doubles)
__device__ __forceinline__ double& gw_rx(const GwState& st, int R, int j, int64_t e)
{
    return st.rxp[(size_t)e * gw_rp(R) + j];
}


// ---- the per-step kernels' arguments and the header in front of the `ip` records (gw_internal.h: gw_blob_header) ----
// Only what changes per call and the addresses of the wave's first loads are arguments -- under 100 bytes.  Everything else
// (the handle's GwDevConst and GwState, the tables) sits in the header of the `ip` allocation and is read from there by
// scalar loads.
//  * As leading scalar arguments the values below are PRELOADED into SGPRs by the command processor
//    (-amdgpu-kernarg-preload-count, Makefile; 14 dwords at most), so the table and state loads issue in the wave's first
//    cycles; read from a GwState in the kernel-argument segment they cost two scalar-cache round trips before the first
//    vector load could leave.  The constants are fetched under the shadow of those loads.
//  * On the host, a launch with both structs by value (1.2 KB) spent 3.6 us of its 5.5 us writing arguments into
//    device-visible memory (tools/launch_floor.hip).
#define GW_LEAD_PARAMS uint32_t* __restrict__ ip, double* __restrict__ tw, uint32_t* __restrict__ tk, uint8_t* __restrict__ qb, \
                       const int32_t* __restrict__ device, const int32_t* __restrict__ duration, uint32_t n_envs, uint32_t dev_stage
// dev_stage: sender count | chunks of the tables to stage << 8 (one argument: the preload window holds 14 dwords)
#define GW_LEAD_ARGS(st_) (st_).ip, (st_).tw, (st_).tk, (st_).qb, device, duration, (uint32_t)(st_).N, \
                          ((uint32_t)(st_).D | ((uint32_t)(st_).stage_chunks << 8))
// (read through the CONSTANT address space: nothing writes the header while a step kernel runs, and as constant-space
//  loads the reads become scalar loads that the compiler may issue anywhere -- in particular at the top of the wave)
// (the host pass of the compiler parses these device functions too and has no address spaces: plain types there)
#if defined(__HIP_DEVICE_COMPILE__)
#define GW_AS_CONST __attribute__((address_space(4)))
#else
#define GW_AS_CONST
#endif
template <int DT>
__device__ __forceinline__ GwDevConst hdr_const(const uint32_t* ip, int n_dev)
{
    const int D = DT > 0 ? DT : n_dev;
    const uint8_t* at = reinterpret_cast<const uint8_t*>(ip) - gw_blob_header(D) + gw_hdr_cst_off(D);
    return *(const GW_AS_CONST GwDevConst*)at;                                   // (only the fields the body uses are loaded)
}
// A pointer READ FROM MEMORY is a generic pointer to the compiler: every access through it becomes a FLAT instruction (both
// the LDS and the vector-memory counters, no scalar base + 32-bit offset addressing; the live-PHY kernel's row loads were all
// flat: in-kernel stamps put 48 of them at ~300 cycles apiece).  Through a cast to the global address space and back, address-
// space inference turns the accesses into global ones.  (Kernel ARGUMENTS are known to be global without this.)
template <class T>
__device__ __forceinline__ T* gw_as_global(T* p)
{
    return (T*)(__attribute__((address_space(1))) T*)p;
}
template <int DT>
__device__ __forceinline__ GwState hdr_state(uint32_t* ip, double* tw, uint32_t* tk, uint8_t* qb, uint32_t n_envs, int n_dev)
{
    const int D = DT > 0 ? DT : n_dev;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(ip) - gw_blob_header(D);
    GwState st = *(const GW_AS_CONST GwState*)(base + gw_hdr_st_off(D));
    st.ip = ip; st.tw = tw; st.tk = tk; st.qb = qb; st.N = (int64_t)n_envs; st.D = D;
    st.blob = base;
#define GW_G(m) st.m = gw_as_global(st.m)
    GW_G(bph); GW_G(sa); GW_G(trans); GW_G(ber); GW_G(cls); GW_G(ber2); GW_G(cls2); GW_G(rxp); GW_G(prx_tab); GW_G(pos_tab);
    GW_G(extra_tab); GW_G(prx_env); GW_G(pos_env); GW_G(bcache); GW_G(talk); GW_G(stamps); GW_G(cst);
#undef GW_G
    return st;
}

__device__ __forceinline__ int ndigits(int v)             // messages.py:51-52 len(str(value)), v >= 0
{
    return 1 + (v >= 10) + (v >= 100) + (v >= 1000) + (v >= 10000) + (v >= 100000) + (v >= 1000000) +
           (v >= 10000000) + (v >= 100000000) + (v >= 1000000000);
}


// Event totals of the generic kernel.  Each wave owns one 64-byte slot (GW_T_COUNT u64 words) in HBM; gw_stats_read sums the
// slots on the host.  What was measured on MI355X on the way here:
//   * eight global atomics per wave on ONE shared line serialised the whole launch (~90 us);
//   * eight 64-lane shuffle reductions + load/add/store of the private slot: 43% of a wave's cycles;
//   * seven words summed by LDS atomics (one ds_add per word, all 64 lanes on one address): the lanes serialise, 2 700 cycles;
//   * now: four DPP reductions (delivered, popped, appended, dropped), two population counts (steps, bad actions), the flags
//     only when a lane has one, and ONE no-return global atomic instruction by lanes 0..7 on the wave's own line.
// sum / OR over the wave's 64 lanes on the DPP path (result in lane 63): an inclusive scan within each row of 16 lanes
// (row_shr 1, 2, 4, 8; lanes shifted in from outside the row contribute 0), then the row totals carried across rows
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  Six VALU instructions per word -- the LDS atomics this
// replaces serialise over the lanes of a wave when they all hit one address: 7 words took 2 700 cycles (in-kernel stamps).
template <int CTRL, int ROW_MASK, bool OR>
__device__ __forceinline__ uint32_t gw_dpp_step(uint32_t v)
{
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
    return OR ? (v | o) : (v + o);
}
template <bool OR>
__device__ __forceinline__ uint32_t gw_wave_total(uint32_t v)
{
    v = gw_dpp_step<0x111, 0xf, OR>(v);
    v = gw_dpp_step<0x112, 0xf, OR>(v);
    v = gw_dpp_step<0x114, 0xf, OR>(v);
    v = gw_dpp_step<0x118, 0xf, OR>(v);
    v = gw_dpp_step<0x142, 0xa, OR>(v);
    v = gw_dpp_step<0x143, 0xc, OR>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// (called by every lane of the wave, converged)
__device__ __forceinline__ void publish_totals(unsigned long long* totals, const Tally& k, uint32_t k_steps,
                                               uint32_t k_bad, uint32_t fl_new)
{
    // every event count gets a full 32-bit word (a lane can pop hundreds of packets per step at a high bit rate);
    // steps, bad <= 64 each share one
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w_deliv = gw_wave_total<false>(k.deliv), w_pop = gw_wave_total<false>(k.pop);
    const uint32_t w_app = gw_wave_total<false>(k.app), w_drop = gw_wave_total<false>(k.drop);
    // steps and bad actions are 0 or 1 per lane: population counts of two lane masks; transmissions are a step's announcement
    // plus its data packets, lane by lane; the sticky flags are rarely set: their reduction runs only if some lane has one
    const uint32_t w_steps = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(k_steps != 0u));
    const uint32_t w_bad = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(k_bad != 0u));
    const uint32_t w_tx = w_steps + w_pop;
    uint32_t w_fl = 0u;
    if (__builtin_amdgcn_ballot_w64(fl_new != 0u)) w_fl = gw_wave_total<true>(fl_new);
    if (lane < (uint32_t)GW_T_COUNT) {
        uint32_t w = w_steps;                                         // GW_T_STEPS
        w = lane == (uint32_t)GW_T_TX ? w_tx : w;
        w = lane == (uint32_t)GW_T_DELIV ? w_deliv : w;
        w = lane == (uint32_t)GW_T_APP ? w_app : w;
        w = lane == (uint32_t)GW_T_POP ? w_pop : w;
        w = lane == (uint32_t)GW_T_DROP ? w_drop : w;
        w = lane == (uint32_t)GW_T_FLAGS ? w_fl : w;
        w = lane == (uint32_t)GW_T_BAD ? w_bad : w;
        const size_t waves_per_block = (blockDim.x + 63) >> 6;
        const size_t wave = (size_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6);
        unsigned long long* t = totals + wave * GW_T_COUNT;
        if (lane == GW_T_FLAGS) { if (w) atomicOr(&t[GW_T_FLAGS], (unsigned long long)w); }
        else if (w) atomicAdd(&t[lane], (unsigned long long)w);
    }
}

} // namespace gwk
