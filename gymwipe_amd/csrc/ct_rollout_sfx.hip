// ct_rollout_sfx.hip -- gw_rollout(): K consecutive env.step() calls from pre-staged actions in ONE
// persistent launch, for the default (suffix) state layout.
//
// Two forms live here: the step-synchronous kernel (ct_rollout_sync_kernel: what gw_rollout launches since round 3, see its
// comment) and the event loop it replaced (ct_rollout_sfx_kernel with the packing / expanding kernels: GW_ROLLOUT_EVENT_LOOP=1,
// the A/B reference, kept under test), whose rationale follows.
// Why a second kernel: with one launch per step every wave lasts as long as its slowest env (0..9
// data transmissions per step, 1.6 on average), so ~80% of the lane-iterations of the window loop are
// idle.  Here a lane is not tied to the step boundary of its neighbours: the loop body is ONE
// TRANSMISSION (the announcement of a step or a data packet, same arithmetic), and a lane that
// finishes a step immediately starts its next one.  Over K steps the per-lane work evens out, the env
// state stays in registers, and the only per-step memory traffic is 2 bytes of action in and 1 byte of
// feedback out:
//     pack kernel    actions int32[K][N] x2  ->  u16[N][K]  (device | duration << 8)
//     rollout kernel state <-> registers once; feedback u8[N][K] = (diff+1) | (reward+10) << 2 | done << 7
//     expand kernel  u8[N][K] -> obs int32[K][N], reward f32[K][N], done u8[K][N]
// Results are bit-identical to K calls of the step kernel (tests compare both with the oracle).
// The step semantics and reference citations are those of ct_step_sfx.hip / ct_common.hip.h.
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
__device__ __forceinline__ uint32_t word_of(const uint4& w, int i)
{
    return i == 0 ? w.x : (i == 1 ? w.y : (i == 2 ? w.z : w.w));
}


// The rollout kernel wants each env's actions and feedback contiguous ([N][Kp]: one 16-byte load = 8 steps of
// actions, one dword store = 4 steps of feedback); the C-ABI takes and returns step-major arrays ([K][N]).  Both
// transposes go through an LDS tile of 64 envs x 64 steps so that every global access is a coalesced 16-byte
// (or, for `done`, 4-byte) access.  Pure streaming kernels, ~42 MB per 64 steps x 65 536 envs each.
constexpr int TP_ENVS = 64, TP_STEPS = 64;

// actions [K][N] int32 x 2  ->  [N][Kp] u16 (device | duration << 8; 0xffff = invalid or beyond K)
__global__ __launch_bounds__(256) void pack_actions_kernel(uint32_t N, int K, int Kp, const int32_t* __restrict__ device,
                                                           const int32_t* __restrict__ duration, uint16_t* __restrict__ packed)
{
    __shared__ __attribute__((aligned(16))) uint16_t tile[TP_ENVS][TP_STEPS + 2];                  // +2: rows start on different banks
    const uint32_t e0 = blockIdx.x * TP_ENVS;
    const int t = threadIdx.x;
    const bool vec_ok = (N & 3u) == 0;                                // rows of [K][N] stay 16-byte aligned
    for (int k0 = 0; k0 < Kp; k0 += TP_STEPS) {
        // load: thread -> (row r of 16, 4 consecutive envs)
        const int c4 = (t & 15) * 4, r = t >> 4;
#pragma unroll
        for (int pass = 0; pass < TP_STEPS / 16; ++pass) {
            const int kk = pass * 16 + r, k = k0 + kk;
            uint32_t v[4] = {0xffffu, 0xffffu, 0xffffu, 0xffffu};
            if (k < K) {
                int32_t dv[4], du[4];
                const uint32_t e = e0 + (uint32_t)c4;
                if (vec_ok && e + 3 < N) {
                    const int4 a = *reinterpret_cast<const int4*>(device + (size_t)k * N + e);
                    const int4 b = *reinterpret_cast<const int4*>(duration + (size_t)k * N + e);
                    dv[0] = a.x; dv[1] = a.y; dv[2] = a.z; dv[3] = a.w;
                    du[0] = b.x; du[1] = b.y; du[2] = b.z; du[3] = b.w;
                } else {
                    for (int j = 0; j < 4; ++j) {
                        const bool in = e + j < N;
                        dv[j] = in ? device[(size_t)k * N + e + j] : -1;
                        du[j] = in ? duration[(size_t)k * N + e + j] : -1;
                    }
                }
                // anything outside a byte is invalid for every configuration (D <= 32, max_duration checked by the caller)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] = ((uint32_t)dv[j] > 0xfeu || (uint32_t)du[j] > 0xfeu) ? 0xffffu : ((uint32_t)dv[j] | ((uint32_t)du[j] << 8));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[c4 + j][kk] = (uint16_t)v[j];
        }
        __syncthreads();
        // store: thread -> (env row, 8 consecutive steps = 16 bytes); 8 threads cover one env's 64 steps
        const int cols = (Kp - k0) < TP_STEPS ? (Kp - k0) : TP_STEPS;   // multiple of 16
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int env = pass * 32 + (t >> 3), s0 = (t & 7) * 8;
            if (e0 + env < N && s0 < cols) {
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (uint32_t)tile[env][s0 + 2 * j] | ((uint32_t)tile[env][s0 + 2 * j + 1] << 16);
                *reinterpret_cast<uint4*>(packed + (size_t)(e0 + env) * Kp + k0 + s0) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        __syncthreads();
    }
}

// feedback u8[N][Kp] -> obs/reward/done [K][N]
__global__ __launch_bounds__(256) void expand_feedback_kernel(uint32_t N, int K, int Kp, int center, int pv, const uint8_t* __restrict__ fb,
                                                              int32_t* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[TP_ENVS][TP_STEPS + 4];                   // rows 68 bytes apart: 4-byte aligned, banks spread
    const uint32_t e0 = blockIdx.x * TP_ENVS;
    const int t = threadIdx.x;
    const bool vec_ok = (N & 3u) == 0;
    for (int k0 = 0; k0 < Kp; k0 += TP_STEPS) {
        const int cols = (Kp - k0) < TP_STEPS ? (Kp - k0) : TP_STEPS;   // multiple of 16
        {   // load: thread -> (env row, 16 consecutive steps = 16 bytes); 4 threads cover one env's 64 steps
            const int env = t >> 2, s0 = (t & 3) * 16;
            if (e0 + env < N && s0 < cols) {
                const uint4 w = *reinterpret_cast<const uint4*>(fb + (size_t)(e0 + env) * Kp + k0 + s0);
                uint32_t* row = reinterpret_cast<uint32_t*>(&tile[env][s0]);
                row[0] = w.x; row[1] = w.y; row[2] = w.z; row[3] = w.w;
            }
        }
        __syncthreads();
        // store: thread -> (step row r of 16, 4 consecutive envs)
        const int c4 = (t & 15) * 4, r = t >> 4;
#pragma unroll
        for (int pass = 0; pass < TP_STEPS / 16; ++pass) {
            const int kk = pass * 16 + r, k = k0 + kk;
            const uint32_t e = e0 + (uint32_t)c4;
            if (k < K && e < N) {
                int32_t o[4]; float rw[4]; uint8_t dn[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t b = tile[c4 + j][kk];
                    o[j] = center + pv * ((int)(b & 3u) - 1);
                    rw[j] = (float)((int)((b >> 2) & 31u) - 10);
                    dn[j] = (uint8_t)(b >> 7);
                }
                if (vec_ok && e + 3 < N) {
                    *reinterpret_cast<int4*>(obs + (size_t)k * N + e) = make_int4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<float4*>(reward + (size_t)k * N + e) = make_float4(rw[0], rw[1], rw[2], rw[3]);
                    *reinterpret_cast<uchar4*>(done + (size_t)k * N + e) = make_uchar4(dn[0], dn[1], dn[2], dn[3]);
                } else {
                    for (int j = 0; j < 4 && e + j < N; ++j) {
                        obs[(size_t)k * N + e + j] = o[j]; reward[(size_t)k * N + e + j] = rw[j]; done[(size_t)k * N + e + j] = dn[j];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// Per-lane arrays of the rollout kernel.  DT > 0: registers (every loop over senders is unrolled, indices are static);
// DT == 0 (any sender count): one LDS column per lane -- the same source lines index either.
template <int N>
struct RegArr {
    uint32_t v[N];
    __device__ __forceinline__ uint32_t& operator[](int i) { return v[i]; }
};
struct LdsArr {
    uint32_t* p;                                         // this lane's column: element i at p[i * 64]
    __device__ __forceinline__ uint32_t& operator[](int i) const { return p[i << 6]; }
};
template <int N> struct ConstRegArr {
    uint32_t v[N];
    __device__ __forceinline__ uint32_t operator[](int i) const { return v[i]; }
};
struct ConstMemArr {                                     // wave-uniform index into the handle's constants in device memory
    const void* p; int shift16;
    __device__ __forceinline__ uint32_t operator[](int i) const
    {
        return shift16 ? (uint32_t)reinterpret_cast<const uint16_t*>(p)[i] : reinterpret_cast<const uint32_t*>(p)[i];
    }
};
template <bool GEN, int N> struct ArrSel { typedef RegArr<N> rw; typedef ConstRegArr<N> ro; };
template <int N> struct ArrSel<true, N> { typedef LdsArr rw; typedef ConstMemArr ro; };

// MODE as in ct_step_sfx.hip: 0 run-time flags, 1 every fast form validated (FAST), 2 FAST and no env can reach the fast
// forms' validity limits during this launch (host-side bound on the simulated time over all K steps).
template <int DT, int MODE>
__global__ __launch_bounds__(256) void ct_rollout_sfx_kernel(GwState st, GwDevConst c, int K, int Kp,
                                                            const uint16_t* __restrict__ actions,
                                                            uint8_t* __restrict__ feedback)
{
    constexpr bool GEN = DT == 0;                        // any sender count: per-lane arrays in LDS, launched with 64 threads
    constexpr int DM = GEN ? GW_MAX_DEVICES : DT;        // capacity
    constexpr int NWC = (2 * DM + 1 + 15) / 16;
    constexpr int S = GW_MAX_NSTATES;
    const int D = GEN ? c.D : DT, R = D + 1, RRM = D;
    const uint32_t N = (uint32_t)st.N;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;

    // ---- lookup tables -> LDS ----------------------------------------------------------------------
    constexpr int TRANS_B = ((DM + 1) * (DM + 1) * S + 15) / 16 * 16;
    __shared__ __attribute__((aligned(16))) uint8_t s_trans[TRANS_B];
    __shared__ __attribute__((aligned(16))) double  s_ber[2 * DM * S];
    __shared__ __attribute__((aligned(16))) uint8_t s_cls[2 * DM * S];
    __shared__ uint32_t s_cols[GEN ? (3 * DM + 1) * 64 : 1];     // GEN: len[D], tb[D], sta[R] columns per lane
    __shared__ uint2 s_mi[GEN ? DM : 1];                 // GEN: {mult, ceil(65536/mult)} and terminal-state masks, indexed by
    __shared__ uint32_t s_term[GEN ? DM : 1];            //      the lane's own addressed sender
    if constexpr (GEN) {
        for (int i = threadIdx.x; i < D; i += blockDim.x) {
            s_mi[i] = make_uint2((uint32_t)st.cst->mult[i], st.cst->inv16[i]);
            s_term[i] = st.cst->term[i];
        }
    }
    {
        const int n_tr = (R * R * S + 15) >> 4, n_be = (2 * D * S * 8) >> 4, n_cl = (2 * D * S) >> 4;
        for (int i = threadIdx.x; i < n_tr; i += blockDim.x) *reinterpret_cast<uint4*>(s_trans + ((uint32_t)i << 4)) = ld<uint4>(st.trans, (uint32_t)i << 4);
        for (int i = threadIdx.x; i < n_be; i += blockDim.x) *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(s_ber) + ((uint32_t)i << 4)) = ld<uint4>(st.ber2, (uint32_t)i << 4);
        for (int i = threadIdx.x; i < n_cl; i += blockDim.x) *reinterpret_cast<uint4*>(s_cls + ((uint32_t)i << 4)) = ld<uint4>(st.cls2, (uint32_t)i << 4);
    }
    __syncthreads();
    if (e >= N) return;

    // ---- state -> registers --------------------------------------------------------------------------
    const uint32_t RB = GEN ? (uint32_t)st.RB : 16u * NWC;
    const uint32_t o16 = e << 4, oq = e * RB;
    const uint4 ip = ld<uint4>(st.ip, o16);
    const double2 tw = ld<double2>(st.tw, o16);
    const uint4 tk = ld<uint4>(st.tk, o16);
    typename ArrSel<GEN, DM>::rw len, tb;
    typename ArrSel<GEN, DM + 1>::rw sta;
    if constexpr (GEN) {
        uint32_t* col = s_cols + (threadIdx.x & 63);
        len.p = col; tb.p = col + DM * 64; sta.p = col + 2 * DM * 64;
        for (int i = 0; i < D; ++i) len[i] = st.qb[oq + (uint32_t)i];
        for (int j = 0; j < R; ++j) sta[j] = st.qb[oq + (uint32_t)(D + j)];
    } else {
        uint4 qw[NWC];
#pragma unroll
        for (int w = 0; w < NWC; ++w) qw[w] = ld<uint4>(st.qb, oq + 16u * w);
#pragma unroll
        for (int i = 0; i < DT; ++i) len[i] = (word_of(qw[i >> 4], (i >> 2) & 3) >> ((i & 3) * 8)) & 0xffu;
#pragma unroll
        for (int j = 0; j < DT + 1; ++j) sta[j] = (word_of(qw[(DT + j) >> 4], ((DT + j) >> 2) & 3) >> (((DT + j) & 3) * 8)) & 0xffu;
    }
    double now = tw.x, wake = tw.y;
    uint32_t tau = tk.x;
    const uint32_t nbp = tk.y;
    GwBp bpc, bpp;
    bpc.t0 = ip.x; bpc.c0 = ip.y;                // (record layout: ct_step_sfx.hip)
    bpp.t0 = ip.z; bpp.c0 = ip.w;
    const GwBp* hist = st.bph + ((size_t)e << 7);
    uint32_t rvm = tk.z;
    int32_t last_abs = (int32_t)(tk.w & 0x7fffffffu);
    uint32_t dn = tk.w >> 31;

    constexpr bool FAST = MODE >= 1, NOLIM = MODE == 2;
    const StepMathT<FAST, NOLIM> m(c);
    const double slot = c.slot, br = c.bit_rate, hd = c.hdr_dur, hdr_bits = c.hdr_bits, interval = c.counter_interval;
    const double inv_interval = c.inv_interval, tie_filter = c.tie_filter;
    const bool fast_ticks = FAST || c.fast_ticks != 0;
    const uint32_t bound = (uint32_t)c.counter_bound, base_bytes = (uint32_t)(c.mac_hdr + c.net_hdr);
    const int mh = c.mac_hdr, pv = c.payload_value;
    uint32_t live_mask = 0u;                             // any-D kernel: bit i = sender i is in a non-terminal noise state
    if constexpr (GEN) {
        for (int i = 0; i < D; ++i) live_mask |= ((s_term[i] >> sta[i]) & 1u) ? 0u : (1u << i);
    }
    typename ArrSel<GEN, DM>::ro mult, term, inv16;
    if constexpr (GEN) {
        mult.p = st.cst->mult; mult.shift16 = 0; term.p = st.cst->term; term.shift16 = 1; inv16.p = st.cst->inv16; inv16.shift16 = 0;
    } else {
#pragma unroll
        for (int i = 0; i < DT; ++i) { mult.v[i] = (uint32_t)c.mult[i]; term.v[i] = c.term[i]; inv16.v[i] = c.inv16[i]; }
    }

    Tally kt = {0, 0, 0, 0, 0};
    uint32_t k_bad = 0, fl = 0;

    // ---- per-step variables of the lane's current step -------------------------------------------------
    // Laziness that keeps the loop body small (all exact):
    //   * queue lengths: len[i] is valid as of tick tb[i]; a sender is brought up to date when it is
    //     addressed (and everyone once at the end): k ticks are min(len + k*mult, 100) in one go;
    //   * counter ticks between the end of a window and the end of the step are not counted at the
    //     step end but by the next step's first count (tick counting is cumulative in time).
    int k = 0;                      // step index
    bool data_mode = false;         // false: the next transmission is the announcement of step k
    bool finish = false;            // the current step is over: close it at the top of the next iteration
    int d = 0;                      // addressed sender of the current step
    uint32_t len_d = 0, mult_d = 0, inv16_d = 65536u;
#pragma unroll
    for (int i = 0; i < D; ++i) tb[i] = tau;
    uint32_t n_data = 0, s_r_run = 0;
    double cur = now, stopw = 0.0, t_end = 0.0;
    bool cls_valid = false;
    uint32_t fbw = 0;               // feedback bytes of up to 4 steps, flushed as one dword

    const uint16_t* act = actions + (size_t)e * Kp;
    uint8_t* fbp = feedback + (size_t)e * Kp;
    uint4 aw = ld<uint4>(act, 0);   // actions of steps 0..7

    auto put_feedback = [&](uint32_t byte) {
        fbw |= byte << ((k & 3) * 8);
        if ((k & 3) == 3 || k == K - 1) { st_(fbp, (uint32_t)(k & ~3), fbw); fbw = 0; }
        k++;
        if ((k & 7) == 0 && k < K) aw = ld<uint4>(act, (uint32_t)k * 2u);                // next 8 actions
    };

    // ---- the lane's event loop: at most ONE transmission per iteration, every block appears once ----------
    while (k < K) {
        // (1) data mode: is there a packet that still fits the window?  (simple_stack.py:397-434)
        uint32_t s = 0;
        if (data_mode && !finish) {
            bool have = true;
            if (len_d == 0) {
                if (mult_d != 0u && wake < stopw) {          // mult 0: a silent sender, nothing will ever arrive
                    cur = wake;
                    wake = wake + interval;
                    tau++;
                    len_d = gw_len_after_ticks(0u, 1u, mult_d, kt);
                } else have = false;
            }
            if (have) {
                const uint32_t age = gw_ceil_div(len_d, mult_d, inv16_d);
                s = base_bytes + gw_tick_value(tau - age, bpc, bpp, nbp, hist, bound);
                const double need = m.over_rate((double)(s * 8u));
                if (!((stopw - cur) > need)) have = false;
            }
            if (!have) finish = true;
        }

        // (2) close the step (A.5 + interpreter feedback); the other senders' queues stay lazy
        if (finish) {
            // A.5, lazily: the counter ticks between the end of the window and the end of the step are counted by the
            // next step's first count (counting is cumulative in time).  What must not be lost is the diagnostic bit
            // for a tick falling EXACTLY on t_end.  A tie needs (t_end - wake) / interval within tie_filter of an
            // integer (host-side bound on the accumulated rounding of the running sum, gw_api.cpp); only then is the
            // exact comparison made, by the plain loop.
            {
                const double dd = t_end - wake;
                if (dd >= 0.0) {
                    const double q = dd * inv_interval;
                    if (!(fabs(q - rint(q)) > tie_filter) || !(wake >= 0.0625) || !(wake < 2097152.0)) {
                        for (double w = wake; w <= t_end; w = w + interval)
                            if (w == t_end) fl |= GW_FLAG_TIE;
                    }
                }
            }
            // the listeners' noise states: nothing to look up once every one of them is terminal
            bool all_term = true;
            if constexpr (GEN) {
                // any-D kernel: which senders sit in a non-terminal noise state is kept as a bit mask per lane, so the
                // common case (everyone terminal) costs no pass over the senders
                all_term = (live_mask & ~(1u << d)) == 0u;
                len[d] = len_d; tb[d] = tau;
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i) all_term = all_term && (i == d || ((term[i] >> sta[i]) & 1u));
#pragma unroll
                for (int i = 0; i < D; ++i)
                    if (i == d) { len[i] = len_d; tb[i] = tau; }
            }
            if (!all_term) {
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    if (i == d) continue;
                    uint32_t si = s_trans[(uint32_t)((i * R + RRM) * S) + sta[i]];      // heard the announcement
                    for (uint32_t n = 0; n < n_data; ++n) {                              // ... and d's data
                        const uint32_t s2 = s_trans[(uint32_t)((i * R + d) * S) + si];
                        if (s2 == si) break;
                        si = s2;
                    }
                    sta[i] = si;
                    if constexpr (GEN) live_mask = ((term[i] >> si) & 1u) ? (live_mask & ~(1u << i)) : (live_mask | (1u << i));
                }
            }
            sta[RRM] = s_r_run;
            const int32_t latest = pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u));
            const int32_t abs_d = latest < 0 ? -latest : latest;
            int32_t r = last_abs - abs_d;
            last_abs = abs_d;
            r = r > 10 ? 10 : (r < -10 ? -10 : r);
            now = t_end;
            finish = false;
            data_mode = false;
            mult_d = 0;                              // no addressed sender between steps
            put_feedback((uint32_t)((int)(rvm & 1u) - (int)((rvm >> 1) & 1u) + 1) | ((uint32_t)(r + 10) << 2) | (dn << 7));
            if (k >= K) break;
        }

        // (3) start of step k (counter_traffic.py:146-158): set up the announcement
        int pay_bytes;
        bool is_data = data_mode;
        if (!data_mode) {
            const uint32_t a = (word_of(aw, (k & 7) >> 1) >> ((k & 1) * 16)) & 0xffffu;
            d = (int)(a & 0xffu);
            const int du = (int)(a >> 8);
            if ((unsigned)d >= (unsigned)D || (unsigned)du >= (unsigned)c.max_duration) {
                fl |= GW_FLAG_BADACT;                // env untouched, feedback repeats the current values
                k_bad++;
                put_feedback((uint32_t)((int)(rvm & 1u) - (int)((rvm >> 1) & 1u) + 1) | (10u << 2) | (dn << 7));
                continue;
            }
            uint32_t l0 = 0, t0 = 0, s_d_old = 0;
            if constexpr (GEN) {
                l0 = len[d]; t0 = tb[d]; s_d_old = sta[d];
                mult_d = s_mi[d].x; inv16_d = s_mi[d].y;
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i)
                    if (i == d) { l0 = len[i]; t0 = tb[i]; mult_d = mult[i]; inv16_d = inv16[i]; s_d_old = sta[i]; }
            }
            len_d = gw_len_after_ticks(l0, tau - t0, mult_d, kt);          // the addressed queue, up to date
            const int slots = du * c.duration_factor;
            pay_bytes = ndigits(slots);
            cur = now;
            cls_valid = NOLIM || now < c.cls_limit;
            const uint32_t s_d = s_trans[(uint32_t)((d * R + RRM) * S) + s_d_old];   // d after hearing the RRM
#pragma unroll
            for (int i = 0; i < (GEN ? 0 : D); ++i) if (i == d) sta[i] = s_d;
            if constexpr (GEN) {
                sta[d] = s_d;
                live_mask = ((s_term[d] >> s_d) & 1u) ? (live_mask & ~(1u << d)) : (live_mask | (1u << d));
            }
            s_r_run = sta[RRM];
            n_data = 0;
            stopw = (double)slots * slot;            // become absolute times once t_r is known
            t_end = (double)(slots + 1) * slot;
        } else {
            len_d--;                                 // simple_stack.py:425
            kt.pop++;
            pay_bytes = (int)s - mh;
        }

        // (4) the transmission: slot alignment, durations, decode at the receiver
        uint32_t cls_x;
        double ber_x;
        if (is_data) {
            s_r_run = s_trans[(uint32_t)((RRM * R + d) * S) + s_r_run];     // the RRM hears sender d (again)
            ber_x = s_ber[(uint32_t)((D + d) * S) + s_r_run];
            cls_x = s_cls[(uint32_t)((D + d) * S) + s_r_run];
        } else {
            uint32_t s_d_now = 0;
#pragma unroll
            for (int i = 0; i < (GEN ? 0 : D); ++i) if (i == d) s_d_now = sta[i];
            if constexpr (GEN) s_d_now = sta[d];
            ber_x = s_ber[(uint32_t)(d * S) + s_d_now];
            cls_x = s_cls[(uint32_t)(d * S) + s_d_now];
        }
        const TxTimes x = tx_times(m, cur, hd, m.over_rate((double)(pay_bytes * 8)));
        kt.tx++;
        const bool ok = decode(m, cls_x, cls_valid, ber_x, x, br, hdr_bits, (double)(pay_bytes * 8) * c.coded_factor, fl);

        // (5) consequences + the counter ticks up to the new current time (one instance of the counting loop)
        bool incl;
        if (is_data) {
            n_data++;
            if (ok) {                                // devices.py:163-168, counter_traffic.py:75-80
                kt.deliv++;
                rvm |= (1u << d);
                if (pv == c.counter_bound) dn = 1u;
            }
            if (!(x.t_e < t_end)) fl |= GW_FLAG_CARRY;
            cur = x.t_e;
            incl = true;                             // ticks are older events than the MAC's resume
            if (!(cur < stopw)) finish = true;       // window timeout already processed
        } else {
            const double t_r = x.t_e;
            stopw = t_r + stopw;                     // simple_stack.py:401
            t_end = t_r + t_end;                     // simple_stack.py:557-558
            cur = t_r;
            incl = false;                            // ties at the window start: the MAC runs first
            data_mode = true;
            if (!ok) finish = true;                  // announcement not decoded: no window
        }
        {
            // the first count of a step also covers the ticks since the previous window closed (up to ~21):
            // one jump (gw_fastmath.h; exact, validated at gw_create) instead of a loop every lane would wait for
            uint32_t kk = 0;
            double wj = wake;
            bool tiej = false;
            if (fast_ticks && gw_tick_jump(wake, cur, interval, inv_interval, incl, &kk, &wj, &tiej)) {
                wake = wj;
                if (tiej) fl |= GW_FLAG_TIE;
            } else {
                kk = 0;
                for (;;) {
                    const double w1 = wake + interval, w2 = w1 + interval, w3 = w2 + interval, w4 = w3 + interval;
                    const bool b0 = incl ? (wake <= cur) : (wake < cur);
                    const bool b1 = incl ? (w1 <= cur) : (w1 < cur);
                    const bool b2 = incl ? (w2 <= cur) : (w2 < cur);
                    const bool b3 = incl ? (w3 <= cur) : (w3 < cur);
                    if (incl && (wake == cur || w1 == cur || w2 == cur || w3 == cur)) fl |= GW_FLAG_TIE;
                    kk += (uint32_t)b0 + (uint32_t)b1 + (uint32_t)b2 + (uint32_t)b3;
                    wake = b3 ? w4 : (b2 ? w3 : (b1 ? w2 : (b0 ? w1 : wake)));
                    if (!b3) break;
                }
            }
            tau += kk;
            len_d = gw_len_after_ticks(len_d, kk, mult_d, kt);
        }
    }

    // ---- catch up: ticks up to the end of the last step, every queue to the final tick ----------------------
    {
        uint32_t kk = 0;
        while (wake <= now) { if (wake == now) fl |= GW_FLAG_TIE; wake = wake + interval; kk++; }
        tau += kk;
#pragma unroll
        for (int i = 0; i < D; ++i) len[i] = gw_len_after_ticks(len[i], tau - tb[i], mult[i], kt);
    }

    // ---- registers -> state --------------------------------------------------------------------------------
    if constexpr (GEN) {
        for (int i = 0; i < D; ++i) st.qb[oq + (uint32_t)i] = (uint8_t)len[i];
        for (int j = 0; j < R; ++j) st.qb[oq + (uint32_t)(D + j)] = (uint8_t)sta[j];
    } else {
        uint32_t nb[16 * NWC];
#pragma unroll
        for (int b = 0; b < 16 * NWC; ++b) nb[b] = 0u;
#pragma unroll
        for (int i = 0; i < DT; ++i) nb[i] = len[i];
#pragma unroll
        for (int j = 0; j < DT + 1; ++j) nb[DT + j] = sta[j];
#pragma unroll
        for (int w = 0; w < NWC; ++w) {
            const int b = 16 * w;
            uint4 o;
            o.x = nb[b + 0] | (nb[b + 1] << 8) | (nb[b + 2] << 16) | (nb[b + 3] << 24);
            o.y = nb[b + 4] | (nb[b + 5] << 8) | (nb[b + 6] << 16) | (nb[b + 7] << 24);
            o.z = nb[b + 8] | (nb[b + 9] << 8) | (nb[b + 10] << 16) | (nb[b + 11] << 24);
            o.w = nb[b + 12] | (nb[b + 13] << 8) | (nb[b + 14] << 16) | (nb[b + 15] << 24);
            st_(st.qb, oq + 16u * w, o);
        }
    }
    st_(st.tw, o16, make_double2(now, wake));
    st_(st.tk, o16, make_uint4(tau, nbp, rvm, (uint32_t)last_abs | (dn << 31)));
    publish_env_counters(st.sa, N, e, kt.pop, kt.deliv, k_bad, fl, (uint32_t)K);
}

// ---- the step-synchronous form (every sender count) --------------------------------------------------------------
// The event loop above was built when a data packet cost as much as an announcement (one pass of a ~140-instruction body
// either way) and letting lanes run ahead of each other evened the work out.  Since the window loop has a straight-line form
// (ct_step_sfx.hip: ~45 instructions per packet, every per-packet decision settled for the step up front), a packet is a third
// of an announcement, and the lock-step of whole steps costs less than the event loop's ~60 exec-mask regions per pass: here
// every lane takes step k together -- announcement, window (straight line, general loop where that declines), feedback -- with
// the env's state in registers across all K steps, queue lengths of the senders not addressed and the ticks behind a window
// lazy exactly as above; DT == 0 = any sender count, the per-lane arrays as LDS columns.  Same results bit for bit (the tests
// run both forms against the oracle; GW_ROLLOUT_EVENT_LOOP=1 selects the event loop).
template <int DT, int MODE>
__global__ __launch_bounds__(64) void ct_rollout_sync_kernel(GwState st, GwDevConst c, int K,
                                                            const int32_t* __restrict__ device, const int32_t* __restrict__ duration,
                                                            int32_t* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done)
{
    // Actions and outputs in the C-ABI's own step-major layout ([K][N]: a step's row is coalesced across the wave's lanes), read
    // and written by this kernel itself: step k + 1's action is loaded while step k is walked, a step's three outputs are
    // stores nothing waits for.  (The event loop reads packed per-env action records and writes feedback bytes, with a
    // transposing kernel on either side: 15 us per 64 steps x 65 536 envs, an eighth of this kernel's own time.)
    constexpr bool GEN = DT == 0;                        // any sender count: per-lane arrays in LDS columns (as in the event loop)
    constexpr int DM = GEN ? GW_MAX_DEVICES : DT;        // capacity
    constexpr int NWC = (2 * DM + 1 + 15) / 16;
    constexpr int S = GW_MAX_NSTATES;
    const int D = GEN ? c.D : DT, R = D + 1, RRM = D;
    const uint32_t N = (uint32_t)st.N;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;

    constexpr int TRANS_B = ((DM + 1) * (DM + 1) * S + 15) / 16 * 16;
    __shared__ __attribute__((aligned(16))) uint8_t s_trans[TRANS_B];
    __shared__ __attribute__((aligned(16))) double  s_ber[2 * DM * S];
    __shared__ __attribute__((aligned(16))) uint8_t s_cls[2 * DM * S];
    __shared__ uint32_t s_cols[GEN ? (3 * DM + 1) * 64 : 1];     // GEN: len[D], tb[D], sta[R] columns per lane
    __shared__ uint2 s_mi[GEN ? DM : 1];                 // GEN: {mult, ceil(65536/mult)} and terminal-state masks, indexed by
    __shared__ uint32_t s_term[GEN ? DM : 1];            //      the lane's own addressed sender
    if constexpr (GEN) {
        for (int i = threadIdx.x; i < D; i += blockDim.x) {
            s_mi[i] = make_uint2((uint32_t)st.cst->mult[i], st.cst->inv16[i]);
            s_term[i] = st.cst->term[i];
        }
    }
    {
        const int n_tr = (R * R * S + 15) >> 4, n_be = (2 * D * S * 8) >> 4, n_cl = (2 * D * S) >> 4;
        for (int i = threadIdx.x; i < n_tr; i += blockDim.x) *reinterpret_cast<uint4*>(s_trans + ((uint32_t)i << 4)) = ld<uint4>(st.trans, (uint32_t)i << 4);
        for (int i = threadIdx.x; i < n_be; i += blockDim.x) *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(s_ber) + ((uint32_t)i << 4)) = ld<uint4>(st.ber2, (uint32_t)i << 4);
        for (int i = threadIdx.x; i < n_cl; i += blockDim.x) *reinterpret_cast<uint4*>(s_cls + ((uint32_t)i << 4)) = ld<uint4>(st.cls2, (uint32_t)i << 4);
    }
    __syncthreads();
    if (e >= N) return;

    // ---- state -> registers ----
    const uint32_t RB = GEN ? (uint32_t)st.RB : 16u * NWC;
    const uint32_t o16 = e << 4, oq = e * RB;
    const uint4 ip = ld<uint4>(st.ip, o16);
    const double2 tw = ld<double2>(st.tw, o16);
    const uint4 tk = ld<uint4>(st.tk, o16);
    typename ArrSel<GEN, DM>::rw len, tb;
    typename ArrSel<GEN, DM + 1>::rw sta;
    if constexpr (GEN) {
        uint32_t* col = s_cols + (threadIdx.x & 63);
        len.p = col; tb.p = col + DM * 64; sta.p = col + 2 * DM * 64;
        for (int i = 0; i < D; ++i) len[i] = st.qb[oq + (uint32_t)i];
        for (int j = 0; j < R; ++j) sta[j] = st.qb[oq + (uint32_t)(D + j)];
    } else {
        uint4 qw[NWC];
#pragma unroll
        for (int w = 0; w < NWC; ++w) qw[w] = ld<uint4>(st.qb, oq + 16u * w);
#pragma unroll
        for (int i = 0; i < DT; ++i) len[i] = (word_of(qw[i >> 4], (i >> 2) & 3) >> ((i & 3) * 8)) & 0xffu;
#pragma unroll
        for (int j = 0; j < DT + 1; ++j) sta[j] = (word_of(qw[(DT + j) >> 4], ((DT + j) >> 2) & 3) >> (((DT + j) & 3) * 8)) & 0xffu;
    }
    double now = tw.x, wake = tw.y;
    uint32_t tau = tk.x;
    const uint32_t nbp = tk.y;
    GwBp bpc, bpp;
    bpc.t0 = ip.x; bpc.c0 = ip.y;
    bpp.t0 = ip.z; bpp.c0 = ip.w;
    const GwBp* hist = st.bph + ((size_t)e << 7);
    uint32_t rvm = tk.z;
    int32_t last_abs = (int32_t)(tk.w & 0x7fffffffu);
    uint32_t dn = tk.w >> 31;

    constexpr bool FAST = MODE >= 1, NOLIM = MODE == 2;
    const StepMathT<FAST, NOLIM> m(c);
    const double slot = c.slot, br = c.bit_rate, hd = c.hdr_dur, hdr_bits = c.hdr_bits, interval = c.counter_interval;
    const double inv_interval = c.inv_interval, tie_filter = c.tie_filter, coded_factor = c.coded_factor;
    const bool fast_ticks = FAST || c.fast_ticks != 0;
    const bool idem = c.idem_states != 0;
    const uint32_t bound = (uint32_t)c.counter_bound, base_bytes = (uint32_t)(c.mac_hdr + c.net_hdr);
    const int mh = c.mac_hdr, pv = c.payload_value;
    uint32_t live_mask = 0u;                             // any-D kernel: bit i = sender i is in a non-terminal noise state
    if constexpr (GEN) {
        for (int i = 0; i < D; ++i) live_mask |= ((s_term[i] >> sta[i]) & 1u) ? 0u : (1u << i);
    }
    typename ArrSel<GEN, DM>::ro mult, term, inv16;
    if constexpr (GEN) {
        mult.p = st.cst->mult; mult.shift16 = 0; term.p = st.cst->term; term.shift16 = 1; inv16.p = st.cst->inv16; inv16.shift16 = 0;
    } else {
#pragma unroll
        for (int i = 0; i < DT; ++i) { mult.v[i] = (uint32_t)c.mult[i]; term.v[i] = c.term[i]; inv16.v[i] = c.inv16[i]; }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) tb[i] = tau;

    Tally kt = {0, 0, 0, 0, 0};
    uint32_t k_bad = 0, fl = 0;
    int k = 0;
    int d_next = device[e], du_next = duration[e];              // step 0's action
    auto put_feedback = [&](int32_t latest, int32_t r) {
        const size_t at = (size_t)k * N + e;
        obs[at] = latest + c.counter_bound;
        reward[at] = (float)r;
        done[at] = (uint8_t)dn;
        k++;
    };

    while (k < K) {
        // ---- start of step k (counter_traffic.py:146-158); the next step's action is requested now ----
        const int d = d_next, du = du_next;
        if (k + 1 < K) { d_next = device[(size_t)(k + 1) * N + e]; du_next = duration[(size_t)(k + 1) * N + e]; }
        if ((unsigned)d >= (unsigned)D || (unsigned)du >= (unsigned)c.max_duration) {
            fl |= GW_FLAG_BADACT;                    // env untouched, feedback repeats the current values
            k_bad++;
            put_feedback(pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u)), 0);
            continue;
        }
        uint32_t l0 = 0, t0 = 0, s_d_old = 0, mult_d = 0, inv16_d = 65536u;
        if constexpr (GEN) {
            l0 = len[d]; t0 = tb[d]; s_d_old = sta[d];
            mult_d = s_mi[d].x; inv16_d = s_mi[d].y;
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i)
                if (i == d) { l0 = len[i]; t0 = tb[i]; mult_d = mult[i]; inv16_d = inv16[i]; s_d_old = sta[i]; }
        }
        uint32_t len_d = gw_len_after_ticks(l0, tau - t0, mult_d, kt);          // the addressed queue, up to date
        const int slots = du * c.duration_factor;                               // counter_traffic.py:149
        const int Ld = ndigits(slots);
        const bool cls_valid = NOLIM || now < c.cls_limit;
        const uint32_t s_d = s_trans[(uint32_t)((d * R + RRM) * S) + s_d_old];  // d after hearing the RRM
#pragma unroll
        for (int i = 0; i < (GEN ? 0 : D); ++i) if (i == d) sta[i] = s_d;
        if constexpr (GEN) {
            sta[d] = s_d;
            live_mask = ((s_term[d] >> s_d) & 1u) ? (live_mask & ~(1u << d)) : (live_mask | (1u << d));
        }
        // ---- A.1 / A.2: announcement ----
        const TxTimes an = tx_times(m, now, hd, m.over_rate((double)(Ld * 8)));
        kt.tx++;
        const bool granted = decode(m, (uint32_t)s_cls[(uint32_t)(d * S) + s_d], cls_valid, s_ber[(uint32_t)(d * S) + s_d], an, br, hdr_bits,
                                    (double)(Ld * 8) * coded_factor, fl);
        const double t_r = an.t_e;
        const double t_end = t_r + (double)(slots + 1) * slot;                  // simple_stack.py:557-558
        uint32_t s_r = sta[RRM];
        uint32_t n_data = 0;

        // counter ticks with wake < t (or <= t): the running sum four at a time, or one jump where the step qualifies
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
            len_d = gw_len_after_ticks(len_d, kk, mult_d, kt);
        };
        double delta = 0.0;
        const bool span_ok = fast_ticks && gw_tick_span_ok(wake, t_end, interval, &delta);
        auto ticks_upto = [&](double t, bool inclusive) {
            uint32_t nj = 0;
            double wj = wake;
            bool tiej = false, sane = false;
            gw_tick_jump_lo(wake, t, delta, c.inv_interval_lo, inclusive, &nj, &wj, &tiej, &sane);
            if (span_ok && sane) {
                wake = wj;
                tau += nj;
                if (tiej) fl |= GW_FLAG_TIE;
                len_d = gw_len_after_ticks(len_d, nj, mult_d, kt);
            } else {
                ticks_to(t, inclusive);
            }
        };

        if (granted) {
            // ---- A.3 / A.4: window at sender d (simple_stack.py:397-434) ----
            const double stopw = t_r + (double)slots * slot;                    // :400-401
            double cur = t_r;
            ticks_upto(cur, false);                   // (covers the ticks since the previous window closed too: counting is cumulative)
            const uint32_t s_r1 = s_trans[(uint32_t)((RRM * R + d) * S) + s_r];            // the RRM after one packet of d
            const double ber_x1 = s_ber[(uint32_t)((D + d) * S) + s_r1];
            const uint32_t cls_x1 = s_cls[(uint32_t)((D + d) * S) + s_r1];
            bool more = true;
            uint32_t pops = 0;
            {
                // the straight-line form: preconditions and reasoning in ct_step_sfx.hip
                const double span = t_end - t_r;
                const bool straight = span_ok && mult_d != 0u && idem && cls_valid && cls_x1 != (uint32_t)GW_CLS_COMPUTE &&
                                      (FAST || (m.fast_fmod && m.fast_div)) && (NOLIM || t_end < m.fmod_limit) && t_r >= span + span;
                if (straight && len_d != 0u) {
                    auto head = [&](uint32_t ln, uint32_t tk_now, bool& deep) {
                        const uint32_t age = __umul24(ln + mult_d - 1u, inv16_d) >> 16;   // gw_ceil_div
                        const uint32_t ht = tk_now - age;
                        const bool older = ht < bpc.t0;
                        deep = older && ht < bpp.t0;
                        return base_bytes + gw_min_u32((older ? bpp.c0 : bpc.c0) + (ht - (older ? bpp.t0 : bpc.t0)), bound);
                    };
                    bool deep = false;
                    uint32_t chk = 0;
                    uint32_t sz = head(len_d, tau, deep);
                    bool go = !deep && (stopw - cur) > gw_fast_div((double)(sz * 8u), m.dr, m.rcp_dr);
                    while (go) {
                        const double pd = gw_fast_div((double)(((int)sz - mh) * 8), m.dr, m.rcp_dr);
                        const double t_s = cur + (m.slot - gw_fast_fmod_lo(cur, m.slot, c.inv_slot_lo));
                        const double t_e = t_s + (hd + pd);
                        uint32_t nj = 0;
                        double wj = wake;
                        bool tiej = false, sane = false;
                        gw_tick_jump_lo(wake, t_e, delta, c.inv_interval_lo, true, &nj, &wj, &tiej, &sane);
                        chk |= (sane ? 0u : (uint32_t)GW_FLAG_INTERNAL) | (tiej ? (uint32_t)GW_FLAG_TIE : 0u);
                        len_d = gw_min_u32(len_d - 1u + __umul24(nj, mult_d), (uint32_t)GW_QUEUE_CAP);
                        tau += nj;
                        wake = wj;
                        cur = t_e;
                        pops++;
                        bool deep_n = false;
                        sz = head(len_d, tau, deep_n);
                        go = cur < stopw && len_d != 0u && !deep_n && (stopw - cur) > gw_fast_div((double)(sz * 8u), m.dr, m.rcp_dr);
                    }
                    (void)head(len_d, tau, deep);
                    more = cur < stopw && (len_d == 0u || deep);
                    fl |= chk | ((pops && !(cur < t_end)) ? (uint32_t)GW_FLAG_CARRY : 0u);
                }
            }
            if (pops) {                                                         // devices.py:163-168, counter_traffic.py:75-80
                const bool okx = cls_x1 == (uint32_t)GW_CLS_OK;
                kt.pop += pops;
                kt.tx += pops;
                n_data += pops;
                s_r = s_r1;
                kt.deliv += okx ? pops : 0u;
                rvm |= okx ? (1u << d) : 0u;
                dn = (okx && pv == c.counter_bound) ? 1u : dn;
            }
            if (more)
            for (;;) {
                if (len_d == 0u) {                                              // :409-416
                    if (mult_d != 0u && wake < stopw) {
                        cur = wake;
                        wake = wake + interval;
                        tau++;
                        len_d = gw_len_after_ticks(0u, 1u, mult_d, kt);
                    } else break;
                }
                const uint32_t age = gw_ceil_div(len_d, mult_d, inv16_d);
                const uint32_t sz = base_bytes + gw_tick_value(tau - age, bpc, bpp, nbp, hist, bound);
                const double need = m.over_rate((double)(sz * 8u));             // messages.py:67-75
                if (!((stopw - cur) > need)) break;                             // :418-420
                len_d--;                                                        // :425
                kt.pop++;
                const int pay = (int)sz - mh;
                const TxTimes x = tx_times(m, cur, hd, m.over_rate((double)(pay * 8)));
                kt.tx++;
                n_data++;
                s_r = s_trans[(uint32_t)((RRM * R + d) * S) + s_r];             // the RRM hears sender d (again)
                const bool ok = decode(m, (uint32_t)s_cls[(uint32_t)((D + d) * S) + s_r], cls_valid, s_ber[(uint32_t)((D + d) * S) + s_r], x, br,
                                       hdr_bits, (double)(pay * 8) * coded_factor, fl);
                kt.deliv += ok ? 1u : 0u;
                rvm |= ok ? (1u << d) : 0u;
                dn = (ok && pv == c.counter_bound) ? 1u : dn;
                fl |= !(x.t_e < t_end) ? (uint32_t)GW_FLAG_CARRY : 0u;
                ticks_upto(x.t_e, true);                                        // ticks are older events than the MAC's resume
                cur = x.t_e;
                if (!(cur < stopw)) break;                                      // window timeout already processed
            }
        }

        // ---- close the step: A.5 lazily (the ticks up to t_end are counted by the next step's first count); what must not be
        //      lost is the diagnostic bit for a tick falling EXACTLY on t_end (see the event loop above) ----
        {
            const double dd = t_end - wake;
            if (dd >= 0.0) {
                const double q = dd * inv_interval;
                if (!(fabs(q - rint(q)) > tie_filter) || !(wake >= 0.0625) || !(wake < 2097152.0)) {
                    for (double w = wake; w <= t_end; w = w + interval)
                        if (w == t_end) fl |= GW_FLAG_TIE;
                }
            }
        }
        bool all_term = true;
        if constexpr (GEN) {
            all_term = (live_mask & ~(1u << d)) == 0u;
            len[d] = len_d; tb[d] = tau;
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) all_term = all_term && (i == d || ((term[i] >> sta[i]) & 1u));
#pragma unroll
            for (int i = 0; i < D; ++i)
                if (i == d) { len[i] = len_d; tb[i] = tau; }
        }
        if (!all_term) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                if (i == d) continue;
                uint32_t si = s_trans[(uint32_t)((i * R + RRM) * S) + sta[i]];  // heard the announcement
                for (uint32_t n = 0; n < n_data; ++n) {                          // ... and d's data
                    const uint32_t s2 = s_trans[(uint32_t)((i * R + d) * S) + si];
                    if (s2 == si) break;
                    si = s2;
                }
                sta[i] = si;
                if constexpr (GEN) live_mask = ((term[i] >> si) & 1u) ? (live_mask & ~(1u << i)) : (live_mask | (1u << i));
            }
        }
        sta[RRM] = s_r;
        const int32_t latest = pv * ((int)(rvm & 1u) - (int)((rvm >> 1) & 1u));
        const int32_t abs_d = latest < 0 ? -latest : latest;
        int32_t r = last_abs - abs_d;
        last_abs = abs_d;
        r = r > 10 ? 10 : (r < -10 ? -10 : r);
        now = t_end;
        put_feedback(latest, r);
    }

    // ---- catch up: ticks up to the end of the last step, every queue to the final tick ----
    {
        uint32_t kk = 0;
        while (wake <= now) { if (wake == now) fl |= GW_FLAG_TIE; wake = wake + interval; kk++; }
        tau += kk;
#pragma unroll
        for (int i = 0; i < D; ++i) len[i] = gw_len_after_ticks(len[i], tau - tb[i], mult[i], kt);
    }
    if constexpr (GEN) {
        for (int i = 0; i < D; ++i) st.qb[oq + (uint32_t)i] = (uint8_t)len[i];
        for (int j = 0; j < R; ++j) st.qb[oq + (uint32_t)(D + j)] = (uint8_t)sta[j];
    } else {
        uint32_t nb[16 * NWC];
#pragma unroll
        for (int b = 0; b < 16 * NWC; ++b) nb[b] = 0u;
#pragma unroll
        for (int i = 0; i < DT; ++i) nb[i] = len[i];
#pragma unroll
        for (int j = 0; j < DT + 1; ++j) nb[DT + j] = sta[j];
#pragma unroll
        for (int w = 0; w < NWC; ++w) {
            const int b = 16 * w;
            uint4 o;
            o.x = nb[b + 0] | (nb[b + 1] << 8) | (nb[b + 2] << 16) | (nb[b + 3] << 24);
            o.y = nb[b + 4] | (nb[b + 5] << 8) | (nb[b + 6] << 16) | (nb[b + 7] << 24);
            o.z = nb[b + 8] | (nb[b + 9] << 8) | (nb[b + 10] << 16) | (nb[b + 11] << 24);
            o.w = nb[b + 12] | (nb[b + 13] << 8) | (nb[b + 14] << 16) | (nb[b + 15] << 24);
            st_(st.qb, oq + 16u * w, o);
        }
    }
    st_(st.tw, o16, make_double2(now, wake));
    st_(st.tk, o16, make_uint4(tau, nbp, rvm, (uint32_t)last_abs | (dn << 31)));
    publish_env_counters(st.sa, N, e, kt.pop, kt.deliv, k_bad, fl, (uint32_t)K);
}

// the step-synchronous form: the caller's step-major arrays directly (no packing / expanding launches)
template <int DT>
int launch_rollout_sync(const GwState& st, const GwDevConst& cst, int K, const int32_t* device, const int32_t* duration,
                        int32_t* obs, float* reward, uint8_t* done, void* stream, bool below_limits)
{
    const unsigned blk = 64;
    const unsigned grid = (unsigned)((st.N + blk - 1) / blk);
    const bool fast = cst.fast_fmod && cst.fast_div && cst.fast_decide && cst.fast_ticks;
    if (fast && below_limits)
        hipLaunchKernelGGL((ct_rollout_sync_kernel<DT, 2>), dim3(grid), dim3(blk), 0, (hipStream_t)stream, st, cst, K, device, duration, obs, reward, done);
    else if (fast)
        hipLaunchKernelGGL((ct_rollout_sync_kernel<DT, 1>), dim3(grid), dim3(blk), 0, (hipStream_t)stream, st, cst, K, device, duration, obs, reward, done);
    else
        hipLaunchKernelGGL((ct_rollout_sync_kernel<DT, 0>), dim3(grid), dim3(blk), 0, (hipStream_t)stream, st, cst, K, device, duration, obs, reward, done);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}

template <int DT>
int launch_rollout(const GwState& st, const GwDevConst& cst, int K, int Kp, const uint16_t* act, uint8_t* fb, void* stream, bool below_limits)
{
    const unsigned blk = 64;
    const unsigned grid = (unsigned)((st.N + blk - 1) / blk);
    const bool fast = cst.fast_fmod && cst.fast_div && cst.fast_decide && cst.fast_ticks;
    if (fast && below_limits)
        hipLaunchKernelGGL((ct_rollout_sfx_kernel<DT, 2>), dim3(grid), dim3(blk), 0, (hipStream_t)stream, st, cst, K, Kp, act, fb);
    else if (fast)
        hipLaunchKernelGGL((ct_rollout_sfx_kernel<DT, 1>), dim3(grid), dim3(blk), 0, (hipStream_t)stream, st, cst, K, Kp, act, fb);
    else
        hipLaunchKernelGGL((ct_rollout_sfx_kernel<DT, 0>), dim3(grid), dim3(blk), 0, (hipStream_t)stream, st, cst, K, Kp, act, fb);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}

} // namespace

// Returns GW_EUNSUPPORTED when this (D, K) has no fused kernel: the caller falls back to K step launches.
int gw_launch_rollout_sfx(const GwState& st, const GwDevConst& cst, int K, const int32_t* device, const int32_t* duration,
                          int32_t* obs, float* reward, uint8_t* done, uint16_t* act_buf, uint8_t* fb_buf, int k_cap, void* stream,
                          bool below_limits)
{
    const int Kp = (K + 15) / 16 * 16;
    if (K <= 0 || Kp > k_cap) return GW_EUNSUPPORTED;
    // A/B switch: the older form -- for handles created while the switch was set (they have its scratch records)
    const bool event_loop = getenv("GW_ROLLOUT_EVENT_LOOP") != nullptr && act_buf != nullptr && fb_buf != nullptr;
    if (!event_loop) {
        switch (st.D) {
        case 2:  return launch_rollout_sync<2>(st, cst, K, device, duration, obs, reward, done, stream, below_limits);
        case 3:  return launch_rollout_sync<3>(st, cst, K, device, duration, obs, reward, done, stream, below_limits);
        case 4:  return launch_rollout_sync<4>(st, cst, K, device, duration, obs, reward, done, stream, below_limits);
        case 5:  return launch_rollout_sync<5>(st, cst, K, device, duration, obs, reward, done, stream, below_limits);
        case 6:  return launch_rollout_sync<6>(st, cst, K, device, duration, obs, reward, done, stream, below_limits);
        case 7:  return launch_rollout_sync<7>(st, cst, K, device, duration, obs, reward, done, stream, below_limits);
        case 8:  return launch_rollout_sync<8>(st, cst, K, device, duration, obs, reward, done, stream, below_limits);
        case 16: return launch_rollout_sync<16>(st, cst, K, device, duration, obs, reward, done, stream, below_limits);
        case 32: return launch_rollout_sync<32>(st, cst, K, device, duration, obs, reward, done, stream, below_limits);
        default: return launch_rollout_sync<0>(st, cst, K, device, duration, obs, reward, done, stream, below_limits);   // any other D: per-lane arrays in LDS
        }
    }
    if (cst.max_duration > 0xfe) return GW_EUNSUPPORTED;  // (the event loop's packed action records hold a byte of duration)
    const uint32_t N = (uint32_t)st.N;
    const unsigned g256 = (unsigned)((st.N + TP_ENVS - 1) / TP_ENVS);      // one block per 64-env tile
    hipLaunchKernelGGL(pack_actions_kernel, dim3(g256), dim3(256), 0, (hipStream_t)stream, N, K, Kp, device, duration, act_buf);
    int rc;
    switch (st.D) {
    case 2:  rc = launch_rollout<2>(st, cst, K, Kp, act_buf, fb_buf, stream, below_limits); break;
    case 3:  rc = launch_rollout<3>(st, cst, K, Kp, act_buf, fb_buf, stream, below_limits); break;
    case 4:  rc = launch_rollout<4>(st, cst, K, Kp, act_buf, fb_buf, stream, below_limits); break;
    case 6:  rc = launch_rollout<6>(st, cst, K, Kp, act_buf, fb_buf, stream, below_limits); break;
    case 8:  rc = launch_rollout<8>(st, cst, K, Kp, act_buf, fb_buf, stream, below_limits); break;
    case 16: rc = launch_rollout<16>(st, cst, K, Kp, act_buf, fb_buf, stream, below_limits); break;
    case 32: rc = launch_rollout<32>(st, cst, K, Kp, act_buf, fb_buf, stream, below_limits); break;
    default: rc = launch_rollout<0>(st, cst, K, Kp, act_buf, fb_buf, stream, below_limits); break;   // any other D (5, 7, ..., 32): per-lane arrays in LDS
    }
    if (rc) return rc;
    hipLaunchKernelGGL(expand_feedback_kernel, dim3(g256), dim3(256), 0, (hipStream_t)stream, N, K, Kp, cst.counter_bound,
                       cst.payload_value, fb_buf, obs, reward, done);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}
