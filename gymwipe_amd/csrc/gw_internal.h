// gw_internal.h -- shared between the host side (gw_api.cpp, gw_tables.cpp) and the
// HIP kernels (ct_step.hip).  Not part of the public C-ABI.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "../../include/gymwipe_amd.h"

#define GW_RING_PHYS      128          // physical ring slots per (env, sender); logical capacity GW_QUEUE_CAP
#define GW_RING_MASK      (GW_RING_PHYS - 1)
#ifndef GW_MAX_NSTATES
#define GW_MAX_NSTATES    16           // rx-power (noise residue) states per radio, see gw_tables.cpp
#endif

// Constants the kernels need, resident in device global memory (read through
// the scalar cache: every field is wave-uniform).
struct GwDevConst {
    int32_t D, R, S;                    // senders, radios (D+1), states per radio (GW_MAX_NSTATES)
    int32_t counter_bound, payload_value, mac_hdr, net_hdr, duration_factor, max_duration;
    int32_t mult[GW_MAX_DEVICES];
    double  slot, data_rate, bit_rate, coded_factor, max_ber, counter_interval;
    double  hdr_dur;                    // (mac_hdr*8)/data_rate
    double  hdr_bits;                   // (mac_hdr*8)*coded_factor
    // exact fast paths, each validated on the host at gw_create (gw_fastmath.h); 0 = use the plain form
    uint32_t inv16[GW_MAX_DEVICES];     // ceil(65536 / mult[i])
    uint16_t term[GW_MAX_RADIOS + 1];   // bit s of term[j]: noise state s of radio j is terminal (no talker changes it)
    double  inv_slot;                   // RN(1/slot)
    double  fmod_limit;                 // fast fmod is used for t < fmod_limit
    double  rcp_data_rate;              // RN(1/data_rate)
    double  cls_limit;                  // decode-certainty classes are valid for t < cls_limit
    int32_t fast_fmod, fast_div, fast_decide, idem_states;
    double  start_time;                 // simulated time at creation (test hook; the reference starts at 0)
    int32_t fast_ticks;                 // gw_tick_jump validated for counter_interval
    double  inv_interval;               // RN(1/counter_interval)
    double  tie_filter;                 // a tick can only fall exactly on t when (t - wake)/interval is this close to an integer
    int32_t no_traffic, peer_receive, float_duration;   // GW_CFG_NO_COUNTER_TRAFFIC / PEER_RECEIVE / FLOAT_DURATION
    int32_t dest[GW_MAX_DEVICES];
    double  ten_log_br, twenty_log_f, tx_power_dbm;   // 10*log10(bit_rate), 20*log10(frequency) (host glibc), tx power: live-PHY kernel
    uint32_t inv20[GW_MAX_DEVICES];     // ceil(2^20 / mult[i]): p / mult for p * mult < 2^20 (generic kernel's append index -> tick)
    double  inv_slot_lo, inv_interval_lo;   // one-sided reciprocals of gw_fast_fmod_lo / gw_tick_jump_lo (gw_fastmath.h)
};

struct GwBp { uint32_t t0, c0; };        // counting restarts at tick t0 with counter value c0
struct alignas(16) GwRec { uint32_t x, y, z, w; };   // a run-length queue's 16-byte record (gw_runq.h)

// Per-handle device state (structure of arrays; N = num_envs, D senders, R = D+1 radios).
struct GwState {
    int64_t   N;
    int32_t   D, R;     // host-side copies of the constants (launch sizing)
    int32_t   block;    // threads per workgroup of the step kernel
    int32_t   stage_chunks;  // suffix mode: 16-byte chunks of the state-major tables a step kernel stages in LDS (GwStripeLayout)
    double*   now;        // [N]        simulated time (SimMan.now)
    double*   wake;       // [N]        next counter tick (all senders tick in lock-step)
    uint32_t* counter;    // [N]        sender.counter (identical for all senders of an env)
    // explicit-queue mode (GW_CFG_EXPLICIT_QUEUE), packed so that an env's scalars are three 16-byte loads and stores:
    double*   xw;         // [N][2]     {now, next counter tick}
    uint32_t* xc;         // [N][4]     {counter, rvmask, last_abs | done << 31, sticky GW_FLAG_* bits}
    uint8_t*  xs;         // [N][XB]    rx-power state index per radio 0..D (stands for phy._receivedPower), padded to 16 bytes
    int32_t   XB;         //            bytes per xs record: 16 * ceil((D + 1) / 16)
    // MAC queue (SimpleMac._packetQueue), one of two encodings:
    //  explicit (GW_CFG_EXPLICIT_QUEUE): run-length deques, gw_runq.h
    GwRec*    qrec;       // [D][N]     16-byte record: head run, tail run, bookkeeping
    uint64_t* runs;       // [N][D][GW_RING_PHYS]  the runs in between (touched only when a run is created or used up)
    //  suffix (default): see gw_queue.h.  Packed so that one env costs four 16-byte loads:
    double*   tw;         // [N][2]     {now, next counter tick}
    uint32_t* tk;         // [N][4]     {tau = ticks so far, nbp = breakpoints so far, rvmask, last_abs | done << 31}: stored whole per step
    uint32_t* ip;         // [N][4]     {newest breakpoint (t0, c0), second newest breakpoint (t0, c0)}: written by reset / init only
    uint8_t*  qb;         // [N][RB]    bytes: queue length of sender 0..D-1, rx-power state of radio 0..D, pad
    GwBp*     bph;        // [N][GW_RING_PHYS]  ring of all breakpoints, entry j at [j & 127] (read only after >2 resets/100 ticks)
    int32_t   RB;         //            bytes per qb record: 16 * ceil((2*D + 1) / 16)
    uint16_t* ract;       // [N][rcap]  rollout scratch: packed actions (device | duration << 8)
    uint8_t*  rfb;        // [N][rcap]  rollout scratch: packed feedback bytes
    int32_t   rcap;       //            steps per fused rollout launch (multiple of 16)
    uint32_t* sa;         // [4N + 2]   per-env event counters that cannot be derived from the state: {popped, delivered} bad
                          //            actions, sticky flags (layout: GW_SA_WORDS), bumped by no-return atomics only where
                          //            something happened; + the handle's step count.
                          //            (steps = launches - bad; transmissions = steps + popped; appended = tau * sum(mult);
                          //            dropped = appended - popped - sum(len): gw_api.cpp derives them)
    uint32_t* rvmask;     // [N]        bit i set <=> receivedValues[i] == payload_value
    int32_t*  last_abs;   // [N]        interpreter._lastAbsDifference
    uint8_t*  done;       // [N]        interpreter._done
    uint8_t*  rxs;        // [R][N]     rx-power state index per radio (stands for phy._receivedPower)
    uint32_t* peer_rx;    // [D][N] or nullptr (GW_CFG_PEER_RECEIVE): packets a receive-mode MAC handed up
    uint32_t* flags;      // [N]        sticky GW_FLAG_* bits
    uint64_t* pe_stats;   // [5][N] or nullptr: n_tx, n_delivered, n_appended, n_popped, n_dropped
    unsigned long long* totals;  // [n_slots][GW_T_COUNT], one 64-B slot per wave of the step launch:
                                 // steps, tx, delivered, appended, popped, dropped, flags_or, bad_actions
    int64_t   n_slots;
    unsigned long long* stamps;  // diagnostic build only (make STAMPS=1): [n_slots][8] s_memtime stamps of the last launch
    const GwDevConst* cst;
    const uint8_t*    trans;     // [R to][R from][S]  state after hearing `from`
    const double*     ber;       // [R to][R from][S]  BER at `to` while hearing `from`, indexed by the NEW state
    const uint8_t*    cls;       // [R to][R from][S]  decode certainty at `to` hearing `from` (GW_CLS_*), by the NEW state
    // compact slices of ber/cls the step actually needs, staged in LDS by the suffix kernel:
    //   [0][d][s]: sender d hearing the RRM's announcement;  [1][d][s]: the RRM hearing sender d
    const double*     ber2;      // [2][D][S]
    const uint8_t*    cls2;      // [2][D][S]
    const uint8_t*    blob;      // GwBlobLayout: the default step kernel's tables in one block
    // live-PHY mode (ct_step_dyn.hip): f64 received power per radio instead of the noise-state bytes.  One env's radios are
    // CONTIGUOUS (rows of RP = gw_rp(R) doubles, 16-byte aligned): the all-pairs update of a step touches every radio of an
    // env and one or two rows of its link matrix, so whole rows are what a lane -- or, for D >= 8, a group of D lanes -- reads.
    double*   rxp;        // [N][RP]    phy._receivedPower, or nullptr (default mode)
    const double* prx_tab;   // [R][R]  link power from -> to, mW (host glibc tables)
    const double* pos_tab;   // [R][2]  the handle's geometry
    const double* extra_tab; // [R][R]  custom attenuation per pair, dB
    double*   prx_env;    // [N][R][RP] per-env link powers, row = talker (GW_CFG_PER_ENV_GEOMETRY), else nullptr
    double*   pos_env;    // [N][R][2]  per-env positions
    double*   bcache;     // [N][D][2][2] {noise power, BER} last evaluated for: sender i hearing the RRM (entry 2i), the RRM
                          //            hearing sender i (entry 2i + 1: the same 32 bytes, one memory line); key NaN = empty.  BpskMcs.calculateBitErrorRate is two
                          //            log10, three pow and a sqrt in f64 (physical.py:25-58,208-212): ~350 instructions that
                          //            a step would otherwise spend twice; received powers settle on a few residue values.
    double*   rxr;        // [N]        default queue mode: a MIRROR of rxp[e][RRM], the RRM's own received power, which the walking lane
                          //            needs every step -- above 16 radios it is the only word of the rxp row's second memory line
                          //            that a step reads (the row's authoritative copy is written along with it, when it changes)
    uint64_t* talk;       // [N]        bit r: radio r has transmitted, i.e. its attenuation models exist (physical.py:500-528
                          //            creates a pair's model at first use) -- what Position.set's keep-stale rules ask
};
#if defined(__HIPCC__)
__host__ __device__
#endif
// doubles per row of rxp / prx_env (the row pitch).  Rows are read as whole 128-byte memory lines, and on this GPU the live-PHY
// step is bound by the NUMBER of lines a CU has in flight times their latency (in-kernel stamps): so a row never
// straddles a line -- up to 16 radios the pitch is the next power of two (one line holds 16 / pitch whole rows), above that a
// multiple of 16 doubles, with the D senders' entries in the first lines and the RRM's (index D, read by the walking lane only)
// behind them: at D = 16 the lane groups' row reads are ONE aligned line each (pitch 18 = 144 bytes took two or three).
constexpr int gw_rp(int R)
{
    return R <= 2 ? 2 : (R <= 4 ? 4 : (R <= 8 ? 8 : (R <= 16 ? 16 : ((R + 15) & ~15))));
}

// byte offsets inside GwState::blob (the default step kernel's tables; see ct_step_sfx.hip)
struct GwBlobLayout {
    int ber, mi, h1, r1, cls, lds_total, h2, total;
#if defined(__HIPCC__)
    __host__ __device__
#endif
    constexpr explicit GwBlobLayout(int D)
        : ber(0), mi(2 * D * GW_MAX_NSTATES * 8), h1(mi + (D * 8 + 15) / 16 * 16), r1(h1 + D * GW_MAX_NSTATES),
          cls(r1 + D * GW_MAX_NSTATES), lds_total(cls + 2 * D * GW_MAX_NSTATES), h2(lds_total),
          total(h2 + D * D * GW_MAX_NSTATES) {}
};

// The same tables for the default (suffix-queue) step kernels, STATE-MAJOR: after the multiplicity table one stripe per noise
// state s, holding everything the step looks up for that state --
//     [ mi u32[D][2] | stripe 0 | stripe 1 | ... | stripe 15 | h2 u8[S][D][D] ]
//     stripe s = { ber0 f64[D] (sender d hears the RRM), ber1 f64[D] (the RRM hears d), h1 u8[D], r1 u8[D], cls0 u8[D], cls1 u8[D] }
// -- so that a step kernel stages only the prefix that covers the states this handle's layout actually has (3 at D = 2, 4 at
// D = 4, 6 at D = 16, 8 at D = 32 with the default geometry; 16 is the limit): 352 bytes instead of 1.3 KB at D = 4, 2 KB instead
// of 5.2 KB at D = 16, through an L1 path that four waves per CU share at 64 bytes per clock.
struct GwStripeLayout {
    int mi, s0, stripe, ber0, ber1, h1, r1, cls0, cls1, h2, total;
#if defined(__HIPCC__)
    __host__ __device__
#endif
    constexpr explicit GwStripeLayout(int D)
        : mi(0), s0((8 * D + 15) / 16 * 16), stripe((20 * D + 15) / 16 * 16), ber0(0), ber1(8 * D), h1(16 * D), r1(17 * D),
          cls0(18 * D), cls1(19 * D), h2(s0 + GW_MAX_NSTATES * stripe), total(h2 + D * D * GW_MAX_NSTATES) {}
#if defined(__HIPCC__)
    __host__ __device__
#endif
    constexpr int staged_chunks(int nstates) const { return (s0 + nstates * stripe + 15) / 16; }   // 16-byte chunks to put in LDS
};

// In the default (suffix-queue) mode the step tables, the handle's GwDevConst and its GwState live in the SAME allocation as
// the `ip` records, in a header of gw_blob_header(D) bytes in front of them (gw_api.cpp fills it at gw_create):
//     [ tables (GwStripeLayout) | GwDevConst at gw_hdr_cst_off | GwState at gw_hdr_st_off | pad to 256 ] [ ip records ... ]
// The per-step kernels get `ip` as a preloaded leading argument and derive everything else from it.  Their argument block
// shrinks from 1.2 KB (both structs by value) to under 100 bytes: the runtime writes a launch's arguments into
// device-visible memory, and for 1.2 KB that alone took 3.6 us of the host's 5.5 us per launch (tools/launch_floor.hip).
#if defined(__HIPCC__)
#define GW_HDC __host__ __device__
#else
#define GW_HDC
#endif
GW_HDC constexpr int gw_hdr_cst_off(int D) { return (GwStripeLayout(D).total + 16 + 15) / 16 * 16; }
GW_HDC constexpr int gw_hdr_st_off(int D) { return gw_hdr_cst_off(D) + ((int)sizeof(GwDevConst) + 15) / 16 * 16; }
GW_HDC constexpr int gw_blob_header(int D) { return (gw_hdr_st_off(D) + (int)sizeof(GwState) + 255) / 256 * 256; }

// decode certainty of a link in a given noise state (host: gw_tables.cpp; valid while t < fmod_limit)
enum { GW_CLS_COMPUTE = 0, GW_CLS_OK = 1, GW_CLS_HDR_FAIL = 2, GW_CLS_PAY_FAIL = 3 };

// layout of GwState::sa (uint32 words): [0, 2N) = per env {popped, delivered} as ONE u64 (one atomic bumps both),
// [2N, 3N) bad actions, [3N, 4N) sticky flags, then two words = the handle's env.step() count as a u64 (bumped by one lane
// per launch, so that launches replayed from a hipGraph are counted too)
enum { GW_SA_WORDS = 4 };

enum { GW_T_STEPS = 0, GW_T_TX, GW_T_DELIV, GW_T_APP, GW_T_POP, GW_T_DROP, GW_T_FLAGS, GW_T_BAD, GW_T_COUNT };

// Linear plant (plant_mfma.hip, gw_plant_api.cpp)
struct GwPlantDev {
    int64_t N;
    double* x;                    // [N][4]
    double* u;                    // [N]
    double* t_last;               // [N]
    unsigned long long* nsub;     // [N] substeps applied so far
    const double* Pop;            // [KMAX/4][64]  A operand of the state MFMA, by lane
    const double* Qtab;           // [KMAX+1][4]   Q_k = sum_{j<k} A^j B, the accumulated input vector of k substeps (Q_0 = 0)
    double dt, inv_dt;
};

// PHY grid (grid_phy.hip, gw_grid_api.cpp)
#define GW_GRID_EVENTS 11
struct GwGridLane {                     // one radio of one replica
    double   ev_t[GW_GRID_EVENTS];      // pending events of this device: time (+inf = none) ...
    uint32_t ev_k[GW_GRID_EVENTS];      // ... and key = priority bit | insertion id
    double   rx_power, tx_stop, rx_stop, rxi_stop, err_sum, ber, t_seg, px, py;
    uint32_t move_k, tx_on;             // moves made so far; 1 while this device's transmission is on the air (NOTIFY..END)
    uint32_t n_sent, queued, hdr_ok, hdr_fail, pay_ok, pay_fail, flags;
    int32_t  rx_src, rxi_src;
    uint8_t  started, handler_running, transmitting, receiving, waiting_rx, rx_running, rx_phase, pad;
};
struct GwGridEnv { double now; uint32_t eid, events, n_tx, first_run; };
struct GwGridDev {
    int64_t N; int32_t n, pad;
    GwGridLane* lanes;                  // [N][n]
    GwGridEnv*  envs;                   // [N]
    const double* prx;                  // [n from][n to] received power, mW
    double slot, send_interval, bit_rate, hdr_bits, pay_bits, hdr_dur, pay_dur, ten_log_br, sqrt2pi;
    uint32_t max_events, mobile;        // bound on events per launch (a logic error must not hang the GPU)
    double   move_interval, move_span, tx_power_dbm, twenty_log_f;
    double*  txp;                       // [N][n][n] mobile: received power of transmission i stored at radio j
    unsigned long long seed;
};
int gw_grid_launch_run(const GwGridDev& g, double seconds, void* stream);
int gw_grid_launch_set_position(const GwGridDev& g, int dev, const double* xs, const double* ys, void* stream);
int gw_grid_launch_init(const GwGridDev& g, const double* delays_dev, const double* pos_dev, double thermal, void* stream);

struct GwHostTables;

// Closed control loop (ctrl_step.hip, gw_ctrl_api.cpp)
struct GwCtrlDev {
    int64_t N;
    double *now, *wake, *x, *u, *ang;          // [N], [N], [N][4], [N], [N]
    uint32_t *ktick, *got, *ntx, *ncmd, *nsub, *flags;   // [N], [N][2], [N] ...
    uint16_t* qhl;                             // [2][N]  head | len << 8 of the sensor's and the controller's queue
    uint8_t*  rxs;                             // [4][N]  rx-power state per radio
    double*   pay;                             // [N][2][GW_RING_PHYS] payload values of the queued packets
    double A[16], B[4];
    uint32_t start, period;
    const GwDevConst* cst;
    const uint8_t* trans;                      // [4][4][S]
    const double*  ber;                        // [4][4][S]
};
int gw_ctrl_launch_step(const GwCtrlDev& c, const int32_t* device, const int32_t* duration, int32_t* obs, float* reward,
                        double* angle_deg, void* stream);
int gw_ctrl_launch_init(const GwCtrlDev& c, const double* x0, double u0, void* stream);
int gw_fill_dev_const(const gw_config& cfg, const GwHostTables& tab, GwDevConst& k);   // gw_api.cpp
int gw_validate_config(const gw_config& cfg);                                           // gw_api.cpp

// Host-side link tables (gw_tables.cpp)
struct GwHostTables {
    int D, R;
    double att[GW_MAX_RADIOS][GW_MAX_RADIOS];        // dB
    double prx[GW_MAX_RADIOS][GW_MAX_RADIOS];        // mW, [from][to]
    double thermal;                                   // mW
    double data_rate, coded_factor;
    int    overflow;                                  // some radio's state closure exceeds GW_MAX_NSTATES: live-PHY kernel
    int    nstates[GW_MAX_RADIOS];
    double state_val[GW_MAX_RADIOS][GW_MAX_NSTATES];  // mW; index 0 = thermal
    // flattened [to][from][s]
    uint8_t trans[GW_MAX_RADIOS * GW_MAX_RADIOS * GW_MAX_NSTATES];
    uint8_t cls[GW_MAX_RADIOS * GW_MAX_RADIOS * GW_MAX_NSTATES];
    double  ber[GW_MAX_RADIOS * GW_MAX_RADIOS * GW_MAX_NSTATES];
};

// returns GW_OK or an error code; msg receives a description on failure
int gw_build_tables(const gw_config& cfg, GwHostTables& out, char* msg, size_t msglen);

// kernel launchers (ct_step.hip); stream is a hipStream_t
int gw_launch_init(const GwState& st, void* stream);
int gw_launch_reset(const GwState& st, const uint8_t* mask, int32_t* obs, void* stream);
int gw_launch_step(const GwState& st, const int32_t* device, const int32_t* duration,
                   int32_t* obs, float* reward, uint8_t* done, void* stream);
int gw_launch_received(const GwState& st, int32_t* out, void* stream);
int gw_launch_enqueue(const GwState& st, int sender, const int32_t* payload_bytes, void* stream);
int gw_launch_step_sfx(const GwState& st, const GwDevConst& cst, const int32_t* device, const int32_t* duration,
                       int32_t* obs, float* reward, uint8_t* done, uint8_t* feedback_byte, void* stream, bool below_limits);
int gw_launch_reset_sfx(const GwState& st, const uint8_t* mask, int32_t* obs, void* stream);
int gw_launch_init_sfx(const GwState& st, void* stream);
int gw_launch_rollout_sfx(const GwState& st, const GwDevConst& cst, int K, const int32_t* device, const int32_t* duration,
                          int32_t* obs, float* reward, uint8_t* done, uint16_t* act_buf, uint8_t* fb_buf, int k_cap, void* stream,
                          bool below_limits);
int gw_launch_received_sfx(const GwState& st, int32_t* out, void* stream);
int gw_launch_delivered_sfx(const GwState& st, uint32_t* out, void* stream);
int gw_launch_clear_flags(const GwState& st, void* stream);          // ct_step_sfx.hip (both queue modes)
int gw_launch_step_dyn(const GwState& st, const GwDevConst& cst, const int32_t* device, const int32_t* duration,
                       int32_t* obs, float* reward, uint8_t* done, void* stream);
int gw_launch_init_dyn(const GwState& st, const GwDevConst& cst, double thermal, void* stream);
int gw_launch_set_position(const GwState& st, const GwDevConst& cst, int radio, const double* xs, const double* ys,
                           const double* all_pos, const uint8_t* mask, void* stream);

int gw_launch_pack_feedback(int64_t count, int center, int pv, const int32_t* obs, const float* reward, const uint8_t* done,
                            uint8_t* packed, uint32_t* bad, void* stream);
int gw_launch_unpack_feedback(int64_t count, int center, int pv, const uint8_t* packed, int32_t* obs, float* reward, uint8_t* done,
                              void* stream);

#define GW_MAX_MULT        100         // packets per tick in the suffix encoding = the handle-wide limit (the deque's capacity: more
                                       // than 100 packets per tick only ever leaves the last 100 in the queue).  gw_ceil_div's
                                       // reciprocal is exact for len <= 100 up to multiplicity 256 (checked exhaustively by
                                       // tests/test_host_logic.py), the kernels' 24-bit multiplies far beyond.
