// gw_api.cpp -- implementation of the C-ABI declared in include/gymwipe_amd.h.
// Host side only: configuration, table upload, HBM allocation, launches, readers.
// There is deliberately no CPU fallback: without a HIP device every compute
// entry point fails with GW_ENODEVICE.
#include "gw_internal.h"
#include "gw_queue.h"
#include "gw_runq.h"
#include "gw_fastmath.h"

#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include <deque>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

} // namespace

// shared with gw_plant_api.cpp
int gw_set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

namespace {

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return fail(GW_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e));            \
    } while (0)

} // namespace

struct gw_env {
    gw_config    cfg;
    GwHostTables tab;
    GwDevConst   cst_host;
    GwState      st;          // device pointers
    void*        blocks[32];  // every hipMalloc'd block, for gw_destroy
    size_t       block_bytes[32];
    int          nblocks;
    int          nblocks_create;   // blocks that exist since gw_create (later ones are scratch: gw_pack_feedback's counter)
    uint64_t     bytes;
    uint32_t*    pack_bad;    // device counter: elements gw_pack_feedback could not represent
    double       t_bound;     // upper bound of every env's simulated time (start + steps launched x step_max)
    double       step_max;    // upper bound of the simulated time one env.step() can take
    double       t_limit;     // below this clock value every validated fast form and certainty class holds
    int          captured;    // a launch was recorded into a hipGraph: its replays advance the clocks unseen by t_bound
    int          dyn;         // live-PHY mode (ct_step_dyn.hip): per-env geometry, or a geometry without a finite noise-state set
};

namespace {

template <class T>
int dev_alloc(gw_env* env, T** out, size_t count)
{
    void* p = nullptr;
    const size_t bytes = count * sizeof(T);
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(GW_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    if (env->nblocks >= (int)(sizeof env->blocks / sizeof env->blocks[0])) {
        (void)hipFree(p);
        return fail(GW_ENOMEM, "internal: block table full");
    }
    env->block_bytes[env->nblocks] = bytes;
    env->blocks[env->nblocks++] = p;
    env->bytes += bytes;
    *out = (T*)p;
    return GW_OK;
}

int select_device(const gw_env* env)
{
    HIP_TRY(hipSetDevice(env->cfg.hip_device));
    return GW_OK;
}

int launch_step(gw_env* env, const int32_t* device, const int32_t* duration,
                int32_t* obs, float* reward, uint8_t* done, void* stream, uint8_t* fb = nullptr, bool* fb_done = nullptr);

} // namespace
bool gw_env_below_limits(gw_env* env, void* stream);
namespace {

int validate(const gw_config& c)
{
    if (c.abi_version != GW_ABI_VERSION) return fail(GW_EINVAL, "abi_version %d != %d", c.abi_version, GW_ABI_VERSION);
    if (c.num_envs <= 0) return fail(GW_EINVAL, "num_envs must be positive");
    if (c.num_devices < 2 || c.num_devices > GW_MAX_DEVICES)
        return fail(GW_EINVAL, "num_devices must be in [2, %d] (observation uses senders 0 and 1)", GW_MAX_DEVICES);
    for (int i = 0; i < c.num_devices; ++i) {
        if (c.mult[i] < 0 || c.mult[i] > GW_QUEUE_CAP) return fail(GW_EINVAL, "mult[%d] out of range", i);   // 0 = a silent sender
        if (c.dest[i] < 0 || c.dest[i] >= c.num_devices) return fail(GW_EINVAL, "dest[%d] out of range", i);
    }
    if (!(c.slot > 0) || !(c.bit_rate > 0) || !(c.code_rate > 0 && c.code_rate <= 1) || !(c.counter_interval > 0))
        return fail(GW_EINVAL, "slot, bit_rate, code_rate and counter_interval must be positive");
    if (c.mac_header_bytes < 1 || c.net_header_bytes < 0 || c.counter_bound < 1 || c.duration_factor < 1 || c.max_duration < 1)
        return fail(GW_EINVAL, "header sizes / bounds out of range");
    if (c.payload_value < 0 || c.payload_value > (1 << 24) || c.counter_bound > (1 << 24))
        return fail(GW_EINVAL, "payload_value / counter_bound out of range (at most 2^24: packed with flag bits in the state)");
    if (!(c.flags & GW_CFG_EXPLICIT_QUEUE)) {
        for (int i = 0; i < c.num_devices; ++i)
            if (c.mult[i] > GW_MAX_MULT)
                return fail(GW_EUNSUPPORTED, "mult[%d] > %d needs GW_CFG_EXPLICIT_QUEUE", i, GW_MAX_MULT);
        // the default kernels address every per-env record with 32-bit byte offsets: e << 5 for the 32-byte counter record,
        // e * RB for the byte record (RB = 16 * ceil((2D + 1) / 16): 48 at D = 16, 80 at D = 32)
        const int64_t rb = 16 * ((2 * (int64_t)c.num_devices + 1 + 15) / 16);
        const int64_t cap = 0xffffffffll / (rb > 32 ? rb : 32);
        if (c.num_envs > cap)
            return fail(GW_EUNSUPPORTED, "num_envs %lld exceeds %lld, the most the default mode addresses at %d devices "
                        "(32-bit record offsets); use several handles", (long long)c.num_envs, (long long)cap, c.num_devices);
    }
    if ((c.flags & (GW_CFG_NO_COUNTER_TRAFFIC | GW_CFG_PEER_RECEIVE | GW_CFG_FLOAT_DURATION)) && !(c.flags & GW_CFG_EXPLICIT_QUEUE))
        return fail(GW_EUNSUPPORTED, "GW_CFG_NO_COUNTER_TRAFFIC / PEER_RECEIVE / FLOAT_DURATION need GW_CFG_EXPLICIT_QUEUE");
    for (int a = 0; a <= c.num_devices; ++a)
        for (int b = 0; b <= c.num_devices; ++b)
            if (!(c.extra_att_db[a][b] == c.extra_att_db[b][a]) || (a == b && c.extra_att_db[a][b] != 0.0))
                return fail(GW_EINVAL, "extra_att_db must be symmetric with a zero diagonal (pair %d,%d)", a, b);
    if (!(c.start_time >= 0.0) || !(c.start_time < 1e12)) return fail(GW_EINVAL, "start_time out of range");
    if ((int64_t)c.max_duration * c.duration_factor > 100000000)
        return fail(GW_EINVAL, "max_duration*duration_factor too large");
    return GW_OK;
}

// exact fast paths (gw_fastmath.h): enabled only when validated for THIS configuration
void set_fast_paths(const gw_config& cfg_ref, const GwHostTables& tab, GwDevConst& k)
{
    const gw_config* cfg = &cfg_ref;
    const int R = tab.R;
    k.inv_slot = 1.0 / cfg->slot;
    k.rcp_data_rate = 1.0 / tab.data_rate;
    k.fmod_limit = 0.0;
    const bool no_fast = getenv("GW_NO_FASTMATH") != nullptr;
    k.fast_fmod = (!no_fast && gw_fast_fmod_ok(cfg->slot, &k.fmod_limit)) ? 1 : 0;
    const int64_t max_bytes = (int64_t)cfg->mac_header_bytes + cfg->net_header_bytes + cfg->counter_bound + 64;
    k.fast_div = (!no_fast && gw_fast_div_ok(tab.data_rate, max_bytes)) ? 1 : 0;
    // round(err)/bits <= 0.25  <=>  4*round(err) <= bits  when bits is an integer (both < 2^53)
    k.cls_limit = (no_fast || getenv("GW_NO_CLASSES")) ? 0.0 : 1.0e6;
    k.fast_decide = (!no_fast && cfg->max_ber == 0.25 && tab.coded_factor * 8.0 == floor(tab.coded_factor * 8.0)) ? 1 : 0;
    k.inv_interval = 1.0 / cfg->counter_interval;
    k.inv_slot_lo = gw_inv_lo(cfg->slot);
    k.inv_interval_lo = gw_inv_c_lo(cfg->counter_interval);
    {
        // If tick number j after `wake` equals t exactly, then t - wake = j*c + E with |E| <= j * ulp(t)/2 (one rounding per
        // addition of the running sum), so (t - wake)/c is within j*ulp(t)/(2c) of j.  j <= jmax ticks per step, t < 2^21 s
        // where the filter is used; x8 for the roundings of the estimate itself and margin.
        const double step_max = (double)cfg->max_duration * cfg->duration_factor * cfg->slot + 0.05;
        const double jmax = ceil(step_max / cfg->counter_interval) + 2.0;
        k.tie_filter = 8.0 * jmax * ldexp(1.0, 21 - 52) / (2.0 * cfg->counter_interval);
        if (!(k.tie_filter < 0.25)) k.tie_filter = 1.0;     // filter useless: always compare exactly
    }
    k.fast_ticks = (!no_fast && !getenv("GW_NO_TICKJUMP") && gw_fast_ticks_ok(cfg->counter_interval)) ? 1 : 0;
    k.idem_states = 1;                              // hearing the same talker twice changes nothing more
    for (int to = 0; to < R && k.idem_states; ++to)
        for (int from = 0; from < R && k.idem_states; ++from) {
            if (to == from) continue;
            for (int s = 0; s < tab.nstates[to]; ++s) {
                const uint8_t s1 = tab.trans[((size_t)to * R + from) * GW_MAX_NSTATES + s];
                const uint8_t s2 = tab.trans[((size_t)to * R + from) * GW_MAX_NSTATES + s1];
                if (s1 != s2) { k.idem_states = 0; break; }
            }
        }
    if (tab.overflow) k.idem_states = 0;
    if (getenv("GW_NO_IDEM")) k.idem_states = 0;    // test switch: take the exact-count path although the map is idempotent
}

// fb: the step's feedback as one byte per env (gw_step_fb), written by the step kernel itself in the default mode;
// *fb_done tells the caller whether it was (else the caller runs the packing kernel)
int launch_step(gw_env* env, const int32_t* device, const int32_t* duration,
                int32_t* obs, float* reward, uint8_t* done, void* stream, uint8_t* fb, bool* fb_done)
{
    if (fb_done) *fb_done = false;
    if (env->dyn && env->st.tk) return gw_launch_step_dyn(env->st, env->cst_host, device, duration, obs, reward, done, stream);
    if (env->st.tk) {
        if (fb_done) *fb_done = true;
        return gw_launch_step_sfx(env->st, env->cst_host, device, duration, obs, reward, done, fb, stream, gw_env_below_limits(env, stream));
    }
    return gw_launch_step(env->st, device, duration, obs, reward, done, stream);
}

// Expand the suffix-encoded queue of one sender into packet byte sizes, head first (gw_queue.h).
void expand_sfx(uint32_t bound, uint32_t base, uint32_t mult, uint32_t len, uint32_t tau,
                GwBp cur, GwBp prev, uint32_t nbp, const GwBp* hist, uint32_t* out /* [GW_QUEUE_CAP] */)
{
    // the stream holds mult*tau packets; the queue is its last `len`; packet a sits in tick a / mult
    const uint64_t total = (uint64_t)mult * tau;
    for (uint32_t p = 0; p < len; ++p) {
        const uint64_t a = total - len + p;
        out[p] = base + gw_tick_value((uint32_t)(a / mult), cur, prev, nbp, hist, bound);
    }
}

} // namespace

// every wave-uniform constant the step kernels need (shared with gw_ctrl_api.cpp)
int gw_fill_dev_const(const gw_config& cfg, const GwHostTables& tab, GwDevConst& k)
{
    const int D = cfg.num_devices, R = D + 1;
    k.D = D; k.R = R; k.S = GW_MAX_NSTATES;
    k.counter_bound = cfg.counter_bound; k.payload_value = cfg.payload_value;
    k.mac_hdr = cfg.mac_header_bytes; k.net_hdr = cfg.net_header_bytes;
    k.duration_factor = cfg.duration_factor; k.max_duration = cfg.max_duration;
    for (int i = 0; i < D; ++i) { k.mult[i] = cfg.mult[i]; k.inv16[i] = cfg.mult[i] > 0 ? (65536u + (uint32_t)cfg.mult[i] - 1u) / (uint32_t)cfg.mult[i] : 0u; }
    for (int i = 0; i < D; ++i) k.inv20[i] = cfg.mult[i] > 0 ? ((1u << 20) + (uint32_t)cfg.mult[i] - 1u) / (uint32_t)cfg.mult[i] : 0u;
    k.ten_log_br = 10 * log10(cfg.bit_rate);            // physical.py:38-42 (the same libm call CPython makes)
    k.twenty_log_f = 20 * log10(cfg.frequency);         // attenuation_models.py:35
    k.tx_power_dbm = cfg.tx_power_dbm;
    k.start_time = cfg.start_time;
    k.no_traffic = (cfg.flags & GW_CFG_NO_COUNTER_TRAFFIC) ? 1 : 0;
    k.peer_receive = (cfg.flags & GW_CFG_PEER_RECEIVE) ? 1 : 0;
    k.float_duration = (cfg.flags & GW_CFG_FLOAT_DURATION) ? 1 : 0;
    for (int i = 0; i < D; ++i) k.dest[i] = cfg.dest[i];
    k.slot = cfg.slot; k.data_rate = tab.data_rate; k.bit_rate = cfg.bit_rate;
    k.coded_factor = tab.coded_factor; k.max_ber = cfg.max_ber; k.counter_interval = cfg.counter_interval;
    {
        volatile double hb = (double)(cfg.mac_header_bytes * 8);
        k.hdr_dur = hb / tab.data_rate;        // physical.py:244
        k.hdr_bits = hb * tab.coded_factor;    // physical.py:259
    }

    set_fast_paths(cfg, tab, k);
    for (int j = 0; j < R; ++j) {
        uint16_t m = 0;
        for (int s0 = 0; s0 < tab.nstates[j]; ++s0) {
            bool fixed = true;
            for (int f = 0; f < R && fixed; ++f)
                if (f != j && tab.trans[((size_t)j * R + f) * GW_MAX_NSTATES + s0] != s0) fixed = false;
            if (fixed) m |= (uint16_t)(1u << s0);
        }
        k.term[j] = m;
    }
    return GW_OK;
}

int gw_validate_config(const gw_config& cfg) { return validate(cfg); }

// Default mode: the per-env event counts.  Only {popped, delivered, bad actions, flags} are counted on the device
// (GwState::sa); the rest follows from the state: every env.step() call steps every env unless its action was bad, every
// step transmits one announcement and one data packet per pop, every counter tick appends mult_i packets at sender i, and
// a packet that was appended is queued, popped or dropped.
struct GwEnvCounts { std::vector<uint64_t> steps, tx, deliv, app, pop, drop, bad; std::vector<uint32_t> flags; };
static int derive_env_counts(gw_env* env, GwEnvCounts& c)
{
    const GwState& st = env->st;
    const int64_t N = st.N;
    const int D = st.D, RB = st.RB;
    std::vector<uint32_t> sa((size_t)N * GW_SA_WORDS + 2), tk((size_t)N * 4);
    std::vector<uint8_t> qb((size_t)N * RB);
    HIP_TRY(hipMemcpy(sa.data(), st.sa, sa.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(tk.data(), st.tk, tk.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(qb.data(), st.qb, qb.size(), hipMemcpyDeviceToHost));
    uint64_t mult_sum = 0;
    for (int i = 0; i < D; ++i) mult_sum += (uint64_t)env->cfg.mult[i];
    const uint64_t n_steps = (uint64_t)sa[(size_t)4 * N] | ((uint64_t)sa[(size_t)4 * N + 1] << 32);   // counted on the device
    c.steps.resize(N); c.tx.resize(N); c.deliv.resize(N); c.app.resize(N); c.pop.resize(N); c.drop.resize(N); c.bad.resize(N);
    c.flags.resize(N);
    for (int64_t e = 0; e < N; ++e) {
        uint64_t queued = 0;
        for (int i = 0; i < D; ++i) queued += qb[(size_t)e * RB + i];
        c.pop[e] = sa[(size_t)2 * e];
        c.deliv[e] = sa[(size_t)2 * e + 1];
        c.bad[e] = sa[(size_t)2 * N + e];
        c.flags[e] = sa[(size_t)3 * N + e];
        c.steps[e] = n_steps - c.bad[e];
        c.tx[e] = c.steps[e] + c.pop[e];
        c.app[e] = (uint64_t)tk[(size_t)e * 4] * mult_sum;
        c.drop[e] = c.app[e] - c.pop[e] - queued;
    }
    return GW_OK;
}

void gw_env_add_steps(gw_env* env, uint64_t n) { env->t_bound += (double)n * env->step_max; }
// May this launch skip the per-lane validity-limit tests of the fast forms?  Only while the host's bound on the simulated
// time is good: a launch recorded into a hipGraph can be replayed any number of times behind the host's back, so the first
// capture seen on a handle switches the shortcut off for good (and the captured launch itself keeps the tests).
bool gw_env_below_limits(gw_env* env, void* stream)
{
    if (env->captured) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        env->captured = 1;
        return false;
    }
    (void)hipGetLastError();
    return env->t_bound + env->step_max < env->t_limit;
}

// for the entry points that drive a gw_env together with another handle (gw_plant_api.cpp: gw_pendulum_step)
int gw_env_internals(gw_env* env, const GwState** st, const GwDevConst** cst, int* hip_device)
{
    if (!env) return fail(GW_EINVAL, "env is NULL");
    if (env->dyn) return fail(GW_EUNSUPPORTED, "not available in the live-PHY mode (per-env geometry / open noise-state set)");
    *st = &env->st; *cst = &env->cst_host; *hip_device = env->cfg.hip_device;
    return GW_OK;
}

extern "C" {

int gw_abi_version(void) { return GW_ABI_VERSION; }

const char* gw_last_error(void) { return g_err; }

int gw_device_count(int* count)
{
    if (!count) return fail(GW_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(GW_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return GW_OK;
}

int gw_config_default(gw_config* c, int64_t num_envs, int32_t D)
{
    if (!c) return fail(GW_EINVAL, "cfg is NULL");
    if (D < 2 || D > GW_MAX_DEVICES) return fail(GW_EINVAL, "num_devices must be in [2, %d]", GW_MAX_DEVICES);
    memset(c, 0, sizeof *c);
    c->abi_version = GW_ABI_VERSION;
    c->hip_device = 0;
    c->num_envs = num_envs;
    c->num_devices = D;
    if (D == 2) {                                   // counter_traffic.py:124-127
        c->pos[0][0] = 0.0; c->pos[0][1] = 2.0;
        c->pos[1][0] = 0.0; c->pos[1][1] = -2.0;
    } else {                                        // circle of radius 2 m around the RRM
        for (int i = 0; i < D; ++i) {
            const double ang = M_PI / 2 - 2 * M_PI * i / D;
            c->pos[i][0] = 2.0 * cos(ang);
            c->pos[i][1] = 2.0 * sin(ang);
        }
    }
    c->pos[D][0] = 0.0; c->pos[D][1] = 0.0;         // counter_traffic.py:133
    for (int i = 0; i < D; ++i) {
        c->mult[i] = (i % 2 == 0) ? 1 : 3;          // counter_traffic.py:125-126
        c->dest[i] = (i + 1) % D;                   // :129-130
    }
    c->slot = 1e-6;                                 // simple_stack.py:27
    c->frequency = 2.4e9;                           // physical.py:298
    c->bandwidth = 22e6;
    c->temperature_c = 20.0;                        // simple_stack.py:57
    c->bit_rate = 133.33333e3;                      // physical.py:196
    c->code_rate = 0.75;                            // physical.py:192
    c->max_ber = 0.25;                              // physical.py:160-185 for 3/4
    c->tx_power_dbm = 0.0;                          // simple_stack.py:364,521
    c->counter_interval = 0.001;                    // counter_traffic.py:31
    c->counter_bound = 65536;                       // :35
    c->payload_value = 2;                           // :57 (swapped constructor arguments)
    c->mac_header_bytes = 13;                       // messages.py:154
    c->net_header_bytes = 12;                       // messages.py:180
    c->duration_factor = 1000;                      // envs/core.py:27
    c->max_duration = 20;                           // envs/core.py:25
    return GW_OK;
}

int gw_create(const gw_config* cfg, gw_env** out)
{
    if (!cfg || !out) return fail(GW_EINVAL, "cfg/out is NULL");
    *out = nullptr;
    int rc = validate(*cfg);
    if (rc) return rc;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GW_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
    if (cfg->hip_device < 0 || cfg->hip_device >= ndev)
        return fail(GW_EINVAL, "hip_device %d out of range (have %d)", cfg->hip_device, ndev);

    gw_env* env = new (std::nothrow) gw_env();
    if (!env) return fail(GW_ENOMEM, "out of host memory");
    memset(env, 0, sizeof *env);
    env->cfg = *cfg;

    char msg[256] = "";
    rc = gw_build_tables(env->cfg, env->tab, msg, sizeof msg);
    if (rc) { delete env; return fail(rc, "%s", msg); }

    if ((rc = select_device(env))) { delete env; return rc; }

    const int D = cfg->num_devices, R = D + 1;
    const int64_t N = cfg->num_envs;
    GwDevConst& k = env->cst_host;
    gw_fill_dev_const(*cfg, env->tab, k);

    {
        // one env.step() advances an env's clock by at most: slot alignment + announcement + window + guard slot
        const double max_slots = (double)(cfg->max_duration - 1) * cfg->duration_factor;
        env->step_max = ((max_slots + 3.0) * cfg->slot + (double)(cfg->mac_header_bytes + 16) * 8.0 / env->tab.data_rate) * (1.0 + 1e-9) + 1e-9;
        env->t_bound = cfg->start_time;
        env->t_limit = (k.fast_fmod && k.cls_limit > 0.0) ? (k.fmod_limit < k.cls_limit ? k.fmod_limit : k.cls_limit) : 0.0;
    }
    GwState& st = env->st;
    st.N = N; st.D = D; st.R = R;
    {
        const char* kb = getenv("GW_BLOCK");        // tuning knob: threads per workgroup of the generic step kernel (the
                                                    // suffix-queue kernels are fixed at 64: measured best, and a compile-time
                                                    // block size keeps the hidden-argument load off their first cycles)
        st.block = kb ? atoi(kb) : 64;
        if (st.block != 16 && st.block != 32 && st.block != 64 && st.block != 128 && st.block != 256) st.block = 64;
    }
    GwDevConst* d_cst = nullptr; uint8_t* d_trans = nullptr; double* d_ber = nullptr;
    const size_t tcount = (size_t)R * R * GW_MAX_NSTATES;
#define TRY_ALLOC(ptr, count) do { rc = dev_alloc(env, &(ptr), (size_t)(count)); if (rc) { gw_destroy(env); return rc; } } while (0)
    const bool explicit_q = (cfg->flags & GW_CFG_EXPLICIT_QUEUE) != 0;
    const bool per_env_geo = (cfg->flags & GW_CFG_PER_ENV_GEOMETRY) != 0;
    env->dyn = (per_env_geo || env->tab.overflow) ? 1 : 0;
    uint8_t* d_cls = nullptr; double* d_ber2 = nullptr; uint8_t* d_cls2 = nullptr; uint8_t* d_blob = nullptr;
    if (explicit_q) {
        st.XB = 16 * ((R + 15) / 16);
        TRY_ALLOC(st.xw, N * 2);    TRY_ALLOC(st.xc, N * 4);    TRY_ALLOC(st.xs, N * st.XB);
        TRY_ALLOC(st.qrec, N * D);  TRY_ALLOC(st.runs, N * D * GW_RING_PHYS);
    } else {
        st.RB = 16 * ((2 * D + 1 + 15) / 16);
        TRY_ALLOC(st.tw, N * 2);   TRY_ALLOC(st.tk, N * 4);
        {                                                   // the step tables and the `ip` records share a block: gw_blob_header()
            uint8_t* blk = nullptr;
            TRY_ALLOC(blk, (size_t)gw_blob_header(D) + (size_t)N * 16);
            d_blob = blk;
            st.ip = reinterpret_cast<uint32_t*>(blk + gw_blob_header(D));
        }
        TRY_ALLOC(st.qb, N * st.RB);  TRY_ALLOC(st.bph, N * GW_RING_PHYS);
        TRY_ALLOC(st.sa, N * GW_SA_WORDS + 2);
        {
            const char* rc_env = getenv("GW_ROLLOUT_CAP");      // steps per fused rollout launch
            int cap = rc_env ? atoi(rc_env) : 64;
            if (cap < 0) cap = 0;
            st.rcap = (cap + 15) / 16 * 16;
            // (scratch of the event-loop form only -- packed action / feedback records, 3 bytes per env and step of a chunk: an A/B
            //  switch since the step-synchronous kernel reads and writes the caller's arrays; the switch is read here)
            if (st.rcap > 0 && getenv("GW_ROLLOUT_EVENT_LOOP")) { TRY_ALLOC(st.ract, N * st.rcap);  TRY_ALLOC(st.rfb, N * st.rcap); }
        }
    }
    double *d_prx = nullptr, *d_pos = nullptr, *d_extra = nullptr;
    if (env->dyn) {
        const int RP = gw_rp(R);                            // rows of one env's radios, 16-byte aligned (gw_internal.h)
        TRY_ALLOC(st.rxp, N * RP);  TRY_ALLOC(d_prx, R * R);  TRY_ALLOC(d_pos, R * 2);  TRY_ALLOC(d_extra, R * R);
        if (!explicit_q) { TRY_ALLOC(st.bcache, N * 2 * D * 2);  TRY_ALLOC(st.rxr, N); }
        if (per_env_geo) { TRY_ALLOC(st.prx_env, N * R * RP);  TRY_ALLOC(st.pos_env, N * R * 2); }
        if (per_env_geo && explicit_q) TRY_ALLOC(st.talk, N);   // (default queue mode: the mask lives in spare bytes of the qb record)
        st.prx_tab = d_prx; st.pos_tab = d_pos; st.extra_tab = d_extra;
    }
    if (explicit_q && (cfg->flags & GW_CFG_PEER_RECEIVE)) TRY_ALLOC(st.peer_rx, N * D);
    if (explicit_q && (cfg->flags & GW_CFG_PER_ENV_STATS)) TRY_ALLOC(st.pe_stats, N * 5);
    st.n_slots = (N + 15) / 16 + 1;                   // one per wave; sized for the narrowest block (16) and for two waves per 64 envs
    if (explicit_q) TRY_ALLOC(st.totals, st.n_slots * GW_T_COUNT);
#ifdef GW_STAMPS
    TRY_ALLOC(st.stamps, st.n_slots * 16);
#endif
    TRY_ALLOC(d_cst, 1);       TRY_ALLOC(d_trans, tcount + 16);  TRY_ALLOC(d_ber, tcount);  TRY_ALLOC(d_cls, tcount);
    TRY_ALLOC(d_ber2, 2 * D * GW_MAX_NSTATES + 2);  TRY_ALLOC(d_cls2, 2 * D * GW_MAX_NSTATES + 16);
    if (!d_blob) TRY_ALLOC(d_blob, GwBlobLayout(D).total + 16);
#undef TRY_ALLOC
    st.cst = d_cst; st.trans = d_trans; st.ber = d_ber; st.cls = d_cls; st.ber2 = d_ber2; st.cls2 = d_cls2; st.blob = d_blob;

#define HIP_TRY_D(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { rc = fail(GW_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); gw_destroy(env); return rc; } } while (0)
    HIP_TRY_D(hipMemcpy(d_cst, &k, sizeof k, hipMemcpyHostToDevice));
    HIP_TRY_D(hipMemcpy(d_trans, env->tab.trans, tcount * sizeof(uint8_t), hipMemcpyHostToDevice));
    HIP_TRY_D(hipMemcpy(d_ber, env->tab.ber, tcount * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY_D(hipMemcpy(d_cls, env->tab.cls, tcount * sizeof(uint8_t), hipMemcpyHostToDevice));
    {
        std::vector<double> b2((size_t)2 * D * GW_MAX_NSTATES);
        std::vector<uint8_t> c2((size_t)2 * D * GW_MAX_NSTATES);
        for (int dd = 0; dd < D; ++dd)
            for (int ss = 0; ss < GW_MAX_NSTATES; ++ss) {
                const size_t ann = ((size_t)dd * R + D) * GW_MAX_NSTATES + ss;      // to = dd, from = RRM
                const size_t dat = ((size_t)D * R + dd) * GW_MAX_NSTATES + ss;      // to = RRM, from = dd
                b2[(size_t)dd * GW_MAX_NSTATES + ss] = env->tab.ber[ann];
                c2[(size_t)dd * GW_MAX_NSTATES + ss] = env->tab.cls[ann];
                b2[((size_t)D + dd) * GW_MAX_NSTATES + ss] = env->tab.ber[dat];
                c2[((size_t)D + dd) * GW_MAX_NSTATES + ss] = env->tab.cls[dat];
            }
        HIP_TRY_D(hipMemcpy(d_ber2, b2.data(), b2.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY_D(hipMemcpy(d_cls2, c2.data(), c2.size(), hipMemcpyHostToDevice));
        // the default step kernel's tables in one block (GwBlobLayout)
        const GwBlobLayout L(D);
        std::vector<uint8_t> blob((size_t)L.total, 0);
        memcpy(blob.data() + L.ber, b2.data(), b2.size() * sizeof(double));
        memcpy(blob.data() + L.cls, c2.data(), c2.size());
        const uint8_t* tr = env->tab.trans;
        const int S = GW_MAX_NSTATES;
        for (int j = 0; j < D; ++j) {
            const uint32_t mi[2] = {(uint32_t)k.mult[j], k.inv16[j]};
            memcpy(blob.data() + L.mi + (size_t)j * 8, mi, 8);
            for (int s0 = 0; s0 < S; ++s0) {
                const uint8_t a = tr[((size_t)j * R + D) * S + s0];                 // j hears the RRM
                blob[(size_t)L.h1 + (size_t)j * S + s0] = a;
                blob[(size_t)L.r1 + (size_t)j * S + s0] = tr[((size_t)D * R + j) * S + s0];   // the RRM hears j
                for (int dd = 0; dd < D; ++dd)
                    blob[(size_t)L.h2 + ((size_t)j * D + dd) * S + s0] = (j == dd) ? a : tr[((size_t)j * R + dd) * S + a];
            }
        }
        if (!st.ip) {                               // explicit mode: the generic kernel's tables (GwBlobLayout)
            HIP_TRY_D(hipMemcpy(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
        } else {                                    // suffix mode: the same numbers state-major (GwStripeLayout), in the ip block's header
            const GwStripeLayout T(D);
            std::vector<uint8_t> sb((size_t)T.total, 0);
            int nst = 1;
            for (int r = 0; r < R; ++r) nst = env->tab.nstates[r] > nst ? env->tab.nstates[r] : nst;
            if (nst > S) nst = S;
            for (int j = 0; j < D; ++j) {
                memcpy(sb.data() + T.mi + (size_t)j * 8, blob.data() + L.mi + (size_t)j * 8, 8);
                for (int s0 = 0; s0 < S; ++s0) {
                    uint8_t* stripe = sb.data() + T.s0 + (size_t)s0 * T.stripe;
                    memcpy(stripe + T.ber0 + (size_t)j * 8, &b2[(size_t)j * S + s0], 8);
                    memcpy(stripe + T.ber1 + (size_t)j * 8, &b2[((size_t)D + j) * S + s0], 8);
                    stripe[T.h1 + j] = blob[(size_t)L.h1 + (size_t)j * S + s0];
                    stripe[T.r1 + j] = blob[(size_t)L.r1 + (size_t)j * S + s0];
                    stripe[T.cls0 + j] = c2[(size_t)j * S + s0];
                    stripe[T.cls1 + j] = c2[((size_t)D + j) * S + s0];
                    for (int dd = 0; dd < D; ++dd)
                        sb[(size_t)T.h2 + ((size_t)s0 * D + j) * D + dd] = blob[(size_t)L.h2 + ((size_t)j * D + dd) * S + s0];
                }
            }
            st.stage_chunks = T.staged_chunks(env->tab.overflow ? S : nst);
            HIP_TRY_D(hipMemcpy(d_blob, sb.data(), sb.size(), hipMemcpyHostToDevice));
        }
    }
    if (st.totals) HIP_TRY_D(hipMemset(st.totals, 0, (size_t)st.n_slots * GW_T_COUNT * sizeof(unsigned long long)));
    if (st.runs) HIP_TRY_D(hipMemset(st.runs, 0, (size_t)N * D * GW_RING_PHYS * sizeof(uint64_t)));
    if (st.bph) HIP_TRY_D(hipMemset(st.bph, 0, (size_t)N * GW_RING_PHYS * sizeof(GwBp)));
    if (env->dyn) {
        std::vector<double> prx((size_t)R * R, 0.0), pos((size_t)R * 2), ext((size_t)R * R, 0.0);
        for (int a = 0; a < R; ++a) {
            pos[(size_t)a * 2] = cfg->pos[a][0]; pos[(size_t)a * 2 + 1] = cfg->pos[a][1];
            for (int b = 0; b < R; ++b) { prx[(size_t)a * R + b] = env->tab.prx[a][b]; ext[(size_t)a * R + b] = cfg->extra_att_db[a][b]; }
        }
        HIP_TRY_D(hipMemcpy(d_prx, prx.data(), prx.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY_D(hipMemcpy(d_pos, pos.data(), pos.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY_D(hipMemcpy(d_extra, ext.data(), ext.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (st.ip) {                                    // the per-step kernels read both structs from the header of the ip block
        HIP_TRY_D(hipMemcpy(d_blob + gw_hdr_cst_off(D), &k, sizeof k, hipMemcpyHostToDevice));
        HIP_TRY_D(hipMemcpy(d_blob + gw_hdr_st_off(D), &st, sizeof st, hipMemcpyHostToDevice));
    }
    rc = explicit_q ? gw_launch_init(st, nullptr) : gw_launch_init_sfx(st, nullptr);
    if (!rc && env->dyn) rc = gw_launch_init_dyn(st, k, env->tab.thermal, nullptr);
    if (rc) { rc = fail(GW_EHIP, "init kernel launch failed"); gw_destroy(env); return rc; }
    HIP_TRY_D(hipDeviceSynchronize());
#undef HIP_TRY_D
    env->nblocks_create = env->nblocks;
    *out = env;
    return GW_OK;
}

int gw_destroy(gw_env* env)
{
    if (!env) return GW_OK;
    (void)hipSetDevice(env->cfg.hip_device);
    for (int i = 0; i < env->nblocks; ++i) (void)hipFree(env->blocks[i]);
    delete env;
    return GW_OK;
}

int gw_reset(gw_env* env, const uint8_t* mask_dev, int32_t* obs_dev, void* stream)
{
    if (!env) return fail(GW_EINVAL, "env is NULL");
    int rc = select_device(env);
    if (rc) return rc;
    if (env->st.tk ? gw_launch_reset_sfx(env->st, mask_dev, obs_dev, stream)
                    : gw_launch_reset(env->st, mask_dev, obs_dev, stream))
        return fail(GW_EHIP, "reset kernel launch failed");
    return GW_OK;
}

int gw_step(gw_env* env, const int32_t* device_dev, const int32_t* duration_dev,
            int32_t* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream)
{
    if (!env) return fail(GW_EINVAL, "env is NULL");
    if (!device_dev || !duration_dev || !obs_dev || !reward_dev || !done_dev)
        return fail(GW_EINVAL, "gw_step: NULL device pointer");
    int rc = select_device(env);
    if (rc) return rc;
    if (launch_step(env, device_dev, duration_dev, obs_dev, reward_dev, done_dev, stream))
        return fail(GW_EHIP, "step kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    gw_env_add_steps(env, 1);
    return GW_OK;
}

int gw_step_fb(gw_env* env, const int32_t* device_dev, const int32_t* duration_dev,
               int32_t* obs_dev, float* reward_dev, uint8_t* done_dev, uint8_t* feedback_byte_dev, void* stream)
{
    if (!feedback_byte_dev) return gw_step(env, device_dev, duration_dev, obs_dev, reward_dev, done_dev, stream);
    if (!env) return fail(GW_EINVAL, "env is NULL");
    if (!device_dev || !duration_dev || !obs_dev || !reward_dev || !done_dev)
        return fail(GW_EINVAL, "gw_step_fb: NULL device pointer");
    int rc = select_device(env);
    if (rc) return rc;
    bool fused = false;
    if (launch_step(env, device_dev, duration_dev, obs_dev, reward_dev, done_dev, stream, feedback_byte_dev, &fused))
        return fail(GW_EHIP, "step kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    gw_env_add_steps(env, 1);
    if (!fused)                                     // generic / live-PHY kernels: the packing kernel on this step's row
        return gw_pack_feedback(env, env->st.N, obs_dev, reward_dev, done_dev, feedback_byte_dev, 0, stream);
    return GW_OK;
}

int gw_rollout(gw_env* env, int32_t steps, const int32_t* device_dev, const int32_t* duration_dev,
               int32_t* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream)
{
    if (!env) return fail(GW_EINVAL, "env is NULL");
    if (steps < 0) return fail(GW_EINVAL, "steps < 0");
    if (!device_dev || !duration_dev || !obs_dev || !reward_dev || !done_dev)
        return fail(GW_EINVAL, "gw_rollout: NULL device pointer");
    int rc = select_device(env);
    if (rc) return rc;
    const int64_t N = env->st.N;
    int32_t s = 0;
    // fused persistent rollout (ct_rollout_sfx.hip) in chunks of up to rcap steps, when this D has one
    while (env->st.tk && !env->dyn && env->st.rcap > 0 && s < steps) {
        const int32_t chunk = steps - s < env->st.rcap ? steps - s : env->st.rcap;
        const int64_t o = (int64_t)s * N;
        rc = gw_launch_rollout_sfx(env->st, env->cst_host, chunk, device_dev + o, duration_dev + o, obs_dev + o,
                                   reward_dev + o, done_dev + o, env->st.ract, env->st.rfb, env->st.rcap, stream,
                                   gw_env_below_limits(env, stream) && env->t_bound + (double)(chunk + 1) * env->step_max < env->t_limit);
        if (rc == GW_EUNSUPPORTED) {
            if (getenv("GW_ROLLOUT_STRICT")) return fail(GW_EUNSUPPORTED, "no fused rollout for this handle (GW_ROLLOUT_STRICT is set)");
            break;                                               // (steps > rollout capacity 0, max_duration > 254)
        }
        if (rc) return fail(GW_EHIP, "rollout kernel launch failed at step %d", s);
        s += chunk;
        gw_env_add_steps(env, (uint64_t)chunk);
    }
    for (; s < steps; ++s) {                                   // generic path: one step launch per step
        const int64_t o = (int64_t)s * N;
        if (launch_step(env, device_dev + o, duration_dev + o, obs_dev + o, reward_dev + o, done_dev + o, stream))
            return fail(GW_EHIP, "step kernel launch failed at step %d", s);
        gw_env_add_steps(env, 1);
    }
    return GW_OK;
}

int gw_delivered(gw_env* env, uint32_t* out_dev, void* stream)
{
    if (!env || !out_dev) return fail(GW_EINVAL, "env/out is NULL");
    if (!env->st.sa) return fail(GW_EUNSUPPORTED, "gw_delivered needs the default (suffix) mode");
    int rc = select_device(env);
    if (rc) return rc;
    if (gw_launch_delivered_sfx(env->st, out_dev, stream)) return fail(GW_EHIP, "delivered kernel launch failed");
    return GW_OK;
}

int gw_enqueue(gw_env* env, int32_t sender, const int32_t* payload_bytes_dev, void* stream)
{
    if (!env || !payload_bytes_dev) return fail(GW_EINVAL, "env/payload_bytes is NULL");
    if (!env->st.runs) return fail(GW_EUNSUPPORTED, "gw_enqueue needs GW_CFG_EXPLICIT_QUEUE");
    if (sender < 0 || sender >= env->st.D) return fail(GW_EINVAL, "sender out of range");
    int rc = select_device(env);
    if (rc) return rc;
    if (gw_launch_enqueue(env->st, sender, payload_bytes_dev, stream)) return fail(GW_EHIP, "enqueue kernel launch failed");
    return GW_OK;
}

int gw_pack_feedback(gw_env* env, int64_t count, const int32_t* obs_dev, const float* reward_dev, const uint8_t* done_dev,
                     uint8_t* packed_dev, int32_t check, void* stream)
{
    if (!env || count < 0) return fail(GW_EINVAL, "env is NULL or count negative");
    if (count > 0 && (!obs_dev || !reward_dev || !done_dev || !packed_dev)) return fail(GW_EINVAL, "a buffer is NULL");
    int rc = select_device(env);
    if (rc) return rc;
    if (!env->pack_bad) {
        if ((rc = dev_alloc(env, &env->pack_bad, 1))) return rc;
        HIP_TRY(hipMemset(env->pack_bad, 0, sizeof(uint32_t)));
    }
    if (count > 0 && gw_launch_pack_feedback(count, env->cfg.counter_bound, env->cfg.payload_value, obs_dev, reward_dev, done_dev,
                                             packed_dev, env->pack_bad, stream))
        return fail(GW_EHIP, "pack kernel launch failed");
    if (check) {
        uint32_t bad = 0;
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        HIP_TRY(hipMemcpy(&bad, env->pack_bad, sizeof bad, hipMemcpyDeviceToHost));
        if (bad) return fail(GW_EINVAL, "%u element group(s) were not the built-in interpreter's feedback (not representable in one byte)", bad);
    }
    return GW_OK;
}

int gw_unpack_feedback(gw_env* env, int64_t count, const uint8_t* packed_dev, int32_t* obs_dev, float* reward_dev,
                       uint8_t* done_dev, void* stream)
{
    if (!env || count < 0) return fail(GW_EINVAL, "env is NULL or count negative");
    if (count == 0) return GW_OK;
    if (!obs_dev || !reward_dev || !done_dev || !packed_dev) return fail(GW_EINVAL, "a buffer is NULL");
    int rc = select_device(env);
    if (rc) return rc;
    if (gw_launch_unpack_feedback(count, env->cfg.counter_bound, env->cfg.payload_value, packed_dev, obs_dev, reward_dev, done_dev, stream))
        return fail(GW_EHIP, "unpack kernel launch failed");
    return GW_OK;
}

int gw_set_position(gw_env* env, int32_t radio, const double* x_dev, const double* y_dev, const uint8_t* mask_dev, void* stream)
{
    if (!env || !x_dev || !y_dev) return fail(GW_EINVAL, "env/x/y is NULL");
    if (!env->st.prx_env) return fail(GW_EUNSUPPORTED, "gw_set_position needs GW_CFG_PER_ENV_GEOMETRY");
    if (radio < 0 || radio > env->st.D) return fail(GW_EINVAL, "radio index out of range");
    int rc = select_device(env);
    if (rc) return rc;
    if (gw_launch_set_position(env->st, env->cst_host, radio, x_dev, y_dev, nullptr, mask_dev, stream))
        return fail(GW_EHIP, "set_position kernel launch failed");
    return GW_OK;
}

int gw_set_positions(gw_env* env, const double* pos_dev, const uint8_t* mask_dev, void* stream)
{
    if (!env || !pos_dev) return fail(GW_EINVAL, "env/pos is NULL");
    if (!env->st.prx_env) return fail(GW_EUNSUPPORTED, "gw_set_positions needs GW_CFG_PER_ENV_GEOMETRY");
    int rc = select_device(env);
    if (rc) return rc;
    if (gw_launch_set_position(env->st, env->cst_host, -1, nullptr, nullptr, pos_dev, mask_dev, stream))
        return fail(GW_EHIP, "set_position kernel launch failed");
    return GW_OK;
}

int gw_received(gw_env* env, int32_t* out_dev, void* stream)
{
    if (!env || !out_dev) return fail(GW_EINVAL, "env/out is NULL");
    int rc = select_device(env);
    if (rc) return rc;
    if (env->st.tk ? gw_launch_received_sfx(env->st, out_dev, stream) : gw_launch_received(env->st, out_dev, stream))
        return fail(GW_EHIP, "received kernel launch failed");
    return GW_OK;
}

int gw_clear_flags(gw_env* env, void* stream)
{
    if (!env) return fail(GW_EINVAL, "env is NULL");
    int rc = select_device(env);
    if (rc) return rc;
    if (gw_launch_clear_flags(env->st, stream)) return fail(GW_EHIP, "clear_flags kernel launch failed");
    return GW_OK;
}

int gw_stats_read(gw_env* env, gw_stats* out)
{
    if (!env || !out) return fail(GW_EINVAL, "env/out is NULL");
    int rc = select_device(env);
    if (rc) return rc;
    unsigned long long t[GW_T_COUNT] = {0};
    HIP_TRY(hipDeviceSynchronize());
    if (env->st.sa) {                                 // default mode: sum the per-env counts
        GwEnvCounts c;
        if ((rc = derive_env_counts(env, c))) return rc;
        memset(out, 0, sizeof *out);
        for (int64_t e = 0; e < env->st.N; ++e) {
            out->steps += c.steps[e]; out->transmissions += c.tx[e]; out->delivered += c.deliv[e]; out->appended += c.app[e];
            out->popped += c.pop[e]; out->dropped += c.drop[e]; out->bad_actions += c.bad[e]; out->flags_or |= c.flags[e];
        }
        return GW_OK;
    }
    std::vector<unsigned long long> slots((size_t)env->st.n_slots * GW_T_COUNT);
    HIP_TRY(hipMemcpy(slots.data(), env->st.totals, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int64_t w = 0; w < env->st.n_slots; ++w)
        for (int j = 0; j < GW_T_COUNT; ++j) {
            if (j == GW_T_FLAGS) t[j] |= slots[(size_t)w * GW_T_COUNT + j];
            else t[j] += slots[(size_t)w * GW_T_COUNT + j];
        }
    out->steps = t[GW_T_STEPS]; out->transmissions = t[GW_T_TX]; out->delivered = t[GW_T_DELIV];
    out->appended = t[GW_T_APP]; out->popped = t[GW_T_POP]; out->dropped = t[GW_T_DROP];
    out->flags_or = t[GW_T_FLAGS]; out->bad_actions = t[GW_T_BAD];
    return GW_OK;
}

int gw_now_ptr(gw_env* env, const void** now_dev, int64_t* stride_bytes)
{
    if (!env || !now_dev || !stride_bytes) return fail(GW_EINVAL, "env/now/stride is NULL");
    if (env->st.tw) { *now_dev = env->st.tw; *stride_bytes = 16; }          // {now, next tick} records
    else { *now_dev = env->st.xw; *stride_bytes = 16; }                    // explicit-queue mode: the same record shape
    return GW_OK;
}

// ---- checkpoint / restore (the reference cannot snapshot its SimPy generators; here an env's state is a
// handful of arrays in HBM).  A snapshot is every device block the handle has held since gw_create, in allocation order, behind a
// header that carries the configuration it belongs to.
struct GwSnapHeader {
    uint32_t magic, abi;
    uint64_t total;
    gw_config cfg;
    int32_t nblocks, dyn;
    uint64_t block_bytes[32];
    double t_bound;
};
static const uint32_t kSnapMagic = 0x4e535747u;            // "GWSN"

int gw_snapshot_bytes(gw_env* env, uint64_t* bytes)
{
    if (!env || !bytes) return fail(GW_EINVAL, "env/bytes is NULL");
    uint64_t t = sizeof(GwSnapHeader);
    for (int i = 0; i < env->nblocks_create; ++i) t += env->block_bytes[i];
    *bytes = t;
    return GW_OK;
}

int gw_get_snapshot(gw_env* env, void* dst, uint64_t bytes)
{
    if (!env || !dst) return fail(GW_EINVAL, "env/dst is NULL");
    uint64_t need = 0;
    int rc = gw_snapshot_bytes(env, &need);
    if (rc) return rc;
    if (bytes != need) return fail(GW_EINVAL, "snapshot needs %llu bytes, got %llu", (unsigned long long)need, (unsigned long long)bytes);
    if ((rc = select_device(env))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    GwSnapHeader h;
    memset(&h, 0, sizeof h);
    h.magic = kSnapMagic; h.abi = GW_ABI_VERSION; h.total = need; h.cfg = env->cfg; h.nblocks = env->nblocks_create; h.dyn = env->dyn;
    for (int i = 0; i < env->nblocks_create; ++i) h.block_bytes[i] = env->block_bytes[i];
    h.t_bound = env->t_bound;
    uint8_t* o = (uint8_t*)dst;
    memcpy(o, &h, sizeof h);
    o += sizeof h;
    for (int i = 0; i < env->nblocks_create; ++i) {
        HIP_TRY(hipMemcpy(o, env->blocks[i], env->block_bytes[i], hipMemcpyDeviceToHost));
        o += env->block_bytes[i];
    }
    return GW_OK;
}

// Restore a snapshot into a handle created with the SAME gw_config (the same handle later on, or a fresh one -- on any GPU:
// hip_device is not compared).  Every later step continues bit for bit as the snapshotted handle would have.
int gw_set_state(gw_env* env, const void* src, uint64_t bytes)
{
    if (!env || !src) return fail(GW_EINVAL, "env/src is NULL");
    if (bytes < sizeof(GwSnapHeader)) return fail(GW_EINVAL, "not a snapshot (too short)");
    GwSnapHeader h;
    memcpy(&h, src, sizeof h);
    if (h.magic != kSnapMagic || h.abi != (uint32_t)GW_ABI_VERSION || h.total != bytes) return fail(GW_EINVAL, "not a snapshot of this ABI");
    gw_config a = h.cfg, b = env->cfg;
    a.hip_device = b.hip_device = 0;
    if (memcmp(&a, &b, sizeof a) != 0 || h.nblocks != env->nblocks_create || h.dyn != env->dyn)
        return fail(GW_EINVAL, "snapshot belongs to a handle with another configuration");
    for (int i = 0; i < h.nblocks; ++i)
        if (h.block_bytes[i] != env->block_bytes[i]) return fail(GW_EINVAL, "snapshot belongs to a handle with another layout");
    int rc = select_device(env);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    const uint8_t* in = (const uint8_t*)src + sizeof h;
    for (int i = 0; i < h.nblocks; ++i) {
        HIP_TRY(hipMemcpy(env->blocks[i], in, env->block_bytes[i], hipMemcpyHostToDevice));
        in += env->block_bytes[i];
    }
    // the header in front of the `ip` records holds THIS handle's device pointers: put them back
    if (env->st.ip) {
        uint8_t* blob = const_cast<uint8_t*>(env->st.blob);
        const int D = env->st.D;
        HIP_TRY(hipMemcpy(blob + gw_hdr_cst_off(D), &env->cst_host, sizeof env->cst_host, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(blob + gw_hdr_st_off(D), &env->st, sizeof env->st, hipMemcpyHostToDevice));
    }
    env->t_bound = h.t_bound > env->t_bound ? h.t_bound : env->t_bound;      // (an upper bound either way)
    HIP_TRY(hipDeviceSynchronize());
    return GW_OK;
}

int gw_state_bytes(gw_env* env, uint64_t* bytes)
{
    if (!env || !bytes) return fail(GW_EINVAL, "env/bytes is NULL");
    *bytes = env->bytes;
    return GW_OK;
}

int gw_link_info(gw_env* env, int32_t from, int32_t to, double* att, double* prx)
{
    if (!env) return fail(GW_EINVAL, "env is NULL");
    const int R = env->tab.R;
    if (from < 0 || from >= R || to < 0 || to >= R || from == to) return fail(GW_EINVAL, "radio index out of range");
    if (att) *att = env->tab.att[from][to];
    if (prx) *prx = env->tab.prx[from][to];
    return GW_OK;
}

int gw_noise_states(gw_env* env, int32_t radio, int32_t* count, double* values)
{
    if (!env || !count) return fail(GW_EINVAL, "env/count is NULL");
    if (radio < 0 || radio >= env->tab.R) return fail(GW_EINVAL, "radio index out of range");
    *count = env->tab.nstates[radio];
    if (values) for (int i = 0; i < GW_MAX_NSTATES; ++i) values[i] = i < *count ? env->tab.state_val[radio][i] : 0.0;
    return GW_OK;
}

// Host-only: which exact fast paths gw_create would enable for cfg (bit 0 fmod, 1 division,
// 2 decision, 3 idempotent noise states), and the number of noise states per radio.
int gw_selftest_fastmath(const gw_config* cfg, int32_t* max_noise_states)
{
    if (!cfg) return fail(GW_EINVAL, "cfg is NULL");
    int rc = validate(*cfg);
    if (rc) return rc;
    GwHostTables* tab = new GwHostTables();
    char msg[256] = "";
    rc = gw_build_tables(*cfg, *tab, msg, sizeof msg);
    if (rc) { delete tab; return fail(rc, "%s", msg); }
    GwDevConst k;
    memset(&k, 0, sizeof k);
    set_fast_paths(*cfg, *tab, k);
    int mx = 0;
    for (int r = 0; r < tab->R; ++r) mx = tab->nstates[r] > mx ? tab->nstates[r] : mx;
    if (tab->overflow) mx = GW_MAX_NSTATES + 1;           // no finite state set: the live-PHY kernel takes this geometry
    if (max_noise_states) *max_noise_states = mx;
    delete tab;
    return (k.fast_fmod ? 1 : 0) | (k.fast_div ? 2 : 0) | (k.fast_decide ? 4 : 0) | (k.idem_states ? 8 : 0) | (k.fast_ticks ? 16 : 0);
}

// Host-only fuzz of the suffix queue encoding (gw_queue.h, the same code the kernel runs) against
// explicit deque(maxlen=GW_QUEUE_CAP) objects, one per sender: random ticks, resets and pops.
// Returns the number of mismatching operations (0 = identical), negative on bad arguments.
int gw_selftest_queue(uint64_t seed, int32_t operations, int32_t mult, int32_t counter_bound)
{
    if (operations < 0 || mult < 1 || mult > GW_MAX_MULT || counter_bound < 1)
        return fail(GW_EINVAL, "gw_selftest_queue: bad arguments");
    uint64_t x = seed * 6364136223846793005ull + 1442695040888963407ull;
    auto rnd = [&x](uint32_t n) {                      // xorshift64*, test-only
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        return (uint32_t)(((x * 2685821657736338717ull) >> 33) % n);
    };
    const uint32_t bound = (uint32_t)counter_bound, base = 25u;
    const uint32_t mults[2] = {(uint32_t)mult, mult > 1 ? 1u : 3u};   // two senders sharing the env's ticks
    uint32_t inv16[2];
    for (int i = 0; i < 2; ++i) inv16[i] = (65536u + mults[i] - 1u) / mults[i];
    std::vector<GwBp> hist(GW_RING_PHYS);
    GwBp cur = {0u, 1u}, prev = cur;
    hist[0] = cur;
    uint32_t nbp = 1, tau = 0, len[2] = {0, 0}, bad = 0, ctr = 1;
    std::deque<uint32_t> ref[2];
    GwTally tally = {0, 0, 0, 0, 0};
    uint64_t ref_drops = 0;
    for (int32_t op = 0; op < operations; ++op) {
        const uint32_t what = rnd(16);
        if (what < 9) {                                 // k counter ticks (counter_traffic.py:53-61)
            const uint32_t kk = 1 + rnd(what < 5 ? 3 : 22);
            for (uint32_t t = 0; t < kk; ++t) {
                for (int i = 0; i < 2; ++i)
                    for (uint32_t j = 0; j < mults[i]; ++j) {
                        if (ref[i].size() == GW_QUEUE_CAP) { ref[i].pop_front(); ++ref_drops; }
                        ref[i].push_back(base + ctr);
                    }
                if (ctr < bound) ++ctr;
            }
            tau += kk;
            for (int i = 0; i < 2; ++i) len[i] = gw_len_after_ticks(len[i], kk, mults[i], tally);
        } else if (what < 12) {                         // reset(): counter_traffic.py:139-140
            ctr = 0;
            if (cur.t0 == tau) { cur.c0 = 0; hist[(nbp - 1) & GW_RING_MASK] = cur; }
            else { prev = cur; cur.t0 = tau; cur.c0 = 0; hist[nbp & GW_RING_MASK] = cur; ++nbp; }
        } else {                                        // pops from one sender
            const int i = (int)rnd(2);
            uint32_t n = 1 + rnd(what == 15 ? 40 : 4);
            while (n-- && !ref[i].empty()) {
                const uint32_t age = gw_ceil_div(len[i], mults[i], inv16[i]);
                const uint32_t hv = base + gw_tick_value(tau - age, cur, prev, nbp, hist.data(), bound);
                if (len[i] != ref[i].size() || hv != ref[i].front()) ++bad;
                ref[i].pop_front();
                len[i]--;
            }
        }
        if (gw_min_u32(cur.c0 + (tau - cur.t0), bound) != ctr) ++bad;       // derived sender.counter
        for (int i = 0; i < 2; ++i) {
            uint32_t out[GW_QUEUE_CAP];
            bool same = len[i] == ref[i].size();
            if (same) {
                expand_sfx(bound, base, mults[i], len[i], tau, cur, prev, nbp, hist.data(), out);
                for (uint32_t p = 0; same && p < len[i]; ++p) same = out[p] == ref[i][p];
            }
            if (!same) ++bad;
        }
    }
    if (tally.drop != ref_drops) ++bad;
    return (int)bad;
}

// Host-only fuzz of the run-length queues (gw_runq.h, the same code the generic kernel runs) against an explicit
// deque(maxlen=GW_QUEUE_CAP): random counter ticks (with a saturating counter and resets), pops, and literal packets.
// Returns the number of mismatches (0 = identical), negative on bad arguments.
int gw_selftest_runq(uint64_t seed, int32_t operations, int32_t mult, int32_t counter_bound)
{
    if (operations < 0 || mult < 1 || mult > GW_QUEUE_CAP || counter_bound < 1) return fail(GW_EINVAL, "gw_selftest_runq: bad arguments");
    uint64_t x = seed * 6364136223846793005ull + 1442695040888963407ull;
    auto rnd = [&x](uint32_t n) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        return (uint32_t)(((x * 2685821657736338717ull) >> 33) % n);
    };
    const uint32_t m = (uint32_t)mult, bound = (uint32_t)counter_bound, base = 25u, cap = base + bound;
    const uint32_t inv20 = ((1u << 20) + m - 1u) / m;
    std::vector<uint64_t> ring(GW_RING_PHYS, 0);
    GwRec zero = {0u, 0u, 0u, 0u};
    GwRunQ q = gw_runq_unpack(zero);
    std::deque<uint32_t> ref;
    GwTally tally = {0, 0, 0, 0, 0};
    uint64_t ref_drops = 0, ref_apps = 0;
    uint32_t ctr = 1, bad = 0;
    int big_left = 6;
    for (int32_t op = 0; op < operations; ++op) {
        const uint32_t what = rnd(16);
        if (what < 8) {                                   // k counter ticks
            // mostly a step's worth of ticks; now and then a LONG step (a counter interval far below the step length):
            // hundreds to tens of thousands of ticks in one call
            uint32_t k = 1 + rnd(what < 5 ? 3 : 22);
            if (what == 7 && rnd(4) == 0) {
                k = 90 + rnd(400);
                if (big_left > 0 && rnd(4) == 0) { k = 4000 + rnd(26000); --big_left; }   // (the reference deque pays k * mult pushes)
            }
            uint32_t c = ctr;
            for (uint32_t t = 0; t < k; ++t) {
                for (uint32_t j = 0; j < m; ++j) {
                    if (ref.size() == GW_QUEUE_CAP) { ref.pop_front(); ++ref_drops; }
                    ref.push_back(base + c);
                    ++ref_apps;
                }
                if (c < bound) ++c;
            }
            gw_runq_ticks(q, k, ctr, bound, base, ring.data(), m, inv20, tally);
            ctr = c;
        } else if (what < 10) {                           // reset(): counters restart
            ctr = 0;
        } else if (what < 12) {                           // SimpleNetworkDevice.send of an arbitrary packet
            const uint32_t size = 1 + rnd(70000);
            if (ref.size() == GW_QUEUE_CAP) { ref.pop_front(); gw_runq_pop_front(q, 1u, ring.data(), m, inv20, cap); }
            ref.push_back(size);
            gw_runq_append_literal(q, size, ring.data());
        } else if (what < 14) {                           // window pops
            uint32_t n = 1 + rnd(what == 13 ? 60 : 4);
            while (n-- && !ref.empty()) {
                if (q.len != ref.size() || q.state == 0u || q.H.v0 != ref.front()) ++bad;
                ref.pop_front();
                gw_runq_pop_front(q, 1u, ring.data(), m, inv20, cap);
            }
        } else {                                          // a window: pop one packet + the ticks inside its transmission, fused
            uint32_t n = 1 + rnd(9);
            while (n-- && !ref.empty()) {
                if (q.len != ref.size() || q.state == 0u || q.H.v0 != ref.front()) ++bad;
                ref.pop_front();
                const uint32_t k = rnd(5);                // 0..4 ticks
                uint32_t c = ctr;
                for (uint32_t t = 0; t < k; ++t) {
                    for (uint32_t j = 0; j < m; ++j) {
                        if (ref.size() == GW_QUEUE_CAP) { ref.pop_front(); ++ref_drops; }
                        ref.push_back(base + c);
                        ++ref_apps;
                    }
                    if (c < bound) ++c;
                }
                gw_runq_pop1_ticks(q, k, ctr, bound, base, ring.data(), m, inv20, tally);
                ctr = c;
            }
        }
        q = gw_runq_unpack(gw_runq_pack(q));              // through the 16-byte record, as between two steps
        uint32_t out[GW_QUEUE_CAP];
        const uint32_t n = gw_runq_expand(q, ring.data(), m, cap, out);
        bool same = n == ref.size() && q.len == ref.size();
        for (uint32_t p = 0; same && p < n; ++p) same = out[p] == ref[p];
        if (!same) ++bad;
    }
    if (tally.drop != ref_drops || tally.app != ref_apps) ++bad;
    return (int)bad;
}

// per-env positions [N][R][2] / link powers [N][R][R] (from -> to) of a GW_CFG_PER_ENV_GEOMETRY handle, logical layout
static int read_geometry(const GwState& st, bool positions, void* dst, size_t bytes, const char* field)
{
    const int64_t N = st.N;
    const int R = st.R, RP = gw_rp(R);
    const int per = positions ? R * 2 : R * R;
    if (bytes != (size_t)N * per * sizeof(double))
        return fail(GW_EFIELD, "field %s needs %zu bytes, got %zu", field, (size_t)N * per * sizeof(double), bytes);
    double* o = (double*)dst;
    if (positions) { HIP_TRY(hipMemcpy(o, st.pos_env, bytes, hipMemcpyDeviceToHost)); return GW_OK; }
    std::vector<double> v((size_t)N * R * RP);
    HIP_TRY(hipMemcpy(v.data(), st.prx_env, v.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t e = 0; e < N; ++e)
        for (int a = 0; a < R; ++a) for (int b = 0; b < R; ++b) o[(e * R + a) * R + b] = v[((size_t)e * R + a) * RP + b];
    return GW_OK;
}

int gw_get_state(gw_env* env, const char* field, void* dst, size_t bytes)
{
    if (!env || !field || !dst) return fail(GW_EINVAL, "env/field/dst is NULL");
    int rc = select_device(env);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    const GwState& st = env->st;
    const int64_t N = st.N;
    const int D = st.D, R = st.R;

#define NEED(count, type) do { if (bytes != (size_t)(count) * sizeof(type)) \
        return fail(GW_EFIELD, "field %s needs %zu bytes, got %zu", field, (size_t)(count) * sizeof(type), bytes); } while (0)

#ifdef GW_STAMPS
    if (!strcmp(field, "stamps")) {
        NEED(st.n_slots * 16, uint64_t);
        HIP_TRY(hipMemcpy(dst, st.stamps, bytes, hipMemcpyDeviceToHost));
        return GW_OK;
    }
#endif
    if (st.tk) {                                     // ---- suffix mode: packed records (ct_step_sfx.hip) ----
        {
            static const char* names[6] = {"n_tx", "n_delivered", "n_appended", "n_popped", "n_dropped", "flags"};
            for (int k = 0; k < 6; ++k)
                if (!strcmp(field, names[k])) {
                    GwEnvCounts c;
                    if ((rc = derive_env_counts(env, c))) return rc;
                    if (k == 5) { NEED(N, uint32_t); memcpy(dst, c.flags.data(), bytes); return GW_OK; }
                    NEED(N, uint64_t);
                    const std::vector<uint64_t>& v = k == 0 ? c.tx : (k == 1 ? c.deliv : (k == 2 ? c.app : (k == 3 ? c.pop : c.drop)));
                    memcpy(dst, v.data(), bytes);
                    return GW_OK;
                }
        }
        const int RB = st.RB;
        std::vector<double> tw((size_t)N * 2);
        std::vector<uint32_t> tk((size_t)N * 4), ip((size_t)N * 4);
        std::vector<uint8_t> qb((size_t)N * RB);
        HIP_TRY(hipMemcpy(tw.data(), st.tw, tw.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(tk.data(), st.tk, tk.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ip.data(), st.ip, ip.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(qb.data(), st.qb, qb.size(), hipMemcpyDeviceToHost));
        const uint32_t bound = (uint32_t)env->cfg.counter_bound;
        const int pv = env->cfg.payload_value;
        if (!strcmp(field, "now")) { NEED(N, double); double* o = (double*)dst; for (int64_t e = 0; e < N; ++e) o[e] = tw[e * 2]; return GW_OK; }
        if (!strcmp(field, "wake")) {
            NEED(N * D, double); double* o = (double*)dst;
            for (int64_t e = 0; e < N; ++e) for (int i = 0; i < D; ++i) o[e * D + i] = tw[e * 2 + 1];
            return GW_OK;
        }
        if (!strcmp(field, "counter")) {             // sender.counter == value of the next tick
            NEED(N * D, uint32_t); uint32_t* o = (uint32_t*)dst;
            for (int64_t e = 0; e < N; ++e) {
                const uint32_t v = gw_min_u32(ip[e * 4 + 1] + (tk[e * 4] - ip[e * 4]), bound);   // newest breakpoint: ip {t0, c0}
                for (int i = 0; i < D; ++i) o[e * D + i] = v;
            }
            return GW_OK;
        }
        if (!strcmp(field, "last_abs")) { NEED(N, int32_t); int32_t* o = (int32_t*)dst; for (int64_t e = 0; e < N; ++e) o[e] = (int32_t)(tk[e * 4 + 3] & 0x7fffffffu); return GW_OK; }
        if (!strcmp(field, "latest_diff")) {
            NEED(N, int32_t); int32_t* o = (int32_t*)dst;
            for (int64_t e = 0; e < N; ++e) { const uint32_t m = tk[e * 4 + 2]; o[e] = pv * ((int)(m & 1u) - (int)((m >> 1) & 1u)); }
            return GW_OK;
        }
        if (!strcmp(field, "received")) {
            NEED(N * D, int32_t); int32_t* o = (int32_t*)dst;
            for (int64_t e = 0; e < N; ++e) for (int i = 0; i < D; ++i) o[e * D + i] = ((tk[e * 4 + 2] >> i) & 1u) ? pv : 0;
            return GW_OK;
        }
        if (!strcmp(field, "qlen")) {
            NEED(N * D, int32_t); int32_t* o = (int32_t*)dst;
            for (int64_t e = 0; e < N; ++e) for (int i = 0; i < D; ++i) o[e * D + i] = qb[(size_t)e * RB + i];
            return GW_OK;
        }
        if (!strcmp(field, "rx_power")) {
            NEED(N * R, double); double* o = (double*)dst;
            if (st.rxp) {                            // live-PHY mode: the f64 itself, rows of gw_rp(R) doubles per env
                const int RP = gw_rp(R);
                std::vector<double> rx((size_t)N * RP);
                HIP_TRY(hipMemcpy(rx.data(), st.rxp, rx.size() * sizeof(double), hipMemcpyDeviceToHost));
                for (int64_t e = 0; e < N; ++e) for (int r = 0; r < R; ++r) o[e * R + r] = rx[(size_t)e * RP + r];
                return GW_OK;
            }
            for (int64_t e = 0; e < N; ++e) for (int r = 0; r < R; ++r) o[e * R + r] = env->tab.state_val[r][qb[(size_t)e * RB + D + r]];
            return GW_OK;
        }
        if (!strcmp(field, "pos") || !strcmp(field, "link_power")) {
            if (!st.prx_env) return fail(GW_EFIELD, "field %s needs GW_CFG_PER_ENV_GEOMETRY", field);
            return read_geometry(st, field[0] == 'p', dst, bytes, field);
        }
        if (!strcmp(field, "queue")) {
            NEED(N * D * GW_QUEUE_CAP, uint32_t);
            std::vector<GwBp> hist((size_t)N * GW_RING_PHYS);
            HIP_TRY(hipMemcpy(hist.data(), st.bph, hist.size() * sizeof(GwBp), hipMemcpyDeviceToHost));
            uint32_t* o = (uint32_t*)dst;
            memset(o, 0, bytes);
            const uint32_t base = (uint32_t)(env->cfg.mac_header_bytes + env->cfg.net_header_bytes);
            for (int64_t e = 0; e < N; ++e) {
                GwBp cur, prev;
                cur.t0 = ip[e * 4]; cur.c0 = ip[e * 4 + 1];
                prev.t0 = ip[e * 4 + 2]; prev.c0 = ip[e * 4 + 3];
                for (int i = 0; i < D; ++i)
                    expand_sfx(bound, base, (uint32_t)env->cfg.mult[i], qb[(size_t)e * RB + i], tk[e * 4], cur, prev,
                               tk[e * 4 + 1], &hist[(size_t)e * GW_RING_PHYS], o + ((size_t)e * D + i) * GW_QUEUE_CAP);
            }
            return GW_OK;
        }
        return fail(GW_EFIELD, "unknown field %s", field);
    }

    // ---- explicit-queue mode: packed records xw {now, wake}, xc {counter, rvmask, last_abs | done << 31, flags} ----
    if (!strcmp(field, "now") || !strcmp(field, "wake")) {
        std::vector<double> w((size_t)N * 2);
        HIP_TRY(hipMemcpy(w.data(), st.xw, w.size() * sizeof(double), hipMemcpyDeviceToHost));
        double* o = (double*)dst;
        if (field[0] == 'n') { NEED(N, double); for (int64_t e = 0; e < N; ++e) o[e] = w[e * 2]; }
        else { NEED(N * D, double); for (int64_t e = 0; e < N; ++e) for (int i = 0; i < D; ++i) o[e * D + i] = w[e * 2 + 1]; }
        return GW_OK;
    }
    if (!strcmp(field, "last_abs") || !strcmp(field, "flags") || !strcmp(field, "counter") || !strcmp(field, "received") ||
        !strcmp(field, "latest_diff")) {
        std::vector<uint32_t> c((size_t)N * 4);
        HIP_TRY(hipMemcpy(c.data(), st.xc, c.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const int pv = env->cfg.payload_value;
        if (!strcmp(field, "last_abs")) { NEED(N, int32_t); int32_t* o = (int32_t*)dst; for (int64_t e = 0; e < N; ++e) o[e] = (int32_t)(c[e * 4 + 2] & 0x7fffffffu); }
        else if (!strcmp(field, "flags")) { NEED(N, uint32_t); uint32_t* o = (uint32_t*)dst; for (int64_t e = 0; e < N; ++e) o[e] = c[e * 4 + 3]; }
        else if (!strcmp(field, "counter")) {
            NEED(N * D, uint32_t); uint32_t* o = (uint32_t*)dst;
            for (int64_t e = 0; e < N; ++e) for (int i = 0; i < D; ++i) o[e * D + i] = c[e * 4];
        } else if (field[0] == 'r') {
            NEED(N * D, int32_t); int32_t* o = (int32_t*)dst;
            for (int64_t e = 0; e < N; ++e) for (int i = 0; i < D; ++i) o[e * D + i] = ((c[e * 4 + 1] >> i) & 1u) ? pv : 0;
        } else {
            NEED(N, int32_t); int32_t* o = (int32_t*)dst;
            for (int64_t e = 0; e < N; ++e) o[e] = pv * ((int)(c[e * 4 + 1] & 1u) - (int)((c[e * 4 + 1] >> 1) & 1u));
        }
        return GW_OK;
    }
    if (!strcmp(field, "qlen") || !strcmp(field, "queue")) {
        std::vector<GwRec> rec((size_t)N * D);
        HIP_TRY(hipMemcpy(rec.data(), st.qrec, rec.size() * sizeof(GwRec), hipMemcpyDeviceToHost));
        if (field[1] == 'l') {
            NEED(N * D, int32_t);
            int32_t* o = (int32_t*)dst;
            for (int64_t e = 0; e < N; ++e) for (int i = 0; i < D; ++i) o[e * D + i] = (int32_t)(rec[(size_t)i * N + e].w >> 24);
            return GW_OK;
        }
        NEED(N * D * GW_QUEUE_CAP, uint32_t);
        std::vector<uint64_t> runs((size_t)N * D * GW_RING_PHYS);
        HIP_TRY(hipMemcpy(runs.data(), st.runs, runs.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        uint32_t* o = (uint32_t*)dst;
        memset(o, 0, bytes);
        const uint32_t cap = (uint32_t)(env->cfg.mac_header_bytes + env->cfg.net_header_bytes + env->cfg.counter_bound);
        for (int64_t e = 0; e < N; ++e)
            for (int i = 0; i < D; ++i) {
                const GwRunQ q = gw_runq_unpack(rec[(size_t)i * N + e]);
                const uint32_t mult = env->cfg.mult[i] > 0 ? (uint32_t)env->cfg.mult[i] : 1u;
                const uint32_t n = gw_runq_expand(q, runs.data() + ((size_t)e * D + i) * GW_RING_PHYS, mult, cap,
                                                  o + ((size_t)e * D + i) * GW_QUEUE_CAP);
                if (n != q.len) return fail(GW_EHIP, "internal: queue record of env %lld sender %d is inconsistent", (long long)e, i);
            }
        return GW_OK;
    }
    if (!strcmp(field, "peer_received")) {
        if (!st.peer_rx) return fail(GW_EFIELD, "field peer_received needs GW_CFG_PEER_RECEIVE");
        NEED(N * D, uint32_t);
        std::vector<uint32_t> p((size_t)N * D);
        HIP_TRY(hipMemcpy(p.data(), st.peer_rx, p.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint32_t* o = (uint32_t*)dst;
        for (int64_t e = 0; e < N; ++e) for (int i = 0; i < D; ++i) o[e * D + i] = p[(size_t)i * N + e];
        return GW_OK;
    }
    if (!strcmp(field, "pos") || !strcmp(field, "link_power")) {
        if (!st.prx_env) return fail(GW_EFIELD, "field %s needs GW_CFG_PER_ENV_GEOMETRY", field);
        return read_geometry(st, field[0] == 'p', dst, bytes, field);
    }
    if (!strcmp(field, "rx_power") && st.rxp) {          // live-PHY mode: the f64 itself, rows of gw_rp(R) doubles per env
        NEED(N * R, double); double* o = (double*)dst;
        const int RP = gw_rp(R);
        std::vector<double> rx((size_t)N * RP);
        HIP_TRY(hipMemcpy(rx.data(), st.rxp, rx.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t e = 0; e < N; ++e) for (int r = 0; r < R; ++r) o[e * R + r] = rx[(size_t)e * RP + r];
        return GW_OK;
    }
    if (!strcmp(field, "rx_power")) {
        NEED(N * R, double);
        std::vector<uint8_t> s((size_t)N * st.XB);
        HIP_TRY(hipMemcpy(s.data(), st.xs, s.size(), hipMemcpyDeviceToHost));
        double* o = (double*)dst;
        for (int64_t e = 0; e < N; ++e) for (int r = 0; r < R; ++r) o[e * R + r] = env->tab.state_val[r][s[(size_t)e * st.XB + r]];
        return GW_OK;
    }
    static const char* pe[5] = {"n_tx", "n_delivered", "n_appended", "n_popped", "n_dropped"};
    for (int k = 0; k < 5; ++k)
        if (!strcmp(field, pe[k])) {
            if (!st.pe_stats) return fail(GW_EFIELD, "field %s needs GW_CFG_PER_ENV_STATS", field);
            NEED(N, uint64_t);
            HIP_TRY(hipMemcpy(dst, st.pe_stats + (size_t)k * N, bytes, hipMemcpyDeviceToHost));
            return GW_OK;
        }
#undef NEED
    return fail(GW_EFIELD, "unknown field %s", field);
}

} // extern "C"
