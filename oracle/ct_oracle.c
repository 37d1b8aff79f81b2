/*
 * oracle/ct_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Layer 2 of the oracle: scalar f64 C restatement of CounterTrafficEnv.step()
 * (reference: gymwipe/envs/counter_traffic.py:146-158) as a direct walk over
 * the event horizon of one step, generalised to D senders.
 *
 * PINNING: the reference cannot be run in the build container (simpy / gym
 * absent).  This file is pinned (tests/test_oracle_pinning.py) by
 *   - the known answer of the reference's tests/envs/test_counter_traffic.py:25-34,
 *   - agreement, field by field and bit for bit, with oracle/des_model.py (the
 *     event-driven restatement that itself reproduces the reference's MAC / PHY /
 *     notifier test answers) on seeded rollouts for D = 2, 4, 16.
 * Beyond that: "parity unpinned" against the live reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp (see oracle/Makefile).
 * Transcendentals go through the same glibc libm entry points CPython uses
 * (log10, pow, sqrt), so every f64 below is bit-identical to the Python
 * expression it restates.  All constants are read from the runtime config so
 * the compiler cannot fold libm calls with a different rounding.
 *
 * Step walk (reference file:line on each function):
 *   A.1  t_s  = t_a + (slot - t_a % slot)                    simtools.py:44-53
 *   A.2  announcement: header 13 B, payload len(str(slots)) B; every sender
 *        hears it; the addressed sender d decides header then payload
 *                                                 simple_stack.py:214-286,536-558
 *   A.3  window at d: pop + transmit while (stop - now) > bits/dataRate
 *                                                 simple_stack.py:397-434
 *   A.4  the RRM decodes each data packet -> interpreter   devices.py:163-168
 *   A.5  t_end = t_r + (slots+1)*slot; counters tick every 1 ms (running sum)
 *        and append `mult` packets of 25+c bytes; reset does not rewind time
 *                                                 counter_traffic.py:53-61,135-144
 *   A.6  ties: earlier-inserted first; process initialisation is URGENT.
 */
#include "ct_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

struct cto_vec {
    cto_config cfg;
    int64_t  n;
    int      D, R;
    /* derived, static geometry */
    double   att[CTO_MAX_RADIOS][CTO_MAX_RADIOS];   /* dB            attenuation_models.py:35 */
    double   prx[CTO_MAX_RADIOS][CTO_MAX_RADIOS];   /* mW, from->to  simple_stack.py:111 */
    double   thermal;                               /* mW            simple_stack.py:77 */
    double   data_rate;                             /* bps           physical.py:197 */
    double   coded_factor;                          /* 2 - codeRate  physical.py:259-263 */
    double   sqrt2pi;
    double   euler;
    /* per-env state */
    double*   now;        /* [n]                                   */
    double*   wake;       /* [n*D]  next counter tick               */
    uint32_t* counter;    /* [n*D]                                 */
    uint32_t* q;          /* [n*D*CAP] ring of packet byte sizes    */
    int32_t*  qhead;      /* [n*D]                                 */
    int32_t*  qlen;       /* [n*D]                                 */
    int32_t*  rv;         /* [n*D]  receivedValues                  */
    int32_t*  latest;     /* [n]    _latestDifference               */
    int32_t*  last_abs;   /* [n]    _lastAbsDifference              */
    uint8_t*  done;       /* [n]                                   */
    double*   rx;         /* [n*R]  _receivedPower incl. residue    */
    uint32_t* flags;      /* [n]                                   */
    uint64_t* n_tx;       /* [n] transmissions (announcements + data) */
    uint64_t* n_deliv;    /* [n] data packets decoded by the RRM    */
    uint64_t* n_app;      /* [n] queue appends                      */
    uint64_t* n_pop;      /* [n] queue pops                         */
    uint64_t* n_drop;     /* [n] drop-oldest events                 */
    uint64_t* talked;     /* [n] bit r: radio r has transmitted, i.e. its attenuation models towards every other radio exist
                           *     (AttenuationModelFactory.getInstance creates a pair's model at its first use, physical.py:500-528) */
};

/* ---- static tables ------------------------------------------------------ */

/* devices/core.py:88-95 + attenuation_models.py:28-36 (CPython: x**2 -> pow(x,2)) */
static double fspl_db(const cto_config* c, int a, int b)
{
    double ax = c->pos[a][0], ay = c->pos[a][1];
    double bx = c->pos[b][0], by = c->pos[b][1];
    if (ax == bx && ay == by) return 0.0;          /* co-located: attenuation stays 0 */
    double dist = sqrt(pow(ax - bx, 2.0) + pow(ay - by, 2.0));
    return 20 * log10(dist) + 20 * log10(c->frequency) - 147.55;
}

/* physical.py:25-58,82-98,208-212 */
static double ber_bpsk(const cto_vec* v, double sig_mw, double noise_mw)
{
    double s = 10 * log10(sig_mw);
    double n = 10 * log10(noise_mw);
    if (s <= n) return 0.5;
    double ratio_db = s - n - 10 * log10(v->cfg.bit_rate);
    double ratio = pow(10.0, ratio_db / 10);
    double x = sqrt(2 * ratio);
    return (1 - pow(v->euler, -1.4 * x)) * pow(v->euler, -(pow(x, 2.0) / 2))
           / (1.135 * v->sqrt2pi * x);
}

static int ndigits(int32_t slots)               /* messages.py:51-52: len(str(value)) */
{
    int n = 1;
    while (slots >= 10) { slots /= 10; ++n; }
    return n;
}

int cto_config_default(cto_config* c, int D)
{
    if (!c || D < 2 || D > CTO_MAX_DEV) return -1;
    memset(c, 0, sizeof *c);
    c->num_devices = D;
    if (D == 2) {                                /* counter_traffic.py:124-127 */
        c->pos[0][0] = 0.0; c->pos[0][1] = 2.0;
        c->pos[1][0] = 0.0; c->pos[1][1] = -2.0;
    } else {                                     /* SURVEY.md 8d */
        for (int i = 0; i < D; ++i) {
            double ang = M_PI / 2 - 2 * M_PI * i / D;
            c->pos[i][0] = 2.0 * cos(ang);
            c->pos[i][1] = 2.0 * sin(ang);
        }
    }
    c->pos[D][0] = 0.0; c->pos[D][1] = 0.0;      /* RRM, counter_traffic.py:133 */
    for (int i = 0; i < D; ++i) {
        c->mult[i] = (i % 2 == 0) ? 1 : 3;
        c->dest[i] = (i + 1) % D;
    }
    c->slot = 1e-6;
    c->frequency = 2.4e9;
    c->bandwidth = 22e6;
    c->temperature_c = 20.0;
    c->bit_rate = 133.33333e3;
    c->code_rate = 0.75;
    c->max_ber = 0.25;
    c->tx_power_dbm = 0.0;
    c->counter_interval = 0.001;
    c->counter_bound = 65536;
    c->payload_value = 2;
    c->mac_header_bytes = 13;
    c->net_header_bytes = 12;
    c->duration_factor = 1000;
    c->max_duration = 20;
    return 0;
}

/* attenuation and received power of every link from the current positions (attenuation_models.py:28-36,
 * physical.py:402-457 joined models, simple_stack.py:111) */
static void build_links(cto_vec* v)
{
    const cto_config* cfg = &v->cfg;
    const int R = v->R;
    for (int a = 0; a < R; ++a)
        for (int b = 0; b < R; ++b) {
            if (a == b) { v->att[a][b] = 0; v->prx[a][b] = 0; continue; }
            v->att[a][b] = fspl_db(cfg, a, b);
            if (cfg->extra_att_db[a][b] != 0.0) {            /* physical.py:457: sum() over [FSPL, custom models] */
                volatile double joined = 0.0 + v->att[a][b];
                joined = joined + cfg->extra_att_db[a][b];
                v->att[a][b] = joined;
            }
            v->prx[a][b] = pow(10.0, (cfg->tx_power_dbm - v->att[a][b]) / 10);
        }
}

/* one pair's attenuation and link power from the current positions (both directions) */
static void build_pair(cto_vec* v, int a, int b)
{
    const cto_config* cfg = &v->cfg;
    double att = fspl_db(cfg, a, b);
    if (cfg->extra_att_db[a][b] != 0.0) {                    /* physical.py:457: sum() over [FSPL, custom models] */
        volatile double joined = 0.0 + att;
        joined = joined + cfg->extra_att_db[a][b];
        att = joined;
    }
    v->att[a][b] = v->att[b][a] = att;
    v->prx[a][b] = v->prx[b][a] = pow(10.0, (cfg->tx_power_dbm - att) / 10);
}

/* Position.set between two steps (devices/core.py:77-86): the attenuation models of the moved radio's pairs hear about it
 * (physical.py:380-386) -- and, as in the reference,
 *   * a pair whose new distance is >= STANDBY_THRESHOLD (3000 m, physical.py:371) does NOT update: the stale attenuation stays;
 *   * a pair that now shares one position does not update either (FsplAttenuation._update returns without setting anything,
 *     attenuation_models.py:31-33: the previous value stays, it does not become 0 dB);
 *   * a pair whose model does not exist yet (neither radio has transmitted: models are created at first use) has nothing
 *     to keep: it is created later from the positions of that moment.
 * Nothing is on the air at a step boundary, so no reception is touched (simple_stack.py:119-128).
 * Returns 0, or -2 when a keep-rule applies to a pair whose model exists in some envs of this handle and not in others (one
 * shared geometry cannot represent that: use one handle per env). */
int cto_set_position(cto_vec* v, int radio, double x, double y)
{
    if (!v || radio < 0 || radio >= v->R) return -1;
    v->cfg.pos[radio][0] = x;
    v->cfg.pos[radio][1] = y;
    for (int b = 0; b < v->R; ++b) {
        if (b == radio) continue;
        const double bx = v->cfg.pos[b][0], by = v->cfg.pos[b][1];
        const int same = (x == bx && y == by);
        const double dist = sqrt(pow(x - bx, 2.0) + pow(y - by, 2.0));    /* devices/core.py:88-95 */
        const uint64_t pair = ((uint64_t)1 << radio) | ((uint64_t)1 << b);
        int exists = (v->talked[0] & pair) != 0;
        if (same || dist >= 3000.0) {
            for (int64_t e = 1; e < v->n; ++e)
                if (((v->talked[e] & pair) != 0) != exists) return -2;
            if (exists) continue;                                         /* the stale value stays */
        }
        build_pair(v, radio, b);
    }
    return 0;
}

cto_vec* cto_create(const cto_config* cfg, int64_t n)
{
    if (!cfg || n <= 0 || cfg->num_devices < 2 || cfg->num_devices > CTO_MAX_DEV) return NULL;
    cto_vec* v = (cto_vec*)calloc(1, sizeof *v);
    if (!v) return NULL;
    v->cfg = *cfg;
    v->n = n;
    int D = v->D = cfg->num_devices;
    int R = v->R = D + 1;
    v->sqrt2pi = sqrt(2 * M_PI);
    v->euler = M_E;
    v->thermal = 1.38e-23 * (cfg->temperature_c + 273.15) * cfg->bandwidth * 1000;
    v->data_rate = cfg->code_rate * cfg->bit_rate;
    v->coded_factor = 2 - cfg->code_rate;
    build_links(v);
#define ALLOC(p, cnt) do { (p) = calloc((size_t)(cnt), sizeof *(p)); if (!(p)) { cto_destroy(v); return NULL; } } while (0)
    ALLOC(v->now, n);       ALLOC(v->wake, n * D);    ALLOC(v->counter, n * D);
    ALLOC(v->q, n * D * CTO_QUEUE_CAP);
    ALLOC(v->qhead, n * D); ALLOC(v->qlen, n * D);    ALLOC(v->rv, n * D);
    ALLOC(v->latest, n);    ALLOC(v->last_abs, n);    ALLOC(v->done, n);
    ALLOC(v->rx, n * R);    ALLOC(v->flags, n);
    ALLOC(v->n_tx, n);      ALLOC(v->n_deliv, n);     ALLOC(v->n_app, n);
    ALLOC(v->n_pop, n);     ALLOC(v->n_drop, n);      ALLOC(v->talked, n);
#undef ALLOC
    for (int64_t e = 0; e < n; ++e) {
        v->now[e] = cfg->start_time;
        for (int i = 0; i < D; ++i) {
            v->wake[e * D + i] = cfg->start_time; /* first tick at t = 0 (at creation) */
            v->counter[e * D + i] = 1;           /* counter_traffic.py:48 */
        }
        for (int r = 0; r < R; ++r) v->rx[e * R + r] = v->thermal;
    }
    return v;
}

void cto_destroy(cto_vec* v)
{
    if (!v) return;
    free(v->now); free(v->wake); free(v->counter); free(v->q); free(v->qhead);
    free(v->qlen); free(v->rv); free(v->latest); free(v->last_abs); free(v->done);
    free(v->rx); free(v->flags); free(v->n_tx); free(v->n_deliv); free(v->n_app);
    free(v->n_pop); free(v->n_drop); free(v->talked);
    free(v);
}

/* counter_traffic.py:135-144, :69-73 */
void cto_reset(cto_vec* v, const uint8_t* mask, int32_t* obs_out)
{
    int D = v->D;
    for (int64_t e = 0; e < v->n; ++e) {
        if (!mask || mask[e]) {
            for (int i = 0; i < D; ++i) { v->counter[e * D + i] = 0; v->rv[e * D + i] = 0; }
            v->latest[e] = 0; v->last_abs[e] = 0; v->done[e] = 0;
        }
        if (obs_out) obs_out[e] = v->latest[e] + v->cfg.counter_bound;
    }
}

/* ---- one env ------------------------------------------------------------- */

typedef struct {
    cto_vec* v; int64_t e; int D, R;
} ectx;

/* one counter tick of sender i: counter_traffic.py:53-61, devices.py:84-86,
 * simple_stack.py:463-471 (deque(maxlen=100).append drops the oldest) */
static void tick(ectx* c, int i)
{
    cto_vec* v = c->v; int64_t k = c->e * c->D + i;
    uint32_t* ring = v->q + k * CTO_QUEUE_CAP;
    uint32_t size = (uint32_t)(v->cfg.mac_header_bytes + v->cfg.net_header_bytes) + v->counter[k];
    for (int m = 0; m < v->cfg.mult[i]; ++m) {
        if (v->qlen[k] == CTO_QUEUE_CAP) {
            v->qhead[k] = (v->qhead[k] + 1) % CTO_QUEUE_CAP;
            v->qlen[k]--;
            v->n_drop[c->e]++;
        }
        ring[(v->qhead[k] + v->qlen[k]) % CTO_QUEUE_CAP] = size;
        v->qlen[k]++;
        v->n_app[c->e]++;
    }
    if (v->counter[k] < (uint32_t)v->cfg.counter_bound) v->counter[k]++;
    v->wake[k] = v->wake[k] + v->cfg.counter_interval;      /* running sum, not k*dt */
}

/* all ticks of sender i with wake < t (strict) or <= t */
static void ticks_until(ectx* c, int i, double t, int inclusive)
{
    cto_vec* v = c->v; int64_t k = c->e * c->D + i;
    for (;;) {
        double w = v->wake[k];
        if (w < t || (inclusive && w == t)) {
            if (w == t) v->flags[c->e] |= CTO_FLAG_TIE;
            tick(c, i);
        } else break;
    }
}

typedef struct { double t_s, t_h, t_e, stop; } txtimes;

/* simple_stack.py:204 + physical.py:244-279 + simtools.py:112-116.
 * `cur` is the time the PHY gets the SEND command. */
static txtimes tx_times(const cto_vec* v, double cur, int hdr_bytes, int pay_bytes)
{
    txtimes x;
    double slot = v->cfg.slot;
    x.t_s = cur + (slot - fmod(cur, slot));                  /* full slot if aligned */
    double hd = (hdr_bytes * 8) / v->data_rate;
    double pd = (pay_bytes * 8) / v->data_rate;
    double dur = hd + pd;
    x.stop = x.t_s + dur;
    double th = x.t_s + hd;
    x.t_h = (th > x.t_s) ? x.t_s + (th - x.t_s) : x.t_s + 0;
    x.t_e = (x.stop > x.t_s) ? x.t_s + (x.stop - x.t_s) : x.t_s + 0;
    return x;
}

/* Reception of one transmission at radio j (simple_stack.py:214-267), assuming
 * nothing else is on the air.  rx[j] already includes the signal power p.
 * Returns 1 iff header and payload decode. */
static int receive(ectx* c, int j, double p, const txtimes* x, int hdr_bytes, int pay_bytes)
{
    cto_vec* v = c->v;
    double rxj = v->rx[c->e * c->R + j];
    double br = v->cfg.bit_rate;
    if (!(rxj - p >= 0)) v->flags[c->e] |= CTO_FLAG_REFEXC;  /* :168 `assert noisePower >= 0` -> AssertionError */
    double ber = ber_bpsk(v, p, rxj - p);                    /* :161-173 */
    double err = 0 + ber * (x->t_h - x->t_s) * br;           /* :180-188 at header end */
    double hdr_bits = (hdr_bytes * 8) * v->coded_factor;
    if (!((nearbyint(err) / hdr_bits) <= v->cfg.max_ber)) return 0;   /* :269-286 */
    ber = ber_bpsk(v, p, rxj - p);                           /* :246-248 */
    double seg = ber * (x->t_e - x->t_h) * br;
    err = 0 + seg;                                           /* from the -p power callback (:223-231) */
    if (!(x->t_e >= x->stop)) v->flags[c->e] |= CTO_FLAG_REFEXC;      /* `not t.completed` -> KeyError */
    err = err + seg;                                         /* and again after the resume (:252) */
    double pay_bits = (pay_bytes * 8) * v->coded_factor;
    return (nearbyint(err) / pay_bits) <= v->cfg.max_ber;
}

static void step_one(cto_vec* v, int64_t e, int d, int duration,
                     int32_t* obs, float* reward, uint8_t* done)
{
    ectx c = { v, e, v->D, v->R };
    const int D = v->D, R = v->R, RRM = v->D;
    const cto_config* cf = &v->cfg;
    double* rx = v->rx + e * R;
    const int mh = cf->mac_header_bytes;

    int32_t slots = duration * cf->duration_factor;          /* counter_traffic.py:149 */
    double t_a = v->now[e];

    /* ---- A.1/A.2 announcement ------------------------------------------ */
    int L = ndigits(slots);
    txtimes an = tx_times(v, t_a, mh, L);
    v->n_tx[e]++;
    v->talked[e] |= (uint64_t)1 << RRM;                               /* every listener's model towards the RRM exists from here on */
    for (int j = 0; j < D; ++j) rx[j] = rx[j] + v->prx[RRM][j];       /* :130-139 */
    int granted = receive(&c, d, v->prx[RRM][d], &an, mh, L);
    for (int j = 0; j < D; ++j) rx[j] = rx[j] + (-v->prx[RRM][j]);    /* :146-154 */
    double t_r = an.t_e;
    double t_end = t_r + (slots + 1) * cf->slot;             /* simple_stack.py:557-558 */

    /* ---- A.3 window at sender d ----------------------------------------- */
    if (granted) {
        int64_t kd = e * D + d;
        uint32_t* ring = v->q + kd * CTO_QUEUE_CAP;
        double total = slots * cf->slot;                     /* :400 */
        double stopw = t_r + total;                          /* :401 == timeout time :406 */
        double cur = t_r;
        ticks_until(&c, d, cur, 0);                          /* ties: MAC init is URGENT, goes first */
        for (;;) {
            if (v->qlen[kd] == 0) {                          /* :409-416 */
                double w = v->wake[kd];
                /* a silent sender (mult 0) never signals packet-added: the MAC waits for the window timeout */
                if (cf->mult[d] > 0 && w < stopw) { cur = w; tick(&c, d); }
                else break;
            }
            uint32_t s = ring[v->qhead[kd]];
            double need = (double)(s * 8u) / v->data_rate;   /* messages.py:67-75 */
            if (!((stopw - cur) > need)) break;              /* :418-420 idle to window end */
            v->qhead[kd] = (v->qhead[kd] + 1) % CTO_QUEUE_CAP;        /* :425 */
            v->qlen[kd]--;
            v->n_pop[e]++;
            txtimes x = tx_times(v, cur, mh, (int)s - mh);
            v->n_tx[e]++;
            v->talked[e] |= (uint64_t)1 << d;
            for (int j = 0; j < R; ++j) if (j != d) rx[j] = rx[j] + v->prx[d][j];
            int ok = receive(&c, RRM, v->prx[d][RRM], &x, mh, (int)s - mh);
            for (int j = 0; j < R; ++j) if (j != d) rx[j] = rx[j] + (-v->prx[d][j]);
            if (ok) {                                        /* devices.py:163-168, counter_traffic.py:75-80 */
                v->n_deliv[e]++;
                v->rv[e * D + d] = cf->payload_value;
                v->latest[e] = v->rv[e * D + 0] - v->rv[e * D + 1];
                if (cf->payload_value == cf->counter_bound) v->done[e] = 1;
            }
            if (!(x.t_e < t_end)) v->flags[e] |= CTO_FLAG_CARRY;
            ticks_until(&c, d, x.t_e, 1);                    /* ticks are older than the MAC's resume event */
            cur = x.t_e;
            if (!(cur < stopw)) break;                       /* window timeout already processed */
        }
    }

    /* ---- A.5 step end ---------------------------------------------------- */
    for (int i = 0; i < D; ++i) ticks_until(&c, i, t_end, 1);
    v->now[e] = t_end;

    int32_t abs_d = v->latest[e] < 0 ? -v->latest[e] : v->latest[e];  /* counter_traffic.py:85-101 */
    int32_t r = v->last_abs[e] - abs_d;
    v->last_abs[e] = abs_d;
    if (r > 10) r = 10; else if (r < -10) r = -10;
    if (obs)    obs[e] = v->latest[e] + cf->counter_bound;
    if (reward) reward[e] = (float)r;
    if (done)   done[e] = v->done[e];
}

int cto_step(cto_vec* v, const int32_t* device, const int32_t* duration,
             int32_t* obs, float* reward, uint8_t* done, int nthreads)
{
    if (!v || !device || !duration) return -1;
    int bad = 0;
    for (int64_t e = 0; e < v->n; ++e)                       /* counter_traffic.py:147 */
        if (device[e] < 0 || device[e] >= v->D || duration[e] < 0 || duration[e] >= v->cfg.max_duration) ++bad;
    if (bad) return bad;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
    for (int64_t e = 0; e < v->n; ++e)
        step_one(v, e, device[e], duration[e], obs, reward, done);
    return 0;
}

/* ---- readers -------------------------------------------------------------- */
int cto_get(const cto_vec* v, const char* f, void* dst, size_t bytes)
{
    if (!v || !f || !dst) return -1;
    int64_t n = v->n; int D = v->D, R = v->R;
#define COPY(name, ptr, cnt) if (!strcmp(f, name)) { size_t need = (size_t)(cnt) * sizeof *(ptr); if (bytes != need) return -2; memcpy(dst, (ptr), need); return 0; }
    COPY("now", v->now, n)            COPY("wake", v->wake, n * D)
    COPY("counter", v->counter, n * D) COPY("qlen", v->qlen, n * D)
    COPY("received", v->rv, n * D)    COPY("latest_diff", v->latest, n)
    COPY("last_abs", v->last_abs, n)  COPY("rx_power", v->rx, n * R)
    COPY("flags", v->flags, n)        COPY("n_tx", v->n_tx, n)
    COPY("n_delivered", v->n_deliv, n) COPY("n_appended", v->n_app, n)
    COPY("n_popped", v->n_pop, n)     COPY("n_dropped", v->n_drop, n)
#undef COPY
    if (!strcmp(f, "queue")) {           /* logical order from the head, zero padded */
        size_t need = (size_t)n * D * CTO_QUEUE_CAP * sizeof(uint32_t);
        if (bytes != need) return -2;
        uint32_t* out = (uint32_t*)dst;
        for (int64_t k = 0; k < n * D; ++k)
            for (int s = 0; s < CTO_QUEUE_CAP; ++s)
                out[k * CTO_QUEUE_CAP + s] = s < v->qlen[k]
                    ? v->q[k * CTO_QUEUE_CAP + (v->qhead[k] + s) % CTO_QUEUE_CAP] : 0u;
        return 0;
    }
    return -3;
}

double cto_attenuation(const cto_vec* v, int a, int b) { return v->att[a][b]; }
double cto_rx_power_mw(const cto_vec* v, int from, int to) { return v->prx[from][to]; }
double cto_thermal_mw(const cto_vec* v) { return v->thermal; }
double cto_ber(const cto_vec* v, double s, double n) { return ber_bpsk(v, s, n); }
double cto_data_rate(const cto_vec* v) { return v->data_rate; }
