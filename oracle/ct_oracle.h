/*
 * oracle/ct_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Layer 2 of the oracle: plain-C, scalar, f64 restatement of one
 * CounterTrafficEnv.step() as a direct walk over the step's event horizon
 * (SURVEY.md Appendix A).  See ct_oracle.c for the reference citations.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; gymwipe_amd/ never does.
 */
#ifndef CT_ORACLE_H
#define CT_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTO_MAX_DEV     32
#define CTO_MAX_RADIOS  (CTO_MAX_DEV + 1)
#define CTO_QUEUE_CAP   100          /* simple_stack.py:361 */

/* sticky per-env flags */
#define CTO_FLAG_CARRY   1u  /* a transmission would outlive the step (horizon not closed) */
#define CTO_FLAG_REFEXC  2u  /* the reference would raise here (KeyError, simple_stack.py:166) */
#define CTO_FLAG_TIE     4u  /* an exact f64 time tie was resolved by the Appendix A.6 rule */

typedef struct cto_config {
    int32_t num_devices;                     /* D senders; radio D is the RRM */
    double  pos[CTO_MAX_RADIOS][2];          /* metres */
    int32_t mult[CTO_MAX_DEV];               /* packets per counter tick */
    int32_t dest[CTO_MAX_DEV];               /* destination sender index */
    double  slot;                            /* 1e-6 s          simple_stack.py:27 */
    double  frequency;                       /* 2.4e9 Hz        physical.py:298 */
    double  bandwidth;                       /* 22e6 Hz         physical.py:298 */
    double  temperature_c;                   /* 20.0            simple_stack.py:57 */
    double  bit_rate;                        /* 133.33333e3     physical.py:196 */
    double  code_rate;                       /* 0.75            physical.py:192 */
    double  max_ber;                         /* 0.25 (VG bound) physical.py:160-185 */
    double  tx_power_dbm;                    /* 0.0             simple_stack.py:364,521 */
    double  counter_interval;                /* 1e-3 s          counter_traffic.py:31 */
    int32_t counter_bound;                   /* 65536           counter_traffic.py:35 */
    int32_t payload_value;                   /* 2 (swapped ctor args, counter_traffic.py:57) */
    int32_t mac_header_bytes;                /* 13              messages.py:154 */
    int32_t net_header_bytes;                /* 12              messages.py:180 */
    int32_t duration_factor;                 /* 1000            envs/core.py:27 */
    int32_t max_duration;                    /* 20              envs/core.py:25 */
    /* sum of the custom attenuation models of a device pair, dB (JoinedAttenuationModel, physical.py:402-457);
       0 = plain FSPL.  Symmetric. */
    double  extra_att_db[CTO_MAX_RADIOS][CTO_MAX_RADIOS];
    double  start_time;                      /* simulated time at creation (test hook; the reference starts at 0) */
} cto_config;

typedef struct cto_vec cto_vec;

/* fill cfg with the reference constants and the SURVEY 8d layout for D senders */
int  cto_config_default(cto_config* cfg, int num_devices);

cto_vec* cto_create(const cto_config* cfg, int64_t num_envs);
void cto_destroy(cto_vec* v);
/* Position.set(x, y) on one radio of EVERY env of the handle, between two steps (devices/core.py:77-86) */
int cto_set_position(cto_vec* v, int radio, double x, double y);   /* 0, or -2: one shared geometry cannot represent the move (see ct_oracle.c) */

/* reset(): counters <- 0, interpreter <- 0, NO time rewind (counter_traffic.py:135-144).
 * mask may be NULL (all envs).  obs_out may be NULL. */
void cto_reset(cto_vec* v, const uint8_t* mask, int32_t* obs_out);

/* one env.step() for every env; returns number of envs whose action was invalid (<0 => error) */
int  cto_step(cto_vec* v, const int32_t* device, const int32_t* duration,
              int32_t* obs, float* reward, uint8_t* done, int nthreads);

/* state readers: copy `field` of every env into dst (layout documented in ct_oracle.c) */
int  cto_get(const cto_vec* v, const char* field, void* dst, size_t bytes);

/* derived tables for tests */
double cto_attenuation(const cto_vec* v, int a, int b);
double cto_rx_power_mw(const cto_vec* v, int from, int to);
double cto_thermal_mw(const cto_vec* v);
double cto_ber(const cto_vec* v, double signal_mw, double noise_mw);
double cto_data_rate(const cto_vec* v);

#ifdef __cplusplus
}
#endif
#endif
