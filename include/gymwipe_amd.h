/*
 * gymwipe_amd.h -- C-ABI of the MI355X-native vectorised Gym-WiPE env.step().
 *
 * Drop-in boundary for ONE path of the reference (Gryph66/gymwipe): the
 * discrete-event advance behind the frequency-band-assignment environment
 * `CounterTrafficEnv` (gymwipe/envs/counter_traffic.py).  N independent
 * environments live structure-of-arrays in HBM and are advanced by hand-written
 * HIP kernels for gfx950.  There is NO CPU fallback in this library: every
 * compute entry point needs a HIP device and fails loudly without one.
 *
 * Conventions
 *   - every function returns 0 on success or a negative GW_E* code;
 *     gw_last_error() returns a thread-local message for the last failure
 *   - `*_dev` pointers are DEVICE pointers owned by the caller (e.g. torch
 *     tensors), contiguous, length num_envs unless stated; `stream` is a
 *     hipStream_t passed as void* (NULL = default stream); gw_step/gw_reset are
 *     asynchronous on that stream and allocate nothing
 *   - one gw_env per GPU; a handle is not thread-safe; different handles may be
 *     driven from different host threads
 *
 * Reference interface each entry point replaces (file:line in the reference):
 *   gw_config_default  class constants: envs/core.py:25,27; counter_traffic.py:31-35,124-127;
 *                      simple_stack.py:27,57,361,364; physical.py:192-197,298; messages.py:154,180
 *   gw_create          CounterTrafficEnv.__init__            counter_traffic.py:114-133
 *   gw_reset           CounterTrafficEnv.reset               counter_traffic.py:135-144
 *   gw_step            CounterTrafficEnv.step                counter_traffic.py:146-158
 *                        -> SimpleRrmDevice.assignFrequencyBand   networking/devices.py:178-203
 *                        -> SimMan.runSimulation(eProcessed)      simtools.py:77-88
 *                        -> Interpreter.getFeedback               envs/core.py:142-153
 *   gw_set_position(s) Position.set -> PositionalAttenuationModel -> FsplAttenuation   devices/core.py:52-86, physical.py:380-386,
 *                      attenuation_models.py:28-36
 *   gw_received        CounterTrafficInterpreter.receivedValues / getInfo   counter_traffic.py:72,109-112
 *   gw_get_state       attribute reads the reference's tests perform: SimMan.now, sender.counter,
 *                      sender._mac._packetQueue, phy._receivedPower
 *   gw_delivered       what SimpleRrmDevice.onPacketReceived feeds a custom Interpreter     networking/devices.py:163-168
 *   gw_enqueue         SimpleNetworkDevice.send -> SimpleMac queue           networking/devices.py:84-86, simple_stack.py:463-471
 *   gw_rollout         a Python loop over step()
 *   gw_step_fb         gw_step + the step's feedback as one byte per env (the row a multi-GPU job gathers)
 *   gw_pack_feedback / gw_unpack_feedback   (multi-GPU exchange format; the reference is single-process)
 *   gw_pendulum_step   InvertedPendulumEnv.step              envs/inverted_pendulum.py:101-113
 *   gw_plant_*         OdePlant.updateState, SlidingPendulum getters / setMotorVelocity   plants/core.py:38-49,
 *                      plants/sliding_pendulum.py:57-85   (builder-defined linear plant)
 *   gw_grid_*          SimMan.runSimulation(seconds) over the benchmark fixture, Position.set
 *                      tests/test_benchmark.py:20-91, devices/core.py:77-86
 *   gw_destroy         (garbage collection)
 */
#ifndef GYMWIPE_AMD_H
#define GYMWIPE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GW_ABI_VERSION   1

#define GW_MAX_DEVICES   32                 /* assignable senders per env */
#define GW_MAX_RADIOS    (GW_MAX_DEVICES + 1)
#define GW_QUEUE_CAP     100                /* simple_stack.py:361 deque(maxlen=100) */

/* error codes */
#define GW_OK             0
#define GW_EINVAL        -1                 /* bad argument / config */
#define GW_ENODEVICE     -2                 /* no usable HIP device */
#define GW_EHIP          -3                 /* HIP runtime error (message has details) */
#define GW_ENOMEM        -4
#define GW_EUNSUPPORTED  -5                 /* config outside what the kernels model */
#define GW_EFIELD        -6                 /* unknown gw_get_state field / size mismatch */

/* per-env sticky flag bits (gw_get_state "flags"; OR over envs in gw_stats) */
#define GW_FLAG_CARRY    1u   /* a transmission would outlive its step: horizon not closed */
#define GW_FLAG_REFEXC   2u   /* the reference would raise here (simple_stack.py:166 KeyError) */
#define GW_FLAG_TIE      4u   /* an exact f64 time tie was resolved by the insertion-order rule */
#define GW_FLAG_BADACT   8u   /* action outside the action space: env left untouched this step */
#define GW_FLAG_INTERNAL 16u  /* a kernel self-check of an exact fast path failed (never expected; the env's state is invalid) */

typedef struct gw_config {
    int32_t abi_version;                    /* GW_ABI_VERSION */
    int32_t hip_device;                     /* device ordinal */
    int64_t num_envs;                       /* N */
    int32_t num_devices;                    /* D senders; radio index D is the RRM */
    int32_t flags;                          /* GW_CFG_* */
    double  pos[GW_MAX_RADIOS][2];          /* metres; [D] = RRM */
    int32_t mult[GW_MAX_DEVICES];           /* packets per counter tick (packetMultiplicity); 0 = a sender with nothing to send */
    int32_t dest[GW_MAX_DEVICES];           /* destination sender index */
    double  slot;                           /* TIME_SLOT_LENGTH 1e-6 s */
    double  frequency;                      /* 2.4e9 Hz */
    double  bandwidth;                      /* 22e6 Hz */
    double  temperature_c;                  /* 20.0 */
    double  bit_rate;                       /* 133.33333e3 bps */
    double  code_rate;                      /* 3/4 */
    double  max_ber;                        /* 0.25 (Varshamov-Gilbert bound for 3/4) */
    double  tx_power_dbm;                   /* 0.0 */
    double  counter_interval;               /* COUNTER_INTERVAL 1e-3 s */
    int32_t counter_bound;                  /* COUNTER_BOUND 65536 */
    int32_t payload_value;                  /* 2: value field of every data payload (swapped ctor args) */
    int32_t mac_header_bytes;               /* 13 */
    int32_t net_header_bytes;               /* 12 */
    int32_t duration_factor;                /* ASSIGNMENT_DURATION_FACTOR 1000 */
    int32_t max_duration;                   /* MAX_ASSIGN_DURATION 20 */
    /* Custom attenuation models (AttenuationModelFactory.setCustomModels + JoinedAttenuationModel, physical.py:402-457,
     * 477-498): geometry is static here, so the models a caller sets for a device pair reduce to one number, the sum of
     * their attenuations in dB, which the joined model adds to the free-space term (sum() over [FSPL, custom...]).
     * Must be symmetric; 0 = the pair keeps the plain FSPL model. */
    double  extra_att_db[GW_MAX_RADIOS][GW_MAX_RADIOS];
    /* Simulated time at creation (the reference always starts at 0: SimMan.init).  A test hook: the clock and the first
     * counter tick start here, which puts the f64 arithmetic of the step (slot remainders, tick sums, decode sums) into
     * the coarser binades a run reaches only after ~10^8 steps, and past the limits of the validated fast paths. */
    double  start_time;
} gw_config;

#define GW_CFG_PER_ENV_STATS  1             /* explicit-queue mode only: keep per-env event counters (the default
                                               mode always keeps them, as 32-bit counters that wrap) */
#define GW_CFG_EXPLICIT_QUEUE 2             /* MAC queues as deques of packet-size runs that hold ANY traffic (generic kernel,
                                               gw_runq.h); default: exact suffix encoding of counter traffic, gw_queue.h */

/* The three flags below widen the path towards the callers SURVEY 8f ranks next (receive-mode MACs and
 * traffic other than counters); they need GW_CFG_EXPLICIT_QUEUE.  Together they replay the reference's
 * test_simple_mac (tests/networking/test_stack.py:134-235). */
#define GW_CFG_NO_COUNTER_TRAFFIC 4         /* no counter processes: packets come from gw_enqueue only */
#define GW_CFG_PEER_RECEIVE   8             /* `device.receiving = True` on every sender (networking/devices.py:71-111: a
                                               RECEIVE command re-issued on completion, as test_stack.py:176-186 does by
                                               hand; SimpleMac side simple_stack.py:435-444,453-461,473-484): data packets
                                               decoded at dest[d] while it is idle are handed up and counted
                                               ("peer_received") */
#define GW_CFG_FLOAT_DURATION 16            /* the assignment duration is passed as a float (test_stack.py:197):
                                               the announcement payload is len(str(float(slots))) bytes */

#define GW_CFG_PER_ENV_GEOMETRY 32          /* positions per ENVIRONMENT (either queue mode): every env starts with cfg.pos and
                                               gw_set_position(s) moves radios between steps -- Position.set,
                                               devices/core.py:52-86; link powers per env, rebuilt on the device (ct_step_dyn.hip) */

typedef struct gw_stats {                   /* totals since gw_create, over all envs */
    uint64_t steps;                         /* env-steps executed */
    uint64_t transmissions;                 /* announcements + data packets */
    uint64_t delivered;                     /* data packets decoded by the RRM */
    uint64_t appended;                      /* queue appends  (k_app of SURVEY 8d) */
    uint64_t popped;                        /* queue pops     (k_pop) */
    uint64_t dropped;                       /* drop-oldest events */
    uint64_t flags_or;                      /* OR of all per-env flag words */
    uint64_t bad_actions;                   /* env-steps skipped for an invalid action */
} gw_stats;

typedef struct gw_env gw_env;

int         gw_abi_version(void);
const char* gw_last_error(void);
int         gw_device_count(int* count);

int gw_config_default(gw_config* cfg, int64_t num_envs, int32_t num_devices);

int gw_create(const gw_config* cfg, gw_env** out);
int gw_destroy(gw_env* env);

/* reset(): counters <- 0, interpreter <- 0, simulated time NOT rewound.
 * mask_dev: NULL = all envs, else uint8[N] (non-zero = reset).  obs_dev may be NULL. */
int gw_reset(gw_env* env, const uint8_t* mask_dev, int32_t* obs_dev, void* stream);

/* one env.step() for all N envs.  device_dev in [0,D), duration_dev in [0,max_duration). */
int gw_step(gw_env* env, const int32_t* device_dev, const int32_t* duration_dev,
            int32_t* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream);

/* gw_step that ALSO writes the step's feedback in the one-byte exchange format of gw_pack_feedback (below) to
 * feedback_byte_dev[N] -- the row a multi-GPU job all-gathers at the end of the step (Interpreter.getFeedback, envs/core.py:142-153,
 * in the form the gather moves).  Fused into the step kernel in the default mode (one more byte stored per env, no extra launch);
 * the other modes run gw_step and then the packing kernel.  feedback_byte_dev == NULL: exactly gw_step. */
int gw_step_fb(gw_env* env, const int32_t* device_dev, const int32_t* duration_dev,
               int32_t* obs_dev, float* reward_dev, uint8_t* done_dev, uint8_t* feedback_byte_dev, void* stream);

/* SimpleNetworkDevice.send(data, dest[sender]) (networking/devices.py:84-86) -> SimpleMac queue append with
 * drop-oldest (simple_stack.py:463-471), for every env with payload_bytes_dev[e] >= 0 (int32[N]: byte size of
 * the network payload; headers are added).  GW_CFG_EXPLICIT_QUEUE only. */
int gw_enqueue(gw_env* env, int32_t sender, const int32_t* payload_bytes_dev, void* stream);

/* One byte per env-step for the end-of-step observation gather of a multi-GPU job: bits 0-1 sign(obs - COUNTER_BOUND) + 1,
 * bits 2-6 reward + 10, bit 7 done -- lossless for the built-in interpreter (counter_traffic.py:85-112, envs/core.py:142-153),
 * 9x less xGMI traffic than the (int32, float32, uint8) triple.  `count` elements in any layout (e.g. [steps][N]).
 * gw_pack_feedback fails with GW_EINVAL (synchronously, it reads a flag back) if a value is not representable, i.e. the
 * buffers do not hold the built-in interpreter's feedback. */
int gw_pack_feedback(gw_env* env, int64_t count, const int32_t* obs_dev, const float* reward_dev, const uint8_t* done_dev,
                     uint8_t* packed_dev, int32_t check, void* stream);
int gw_unpack_feedback(gw_env* env, int64_t count, const uint8_t* packed_dev, int32_t* obs_dev, float* reward_dev,
                       uint8_t* done_dev, void* stream);

/* K consecutive env.step() calls from pre-staged actions, inputs/outputs laid out [K][N].  In the default
 * mode this is ONE persistent launch per 64 steps (state in registers, lanes free-running through their
 * own event sequences: ct_rollout_sfx.hip); results are identical to K gw_step calls. */
int gw_rollout(gw_env* env, int32_t steps, const int32_t* device_dev, const int32_t* duration_dev,
               int32_t* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream);

/* Cumulative number of data packets the RRM has decoded per env since gw_create: uint32[N], device pointer
 * (default mode).  A custom Interpreter (envs/core.py:59-159) differences this across a step to learn how many
 * packets of the assigned sender the RRM sniffed (networking/devices.py:163-168). */
int gw_delivered(gw_env* env, uint32_t* out_dev, void* stream);

/* Position.set(x, y) on radio `radio` (0..D-1 senders, D = the RRM) of every env with mask_dev == NULL or mask_dev[e] != 0,
 * at the envs' current time, i.e. between two env.step() calls (devices/core.py:77-86); the attenuation of every link of
 * that radio is recomputed (physical.py:380-386, attenuation_models.py:28-36) with the device libm.  x_dev / y_dev:
 * double[N] device pointers.  Needs GW_CFG_PER_ENV_GEOMETRY. */
int gw_set_position(gw_env* env, int32_t radio, const double* x_dev, const double* y_dev, const uint8_t* mask_dev, void* stream);
/* the same for every radio at once: pos_dev is double[N][R][2] (device pointer), R = num_devices + 1 */
int gw_set_positions(gw_env* env, const double* pos_dev, const uint8_t* mask_dev, void* stream);

/* receivedValues of every env: int32[N][D] (row-major), device pointer. */
int gw_received(gw_env* env, int32_t* out_dev, void* stream);

/* Synchronous state readers for tests/debugging: copies `field` of every env to HOST memory.
 *   "now" f64[N] | "wake" f64[N][D] | "counter" u32[N][D] | "qlen" i32[N][D]
 *   "queue" u32[N][D][GW_QUEUE_CAP] (logical order from the head, zero padded)
 *   "received" i32[N][D] | "latest_diff" i32[N] | "last_abs" i32[N] | "rx_power" f64[N][R]
 *   "flags" u32[N] | "n_tx","n_delivered","n_appended","n_popped","n_dropped" u64[N] (per-env event counts)
 *   GW_CFG_PER_ENV_GEOMETRY: "pos" f64[N][R][2] | "link_power" f64[N][R][R] (mW, from -> to) */
int gw_get_state(gw_env* env, const char* field, void* dst_host, size_t bytes);

/* Ranges of the event counts in the DEFAULT queue mode, where most of them are derived rather than counted (only popped,
 * delivered, bad actions and flags are bumped on the device): steps = env.step() launches - bad actions; transmissions =
 * steps + popped; appended = ticks * sum(multiplicity); dropped = appended - popped - queued.  `popped` and `delivered` of
 * one env share a 64-bit word (popped in the low half): beyond 2^32 - 1 pops per env the carry would reach `delivered`; the
 * tick counter is 32 bits (2^32 ticks = 49 days of simulated time at the reference's 1 ms interval).  Neither limit is
 * checked; a handle that may get there should use GW_CFG_EXPLICIT_QUEUE | GW_CFG_PER_ENV_STATS (64-bit counts per event). */
int gw_stats_read(gw_env* env, gw_stats* out);          /* synchronises the device */
/* the sticky per-env flag words back to zero (gw_get_state "flags", gw_stats.flags_or), so that a later check tells
 * WHEN a condition arose; counters are untouched */
int gw_clear_flags(gw_env* env, void* stream);
int gw_state_bytes(gw_env* env, uint64_t* bytes);       /* HBM held by this handle */

/* Checkpoint / restore.  The reference cannot snapshot a running simulation (SimPy generators; only the agent's weights are
 * saved, agents/dqn_counter_traffic.py:73); here an env's state is a handful of arrays in HBM.  gw_get_snapshot copies all of it
 * to host memory (gw_snapshot_bytes bytes; synchronises the device); gw_set_state restores it into a handle created with the
 * same gw_config -- the same handle later on, or a fresh one on any GPU -- after which every step continues bit for bit as the
 * snapshotted handle would have.  GW_EINVAL for a blob that is not a snapshot of this ABI / configuration. */
int gw_snapshot_bytes(gw_env* env, uint64_t* bytes);
int gw_get_snapshot(gw_env* env, void* dst_host, uint64_t bytes);
int gw_set_state(gw_env* env, const void* src_host, uint64_t bytes);

/* static link tables (host side, for tests): attenuation dB, rx power mW, thermal mW,
 * and the rx-power state machine used instead of per-env f64 noise state */
int gw_link_info(gw_env* env, int32_t from, int32_t to, double* attenuation_db, double* rx_power_mw);
int gw_noise_states(gw_env* env, int32_t radio, int32_t* count, double* values_mw /* [16] */);

/* ---------------------------------------------------------------------------------------------------
 * Linear plant (BASELINE config 4): the plant side of the pendulum env.
 *   replaces  OdePlant.updateState -- "advance the plant to the current simulated time, lazily"
 *             (gymwipe/plants/core.py:38-49) and the SlidingPendulum getters / setMotorVelocity
 *             (gymwipe/plants/sliding_pendulum.py:57-85)
 * with a fixed-dt linear model  x <- A x + B u  (4 states, 1 input), n = round((now - last)/dt) substeps
 * per call, evaluated on the f64 matrix cores (v_mfma_f64_16x16x4_f64).  BUILDER-DEFINED: the
 * reference's plant is an ODE rigid-body world (py3ode, absent) inside an env that cannot be
 * constructed (simtools.py:39-42), so there is nothing to pin parity to; the oracle is this library's
 * own scalar restatement of the same recurrence (test tree), tolerance 1e-5 relative.
 * ------------------------------------------------------------------------------------------------- */
#define GW_PLANT_DIM   4
#define GW_PLANT_KMAX  32                   /* substeps applied per matrix-core pass */

typedef struct gw_plant_config {
    int32_t abi_version;                    /* GW_ABI_VERSION */
    int32_t hip_device;
    int64_t num_envs;
    double  A[GW_PLANT_DIM * GW_PLANT_DIM]; /* row-major one-substep state matrix */
    double  B[GW_PLANT_DIM];                /* one-substep input vector */
    double  dt;                             /* substep length, s */
    double  x0[GW_PLANT_DIM];               /* initial state {wagon pos, wagon vel, angle, angle rate} */
    double  u0;                             /* initial input (motor velocity; reference: 0.1) */
} gw_plant_config;

typedef struct gw_plant gw_plant;

/* pendulum-like default: velocity-servoed wagon, small-angle pendulum, forward-Euler at dt = 1 ms */
int gw_plant_config_default(gw_plant_config* cfg, int64_t num_envs);
int gw_plant_create(const gw_plant_config* cfg, gw_plant** out);
int gw_plant_destroy(gw_plant* p);
/* advance env e to time *(double*)((char*)now_dev + e*stride_bytes); no-op where that is not later than its last update */
int gw_plant_update(gw_plant* p, const void* now_dev, int64_t stride_bytes, void* stream);
/* u[e] <- u_dev[e] where mask_dev is NULL or mask_dev[e] != 0 (setMotorVelocity; the caller updates first) */
int gw_plant_set_input(gw_plant* p, const double* u_dev, const uint8_t* mask_dev, void* stream);
/* InvertedPendulumInterpreter (envs/inverted_pendulum.py:27-57) for every plant: obs = int(degrees(angle)),
 * reward = float(abs(180 - degrees(angle))), optionally the angle in degrees ("Sensor angle").  Any pointer may be NULL. */
int gw_plant_feedback(gw_plant* p, int32_t* obs_dev, float* reward_dev, double* angle_deg_dev, void* stream);
/* gw_plant_update and gw_plant_feedback in one launch (what an env.step() of the pendulum env needs after the network step) */
int gw_plant_update_feedback(gw_plant* p, const void* now_dev, int64_t stride_bytes, int32_t* obs_dev, float* reward_dev,
                             double* angle_deg_dev, void* stream);
/* env.step() of the pendulum env (InvertedPendulumEnv.step, envs/inverted_pendulum.py:101-113) in ONE launch: gw_step of the env's
 * network (two assignable devices, default queue mode; its own CounterTraffic feedback is not produced) + gw_plant_update to the
 * env's new clock + gw_plant_feedback.  Same results as the three calls.  Any output pointer may be NULL. */
int gw_pendulum_step(gw_env* env, gw_plant* p, const int32_t* device_dev, const int32_t* duration_dev, int32_t* obs_dev,
                     float* reward_dev, double* angle_deg_dev, void* stream);
/* device pointer to the state, double[N][4] (row e = {pos, vel, angle, rate}); valid until gw_plant_destroy */
int gw_plant_state_ptr(gw_plant* p, double** x_dev);
/* host copies for tests: "x" f64[N][4] | "u" f64[N] | "t_last" f64[N] | "substeps" u64[N] */
int gw_plant_get_state(gw_plant* p, const char* field, void* dst_host, size_t bytes);

/* device pointer + stride of the per-env simulated time of a gw_env (default mode), for gw_plant_update */
int gw_now_ptr(gw_env* env, const void** now_dev, int64_t* stride_bytes);

/* ---------------------------------------------------------------------------------------------------
 * PHY grid: N independent replicas of the reference's benchmark scenario
 * (tests/test_benchmark.py:20-91) -- n uncoordinated PHY-only devices, each sending one 39-byte packet
 * every 10 ms at 40 dBm from its own phase.  The one workload of the reference with CONCURRENT
 * transmissions: every radio sums the power of all active transmissions (simple_stack.py:99-157) and
 * re-integrates its bit errors at every power change (:161-188,:214-267); a device that wants to send
 * while receiving waits for the reception to end (:199-200).
 *   replaces  SimMan.runSimulation(seconds) over SendingDevice / SimplePhy / FrequencyBand
 * One wave per replica, one lane per radio; events are popped in SimPy's (time, priority, insertion id)
 * order.  Integer outcomes are compared bit-exact with the event-driven oracle; BER uses the device
 * libm (log10/pow/sqrt), so floating-point state is compared to 1e-9 relative.
 * ------------------------------------------------------------------------------------------------- */
#define GW_GRID_MAX_DEVICES 64

typedef struct gw_grid_config {
    int32_t abi_version;                    /* GW_ABI_VERSION */
    int32_t hip_device;
    int64_t num_envs;                       /* replicas */
    int32_t num_devices;                    /* n <= GW_GRID_MAX_DEVICES */
    int32_t mobile;                         /* 1: every device also random-walks (mobile_device_grid, :73-85) */
    double  pos[GW_GRID_MAX_DEVICES][2];    /* metres (default: (i / cols, i % cols), cols = int(sqrt(n))) */
    double  slot, frequency, bandwidth, temperature_c, bit_rate, code_rate, max_ber;
    double  tx_power_dbm;                   /* 40.0   tests/test_benchmark.py:47 */
    double  send_interval;                  /* 1e-2   :17 */
    int32_t header_bytes;                   /* 13 */
    int32_t payload_bytes;                  /* 26 = len("A message to all my homies") */
    double  move_interval;                  /* 1e-3   tests/test_benchmark.py:18 */
    double  move_span;                      /* 0.2: offsets uniform(-span, span) per axis and move (:79-80) */
    uint64_t seed;                          /* the walk of replica e, device i, move k is splitmix64(seed, e, i, k) */
} gw_grid_config;

typedef struct gw_grid gw_grid;

int gw_grid_config_default(gw_grid_config* cfg, int64_t num_envs, int32_t num_devices);
/* initial_delays_host: double[N][n], the random.uniform(0, SEND_INTERVAL) of tests/test_benchmark.py:67 */
int gw_grid_create(const gw_grid_config* cfg, const double* initial_delays_host, gw_grid** out);
int gw_grid_destroy(gw_grid* g);
/* SimMan.runSimulation(seconds) for every replica */
int gw_grid_run(gw_grid* g, double seconds, void* stream);
/* Position.set(x, y) of device `device` in every replica at the replicas' current time (devices/core.py:77-86):
 * the attenuation of every link of that device changes and every radio that is hearing a transmission over
 * such a link updates its received power and re-integrates its bit errors (simple_stack.py:119-128).
 * x_host / y_host: double[N].  Needs cfg.mobile != 0 (per-replica geometry); set move_interval very large to
 * have scripted moves only. */
int gw_grid_set_position(gw_grid* g, int32_t device, const double* x_host, const double* y_host, void* stream);
/* host copies: "now" f64[N] | "events","n_tx","flags","on_air" u32[N] (on_air = transmissions currently active) | "n_sent","hdr_ok","hdr_fail","pay_ok","pay_fail" u32[N][n]
 *              | "rx_power" f64[N][n] | "pos" f64[N][n][2] */
int gw_grid_get_state(gw_grid* g, const char* field, void* dst_host, size_t bytes);

/* ---- Control loop (second half): the pendulum env with its loop closed -------------------------------
 * BUILDER-DEFINED: the reference intends this loop (plants/sliding_pendulum.py:116-155, control/inverted_pendulum.py:16-69,
 * envs/inverted_pendulum.py:60-113) but never runs it -- nobody sets `receiving`, and the env cannot be constructed.  Three
 * network devices (sensor 0, controller 1, actuator 2 -- not assignable) and the RRM; the sensor queues the plant angle
 * every counter tick and the plant then advances one substep x <- A x + B u; every ctrl_period_ticks from ctrl_start_tick on
 * the controller queues -angle_deg (the shipped PID gains kp = 1, ki = kd = 0) unless its angle is 0; packets decoded by
 * their destination's receive-mode MAC are handed up (controller: angle := degrees(value); actuator: u := value);
 * observation and reward as InvertedPendulumInterpreter computes them.  Checked bit for bit against an event-driven
 * model of the same rules; parity with the reference is unpinned by construction. */
typedef struct gw_ctrl_config {
    gw_config net;                          /* num_devices must be 3; geometry, radio constants, counter_interval */
    double  A[16], B[4];                    /* one-substep plant matrices (row-major), substep = counter_interval */
    double  x0[4], u0;                      /* initial plant state {wagon pos, wagon vel, angle, angle rate}, initial input */
    int32_t ctrl_start_tick;                /* first control tick */
    int32_t ctrl_period_ticks;              /* control period in ticks (10 = the reference's 10 ms) */
} gw_ctrl_config;

typedef struct gw_ctrl gw_ctrl;

int gw_ctrl_config_default(gw_ctrl_config* cfg, int64_t num_envs);
int gw_ctrl_create(const gw_ctrl_config* cfg, gw_ctrl** out);
int gw_ctrl_destroy(gw_ctrl* c);
/* one env.step() of every env: device in {0 sensor, 1 controller}, duration in [0, max_duration); angle_deg_dev may be NULL */
int gw_ctrl_step(gw_ctrl* c, const int32_t* device_dev, const int32_t* duration_dev, int32_t* obs_dev, float* reward_dev,
                 double* angle_deg_dev, void* stream);
/* host copies: "now","wake","u","angle_deg" f64[N] | "x" f64[N][4] | "rx_power" f64[N][4] | "qlen" int32[N][2] |
 *              "received" u32[N][2] (controller, actuator) | "n_tx","commands","substeps","flags" u32[N] */
int gw_ctrl_get_state(gw_ctrl* c, const char* field, void* dst_host, size_t bytes);

/* Host-only self-test hook (no GPU needed): fuzzes the MAC-queue encoding the default kernel uses
 * against an explicit deque(maxlen=100).  Returns the number of mismatches (0 = identical). */
int gw_selftest_queue(uint64_t seed, int32_t operations, int32_t mult, int32_t counter_bound);

/* Host-only: the same fuzz for the run-length queues of the generic kernel (GW_CFG_EXPLICIT_QUEUE, csrc/gw_runq.h): counter ticks,
 * resets, pops and arbitrary gw_enqueue packets; mult up to GW_QUEUE_CAP.  Returns the number of mismatches. */
int gw_selftest_runq(uint64_t seed, int32_t operations, int32_t mult, int32_t counter_bound);

/* Host-only: bit mask of the exact arithmetic fast paths gw_create enables for cfg after validating
 * them (1 slot remainder, 2 division by the data rate, 4 integer decode decision, 8 idempotent
 * noise-state map, 16 counter ticks counted in one jump); negative on error.  max_noise_states may be NULL. */
int gw_selftest_fastmath(const gw_config* cfg, int32_t* max_noise_states);

#ifdef __cplusplus
}
#endif
#endif /* GYMWIPE_AMD_H */
