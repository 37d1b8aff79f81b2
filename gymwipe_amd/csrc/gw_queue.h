// gw_queue.h -- "suffix" encoding of the MAC queues of CounterTraffic, shared by the HIP kernel
// (ct_step_sfx.hip) and the host (state reader, reset bookkeeping, gw_selftest_queue fuzz).
//
// Facts of the reference this rests on (gymwipe/envs/counter_traffic.py):
//   * every sender runs the same process: every COUNTER_INTERVAL it enqueues `mult` packets of
//     25 + counter bytes and increments its counter while < COUNTER_BOUND (:53-61); all senders start
//     at t = 0 and reset() zeroes all counters together (:139-140), so tick times and counter values
//     are identical across the senders of one env;
//   * a MAC queue is a deque(maxlen=100) (simple_stack.py:361): packets only ever leave from the HEAD
//     (window pops :425, drop-oldest on append :469).
// Hence the queue of sender i is exactly the LAST len_i packets of its append stream; packet number a
// of that stream belongs to tick a / mult_i, and the stream has mult_i * tau packets after tau ticks.
// Per sender the state is ONE BYTE (len_i); per env a tick counter tau and the list of "breakpoints"
// (tick, counter value) where a new counting sequence started (env creation, each reset()).
//   append k ticks :  len_i = min(len_i + k*mult_i, 100)          (drop-oldest is the min)
//   pop            :  len_i -= 1
//   head packet    :  tick tau - ceil(len_i / mult_i), size 25 + value(tick)
//   value(t)       :  min(c0 + (t - t0), bound) for the latest breakpoint (t0, c0) with t0 <= t
// The encoding is exact; tests expand it and compare with the oracle's explicit deque.
#pragma once
#include <stdint.h>
#include "gw_internal.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GW_HD __host__ __device__ __forceinline__
#else
#define GW_HD inline
#endif

struct GwTally { uint32_t app, pop, drop, tx, deliv; };

GW_HD uint32_t gw_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// ceil(len / mult) for len <= GW_QUEUE_CAP, mult <= GW_MAX_MULT, with inv16 = ceil(65536 / mult) (exact up to mult 256)
GW_HD uint32_t gw_ceil_div(uint32_t len, uint32_t mult, uint32_t inv16)
{
    return ((len + mult - 1u) * inv16) >> 16;
}

// counter value of tick t (t within the last GW_QUEUE_CAP ticks).  cur/prev are the two newest
// breakpoints (entries nbp-1 and nbp-2 of the history ring); older ones are read from `hist`.
GW_HD uint32_t gw_tick_value(uint32_t t, GwBp cur, GwBp prev, uint32_t nbp, const GwBp* hist, uint32_t bound)
{
    const bool older = t < cur.t0;               // (a select, not a branch: the common cases stay straight-line)
    GwBp b;
    b.t0 = older ? prev.t0 : cur.t0;
    b.c0 = older ? prev.c0 : cur.c0;
    if (older && t < prev.t0) {                  // more than two resets inside the queue's span: rare
        uint32_t j = nbp - 2u;
        do { --j; b = hist[j & GW_RING_MASK]; } while (t < b.t0);
    }
    return gw_min_u32(b.c0 + (t - b.t0), bound);
}

// k counter ticks for one sender: returns the new length, counts appends and drop-oldest events
GW_HD uint32_t gw_len_after_ticks(uint32_t len, uint32_t k, uint32_t mult, GwTally& t)
{
    const uint32_t add = k * mult;
    const uint32_t want = len + add;
    const uint32_t now = gw_min_u32(want, (uint32_t)GW_QUEUE_CAP);
    t.app += add;
    t.drop += want - now;
    return now;
}
