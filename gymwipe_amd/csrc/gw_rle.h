// gw_rle.h -- run-length encoded MAC queue, shared by the HIP kernel (ct_step_rle.hip) and the
// host (gw_api.cpp: state reader and the gw_selftest_rle fuzz against an explicit deque).
// See ct_step_rle.hip for the rationale and the word layout.
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

struct RQ {
    uint32_t mhead, nruns, used, len;     // meta
    uint32_t hc, hn, tc, tn;              // head / tail run (counter of first live tick, ticks)
    uint32_t* mid;                        // middle runs of this (env, sender)
};

GW_HD void rq_unpack(RQ& q, uint32_t wx, uint32_t wy, uint32_t wz, uint32_t* mid)
{
    q.mhead = GW_META_MHEAD(wx); q.nruns = GW_META_NRUNS(wx);
    q.used = GW_META_USED(wx);   q.len = GW_META_LEN(wx);
    q.hc = GW_RUN_C(wy); q.hn = GW_RUN_N(wy);
    q.tc = GW_RUN_C(wz); q.tn = GW_RUN_N(wz);
    q.mid = mid;
}

GW_HD void rq_pack(const RQ& q, uint32_t& wx, uint32_t& wy, uint32_t& wz)
{
    wx = GW_META_PACK(q.mhead, q.nruns, q.used, q.len);
    wy = GW_RUN_PACK(q.hc, q.hn);
    wz = GW_RUN_PACK(q.tc, q.tn);
}

// counter value of the head packet (needs len > 0)
GW_HD uint32_t rq_head_value(const RQ& q)
{
    return q.nruns == 1 ? q.tc : q.hc;    // runs are stored capped at the bound
}

// remove x packets from the front (x <= len): pops and drop-oldest share this
GW_HD void rq_consume(RQ& q, uint32_t x, uint32_t mult, uint32_t bound)
{
    q.len -= x;
    uint32_t off = q.used + x, ticks = 0;
    if (off >= mult) { ticks = off / mult; off -= ticks * mult; }
    q.used = off;
    while (ticks > 0) {
        if (q.nruns == 1) {                           // the tail run is the head
            const uint32_t take = ticks < q.tn ? ticks : q.tn;
            q.tc = gw_min_u32(q.tc + take, bound);
            q.tn -= take;
            ticks -= take;
            if (q.tn == 0) { q.nruns = 0; ticks = 0; }
        } else {
            const uint32_t take = ticks < q.hn ? ticks : q.hn;
            q.hc = gw_min_u32(q.hc + take, bound);
            q.hn -= take;
            ticks -= take;
            if (q.hn == 0) {
                if (q.nruns == 2) {
                    q.nruns = 1;                      // the tail run becomes the head
                } else {
                    const uint32_t w = q.mid[q.mhead];
                    q.hc = GW_RUN_C(w); q.hn = GW_RUN_N(w);
                    q.mhead = (q.mhead + 1) & GW_RING_MASK;
                    q.nruns--;
                }
            }
        }
    }
}

// k (<= 20) consecutive ticks with counter values v0, gw_min_u32(v0+1,bound), ...: `mult` packets each.
// counter_traffic.py:53-61 -> devices.py:84-86 -> simple_stack.py:463-471
GW_HD void rq_append(RQ& q, uint32_t v0, uint32_t k, uint32_t mult, uint32_t bound, GwTally& t)
{
    if (q.nruns > 0 && v0 == gw_min_u32(q.tc + q.tn, bound)) {
        q.tn += k;                                    // continues the tail run
    } else {
        if (q.nruns == 0) {
            q.used = 0;
        } else if (q.nruns == 1) {
            q.hc = q.tc; q.hn = q.tn;                 // the old (only) run becomes the head
        } else {
            q.mid[(q.mhead + q.nruns - 2) & GW_RING_MASK] = GW_RUN_PACK(q.tc, q.tn);
        }
        q.tc = v0; q.tn = k;
        q.nruns++;
    }
    const uint32_t add = k * mult;
    q.len += add;
    t.app += add;
    if (q.len > GW_QUEUE_CAP) {                       // deque(maxlen=100): keep the newest 100
        const uint32_t x = q.len - GW_QUEUE_CAP;
        t.drop += x;
        rq_consume(q, x, mult, bound);
    }
}

GW_HD void rq_bulk(RQ& q, uint32_t v0, uint32_t k, uint32_t mult, uint32_t bound, GwTally& t)
{
    while (k > 0) {
        const uint32_t kk = k < 20u ? k : 20u;
        rq_append(q, v0, kk, mult, bound, t);
        v0 = gw_min_u32(v0 + kk, bound);
        k -= kk;
    }
}

