// gw_runq.h -- run-length MAC queues of the GENERIC step kernel (GW_CFG_EXPLICIT_QUEUE), shared by the HIP
// kernels (ct_step.hip), the host state reader (gw_api.cpp) and the host-side fuzz (gw_selftest_runq).
//
// The reference's SimpleMac queue is a deque(maxlen=100) of packets (simple_stack.py:361): appended at the tail by
// SimpleNetworkDevice.send (networking/devices.py:84-86 -> simple_stack.py:463-471, drop-oldest when full), popped at
// the head by the window loop (:425).  Any traffic is a sequence of packet byte sizes; it is held here as a deque of RUNS:
//   counter run  {v0, n, j}: n packets, packet i has size min(v0 + (j + i) / mult, cap) -- what `mult` packets per
//                counter tick of 25 + counter bytes look like (counter_traffic.py:53-61), cap = 25 + COUNTER_BOUND;
//                j < mult is how far into its tick's group of `mult` the first packet is;
//   literal run  {v0}: one packet of any size (gw_enqueue).
// A tick EXTENDS the tail run when it continues it (same law, group boundary), a pop or a drop ADVANCES the head run, so
// a step of counter traffic touches only the two runs at the ends -- and those live, with the queue's bookkeeping, in ONE
// 16-byte record per (sender, env), loaded and stored coalesced.  Runs in between (after a reset() restarted the
// counters, or interleaved gw_enqueue packets) wait in a ring of 128 runs per (env, sender) in HBM that a step touches
// only when a run is created or exhausted.  100 literal packets are 100 runs: nothing is approximated for any traffic.
//
// Record (uint4):  x = head.v0   y = head.n | head.j << 8 | tail.n << 16 | tail.j << 24   z = tail.v0
//                  w = state (2 bits: 0 empty, 1 one run = head, 2 head + M middle runs + tail) | head.lit << 2
//                      | tail.lit << 3 | mid_head << 8 | M << 16 | len << 24
#pragma once
#include <stdint.h>
#include "gw_internal.h"
#include "gw_queue.h"       // GW_HD, GwTally

struct GwRun { uint32_t v0, n, j, lit; };

struct GwRunQ {
    GwRun H, T;
    uint32_t state, mid_head, M, len;
};

GW_HD uint64_t gw_run_pack(const GwRun& r)
{
    return (uint64_t)r.v0 | ((uint64_t)(r.n & 0xffu) << 32) | ((uint64_t)(r.j & 0xffu) << 40) | ((uint64_t)(r.lit & 1u) << 48);
}
GW_HD GwRun gw_run_unpack(uint64_t w)
{
    GwRun r;
    r.v0 = (uint32_t)w; r.n = (uint32_t)(w >> 32) & 0xffu; r.j = (uint32_t)(w >> 40) & 0xffu; r.lit = (uint32_t)(w >> 48) & 1u;
    return r;
}

GW_HD GwRunQ gw_runq_unpack(const GwRec& r)
{
    GwRunQ q;
    q.H.v0 = r.x; q.H.n = r.y & 0xffu; q.H.j = (r.y >> 8) & 0xffu; q.H.lit = (r.w >> 2) & 1u;
    q.T.v0 = r.z; q.T.n = (r.y >> 16) & 0xffu; q.T.j = r.y >> 24; q.T.lit = (r.w >> 3) & 1u;
    q.state = r.w & 3u; q.mid_head = (r.w >> 8) & 0xffu; q.M = (r.w >> 16) & 0xffu; q.len = r.w >> 24;
    return q;
}
GW_HD GwRec gw_runq_pack(const GwRunQ& q)
{
    GwRec r;
    r.x = q.H.v0;
    r.y = (q.H.n & 0xffu) | ((q.H.j & 0xffu) << 8) | ((q.T.n & 0xffu) << 16) | ((q.T.j & 0xffu) << 24);
    r.z = q.T.v0;
    r.w = (q.state & 3u) | ((q.H.lit & 1u) << 2) | ((q.T.lit & 1u) << 3) | ((q.mid_head & 0xffu) << 8) | ((q.M & 0xffu) << 16) | (q.len << 24);
    return r;
}

// size of packet i of a run (host-side expansion; the kernels only ever need i = 0, which is v0 itself)
GW_HD uint32_t gw_run_value(const GwRun& r, uint32_t i, uint32_t mult, uint32_t cap)
{
    if (r.lit) return r.v0;
    const uint32_t v = r.v0 + (r.j + i) / mult;
    return v < cap ? v : cap;
}

// drop the first t packets of a run (t <= n).  inv20 = ceil(2^20 / mult): (j + t) / mult exactly for (j + t) * mult < 2^20
GW_HD void gw_run_advance(GwRun& r, uint32_t t, uint32_t mult, uint32_t inv20, uint32_t cap)
{
    r.n -= t;
    if (r.lit) return;
    const uint32_t jj = r.j + t;
    const uint32_t q = (jj * inv20) >> 20;
    const uint32_t v = r.v0 + q;
    r.j = jj - q * mult;
    r.v0 = v < cap ? v : cap;
}

// drop the first t packets of a run, as selects (t == 0 changes nothing; a literal run only counts down)
GW_HD void gw_run_advance_sel(GwRun& r, uint32_t t, uint32_t mult, uint32_t inv20, uint32_t cap)
{
    const uint32_t jj = r.j + t;
    const uint32_t q = r.lit ? 0u : (jj * inv20) >> 20;
    const uint32_t v = r.v0 + q;
    r.n -= t;
    r.j = r.lit ? r.j : jj - q * mult;
    r.v0 = r.lit ? r.v0 : (v < cap ? v : cap);
}

// remove `count` packets from the head while NO run waits in the ring (M == 0: the queue is its head run, or head + tail):
// straight-line, nothing but selects.  On the GPU every `if` of the general form below is an exec-mask region of its own
// (three scalar instructions and, in this kernel, a scalar-register reload), and a wave with one lane per env runs them
// all: in-kernel stamps put one call of the general form at ~1 000 cycles.
GW_HD void gw_runq_pop_front_m0(GwRunQ& q, uint32_t count, uint32_t mult, uint32_t inv20, uint32_t cap)
{
    q.len -= count;
    const uint32_t t1 = count < q.H.n ? count : q.H.n;
    gw_run_advance_sel(q.H, t1, mult, inv20, cap);
    const uint32_t rem = count - t1;
    const bool used_up = count != 0u && q.H.n == 0u;     // head run used up: the tail moves up (or the queue is empty)
    const bool two = q.state == 2u;
    q.H.v0 = (used_up && two) ? q.T.v0 : q.H.v0;  q.H.n = (used_up && two) ? q.T.n : q.H.n;
    q.H.j = (used_up && two) ? q.T.j : q.H.j;    q.H.lit = (used_up && two) ? q.T.lit : q.H.lit;
    q.state = used_up ? (two ? 1u : 0u) : q.state;
    gw_run_advance_sel(q.H, rem, mult, inv20, cap);      // rem != 0 only after the move (count <= len)
    q.state = (rem != 0u && q.H.n == 0u) ? 0u : q.state;
}

// remove `count` packets from the head (window pops, drop-oldest); count <= len
GW_HD void gw_runq_pop_front(GwRunQ& q, uint32_t count, uint64_t* ring, uint32_t mult, uint32_t inv20, uint32_t cap)
{
    if (q.M == 0u) { gw_runq_pop_front_m0(q, count, mult, inv20, cap); return; }
    q.len -= count;
    while (count > 0u) {
        const uint32_t t = count < q.H.n ? count : q.H.n;
        gw_run_advance(q.H, t, mult, inv20, cap);
        count -= t;
        if (q.H.n == 0u) {                               // head run used up: the next run moves up
            if (q.state == 2u) {
                if (q.M > 0u) {
                    q.H = gw_run_unpack(ring[q.mid_head]);
                    q.mid_head = (q.mid_head + 1u) & GW_RING_MASK;
                    q.M--;
                } else {
                    q.H = q.T;
                    q.state = 1u;
                }
            } else {
                q.state = 0u;
            }
        }
    }
}

// make `r` the new tail run
GW_HD void gw_runq_push_run(GwRunQ& q, const GwRun& r, uint64_t* ring)
{
    if (q.state == 2u) {                                 // the old tail becomes a middle run
        ring[(q.mid_head + q.M) & GW_RING_MASK] = gw_run_pack(q.T);
        q.M++;
    }
    // into the head slot of an empty queue, else the tail slot -- as selects on the fields: a store through a pointer
    // that is either &q.H or &q.T would put the whole queue on the stack (GPU scratch memory)
    const bool empty = q.state == 0u;
    q.H.v0 = empty ? r.v0 : q.H.v0;  q.H.n = empty ? r.n : q.H.n;  q.H.j = empty ? r.j : q.H.j;  q.H.lit = empty ? r.lit : q.H.lit;
    q.T.v0 = empty ? q.T.v0 : r.v0;  q.T.n = empty ? q.T.n : r.n;  q.T.j = empty ? q.T.j : r.j;  q.T.lit = empty ? q.T.lit : r.lit;
    q.state = empty ? 1u : 2u;
}

// append `add` packets of counter traffic whose first packet has size v (= 25 + counter, starting a tick's group).
// The caller removes what exceeds the deque's capacity afterwards (gw_runq_pop_front): n may pass 100 in between.
GW_HD void gw_runq_append_counter(GwRunQ& q, uint32_t add, uint32_t v, uint64_t* ring, uint32_t mult, uint32_t inv20, uint32_t cap)
{
    if (add == 0u) return;
    q.len += add;
    if (q.state != 0u) {
        // the last run, by value (selecting a REFERENCE to one of the two would put the queue on the stack)
        const bool one = q.state == 1u;
        const uint32_t lv0 = one ? q.H.v0 : q.T.v0, ln = one ? q.H.n : q.T.n, lj = one ? q.H.j : q.T.j, llit = one ? q.H.lit : q.T.lit;
        const uint32_t e = lj + ln;
        const uint32_t qq = (e * inv20) >> 20;
        const uint32_t nv = lv0 + qq;
        if (!llit && e - qq * mult == 0u && (nv < cap ? nv : cap) == v) {   // the tick continues the tail run
            q.H.n += one ? add : 0u;
            q.T.n += one ? 0u : add;
            return;
        }
    }
    GwRun r;
    r.v0 = v; r.n = add; r.j = 0u; r.lit = 0u;
    gw_runq_push_run(q, r, ring);
}

// SimpleNetworkDevice.send: one packet of `size` bytes (the caller dropped the oldest first if the deque was full)
GW_HD void gw_runq_append_literal(GwRunQ& q, uint32_t size, uint64_t* ring)
{
    GwRun r;
    r.v0 = size; r.n = 1u; r.j = 0u; r.lit = 1u;
    q.len += 1u;
    gw_runq_push_run(q, r, ring);
}

// k counter ticks at counter value c (counter_traffic.py:53-61 -> simple_stack.py:463-471): append, then drop-oldest
GW_HD void gw_runq_ticks(GwRunQ& q, uint32_t k, uint32_t c, uint32_t bound, uint32_t base_bytes, uint64_t* ring,
                         uint32_t mult, uint32_t inv20, GwTally& t)
{
    if (k == 0u || mult == 0u) return;
    if (k > (uint32_t)GW_QUEUE_CAP) {
        // A long step (counter_interval far below the step's length: thousands of ticks in one call).  Only the packets of
        // the last ceil(CAP / mult) ticks can be in a deque(maxlen=CAP) afterwards: everything queued now and every packet
        // of the earlier ticks is dropped on the way, so the queue restarts empty at the counter value those ticks left.
        // (Appending all k * mult packets first and popping the surplus, as below, is exact only while the run arithmetic's
        // reciprocal division holds -- (j + t) * mult < 2^20 -- which a call with more than ~4 000 ticks broke.)
        const uint32_t keep = ((uint32_t)GW_QUEUE_CAP + mult - 1u) / mult;
        const uint32_t skip = k - keep;
        t.app += skip * mult;
        t.drop += q.len + skip * mult;
        q.state = 0u; q.len = 0u; q.M = 0u; q.mid_head = 0u;
        q.H.n = 0u; q.T.n = 0u;
        c = (bound - (c < bound ? c : bound)) > skip ? c + skip : bound;      // min(c + skip, bound) without overflow
        k = keep;
    }
    const uint32_t add = k * mult;
    const uint32_t cap = base_bytes + bound;
    const uint32_t want = q.len + add;
    const uint32_t drops = want > (uint32_t)GW_QUEUE_CAP ? want - (uint32_t)GW_QUEUE_CAP : 0u;
    const uint32_t v = base_bytes + (c < bound ? c : bound);
    t.app += add;
    t.drop += drops;
    // ---- the usual case, straight-line: nothing waits in the ring and the append does not put anything there (it extends
    //      the last run, or starts the queue's first or second run) ----
    const bool none = q.state == 0u, one = q.state == 1u;
    const uint32_t lv0 = one ? q.H.v0 : q.T.v0, ln = one ? q.H.n : q.T.n, lj = one ? q.H.j : q.T.j, llit = one ? q.H.lit : q.T.lit;
    const uint32_t e = lj + ln;
    const uint32_t qq = (e * inv20) >> 20;
    const uint32_t nv = lv0 + qq;
    const bool cont = !none && !llit && e - qq * mult == 0u && (nv < cap ? nv : cap) == v;   // the tick continues the last run
    if (one && cont) {
        // the steady state of counter traffic: ONE run, and the ticks extend it -- lengthen it, advance its head past the drops
        // (drops < n + add: the queue keeps CAP packets)
        q.H.n += add;
        q.len = want - drops;
        gw_run_advance_sel(q.H, drops, mult, inv20, cap);
        return;
    }
    if (q.M == 0u && (cont || q.state != 2u)) {
        const bool to_h = none || (one && cont), to_t = !none && !(one && cont);            // which slot the packets go to
        const bool fresh = !cont;                                                            // ... as a new run
        q.H.v0 = (to_h && fresh) ? v : q.H.v0;   q.H.j = (to_h && fresh) ? 0u : q.H.j;   q.H.lit = (to_h && fresh) ? 0u : q.H.lit;
        q.H.n = to_h ? (fresh ? add : q.H.n + add) : q.H.n;
        q.T.v0 = (to_t && fresh) ? v : q.T.v0;   q.T.j = (to_t && fresh) ? 0u : q.T.j;   q.T.lit = (to_t && fresh) ? 0u : q.T.lit;
        q.T.n = to_t ? (fresh ? add : q.T.n + add) : q.T.n;
        q.state = none ? 1u : ((one && !cont) ? 2u : q.state);
        q.len = want;
        gw_runq_pop_front_m0(q, drops, mult, inv20, cap);
        return;
    }
    gw_runq_append_counter(q, add, v, ring, mult, inv20, cap);
    if (drops) gw_runq_pop_front(q, drops, ring, mult, inv20, cap);
}

// the window loop's pair of operations on the addressed sender's queue, back to back: pop the head packet (simple_stack.py:425),
// then the k counter ticks that fall into its transmission (k may be 0).  In the steady state -- one counter run of at least two
// packets which the ticks continue -- the two are ONE advance of the run's head by 1 + drops.
GW_HD void gw_runq_pop1_ticks(GwRunQ& q, uint32_t k, uint32_t c, uint32_t bound, uint32_t base_bytes, uint64_t* ring,
                              uint32_t mult, uint32_t inv20, GwTally& t)
{
    const uint32_t cap = base_bytes + bound;
    const uint32_t add = k * mult;
    const uint32_t e = q.H.j + q.H.n;                    // (invariant under advancing the head)
    const uint32_t qq = (e * inv20) >> 20;
    const uint32_t nv = q.H.v0 + qq;
    const uint32_t v = base_bytes + (c < bound ? c : bound);
    const bool cont = e - qq * mult == 0u && (nv < cap ? nv : cap) == v;
    if (q.state == 1u && q.H.lit == 0u && q.H.n >= 2u && k <= (uint32_t)GW_QUEUE_CAP && mult != 0u && (add == 0u || cont)) {
        const uint32_t want = q.len - 1u + add;
        const uint32_t drops = want > (uint32_t)GW_QUEUE_CAP ? want - (uint32_t)GW_QUEUE_CAP : 0u;
        t.app += add;
        t.drop += drops;
        q.H.n += add;
        q.len = want - drops;
        gw_run_advance_sel(q.H, 1u + drops, mult, inv20, cap);
        return;
    }
    gw_runq_pop_front(q, 1u, ring, mult, inv20, cap);
    gw_runq_ticks(q, k, c, bound, base_bytes, ring, mult, inv20, t);
}

// host: the queue's packets, head first (out has room for GW_QUEUE_CAP); returns the length
GW_HD uint32_t gw_runq_expand(const GwRunQ& q, const uint64_t* ring, uint32_t mult, uint32_t cap, uint32_t* out)
{
    uint32_t p = 0;
    if (q.state == 0u) return 0u;
    for (uint32_t i = 0; i < q.H.n && p < (uint32_t)GW_QUEUE_CAP; ++i) out[p++] = gw_run_value(q.H, i, mult, cap);
    if (q.state == 2u) {
        for (uint32_t m = 0; m < q.M; ++m) {
            const GwRun r = gw_run_unpack(ring[(q.mid_head + m) & GW_RING_MASK]);
            for (uint32_t i = 0; i < r.n && p < (uint32_t)GW_QUEUE_CAP; ++i) out[p++] = gw_run_value(r, i, mult, cap);
        }
        for (uint32_t i = 0; i < q.T.n && p < (uint32_t)GW_QUEUE_CAP; ++i) out[p++] = gw_run_value(q.T, i, mult, cap);
    }
    return p;
}
