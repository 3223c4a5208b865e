// The ordering rules of the queued-query scheduler (Engine::compute_powers), as pure functions of a small state: which pooled powers
// buffer a query takes, which walk of the PowersDag it gets (one stream / low and high halves on two streams / the whole walk on the
// second stream next to the evaluation of the query in front), and which events each stream waits for before it may write the buffer.
// No HIP in here: Engine::compute_powers fills the state from its event queries and executes the plan; tests/test_host_logic.py
// enumerates every state through the CPU emulation library and holds the plan to the invariant the one real race of the project
// violated (round 5: a pooled buffer kept the `last_use` mark of an OLDER evaluation):
//     a buffer whose writers or readers may still be queued on either stream is never written by a stream that has not been ordered
//     behind all of them.
// The reference gets this isolation for free -- a fresh `all_powers` per Receiver::RunQuery (receiver/apsu/receiver_osn.cpp:286-302).
#pragma once
#include <cstddef>

namespace apsu_he {

// ---- which pooled buffer -------------------------------------------------------------------------------------------------------
struct PoolEntryState {
    bool fits;            // same three allocation sizes as this query needs
    bool last_use_set;    // an evaluation has read it since it was last written (Powers::last_use_set)
    bool last_use_done;   // ... and that evaluation is over (hipEventQuery(last_use) == hipSuccess); meaningless when !last_use_set
};
// index of the pooled buffer to take, or -1 for a new one.  Without the caller's overlap promise (inputs_ready) the first that fits; with
// it the first whose reader is done (an unmarked one counts as takeable: the walk then orders itself conservatively, plan_walk), and when
// three that fit are all busy the oldest of them -- the host runs at most two queries ahead: one buffer is being read, one is written
// or waits for its evaluation, the third takes the next query.
inline int pick_pooled_buffer(const PoolEntryState *pool, size_t count, bool inputs_ready)
{
    size_t fits = 0;
    int pick = -1, first = -1;
    for (size_t i = 0; i < count; i++) {
        if (!pool[i].fits) continue;
        fits++;
        if (first < 0) first = (int)i;
        if (pick < 0 && (!inputs_ready || !pool[i].last_use_set || pool[i].last_use_done)) pick = (int)i;
    }
    if (pick < 0 && (fits >= 3 || !inputs_ready)) pick = first;
    return pick;
}

// ---- which walk, behind which events ---------------------------------------------------------------------------------------------
struct WalkState {
    bool recycled;        // the buffer came from the pool (else: freshly allocated, nobody has touched it)
    bool last_use_set;    // recycled: an evaluation has read it since it was last written
    bool last_use_done;   // recycled && last_use_set: that evaluation is over
    bool high_async;      // recycled: its previous walk wrote (part of) it on the second stream and recorded high_ready behind that
    bool split_ok;        // the PowersDag splits into a low and a high half
    bool prof_on;         // per-kernel event profiling (always one stream)
    int split_mode;       // -1 default, 0 one stream, 1 two streams (apsu_he_set_two_stream, then APSU_HE_SPLIT)
    bool pipe_cp;         // apsu_he_set_query_overlap modes 1 and 3: the pipelined walk is allowed
    bool force_pipe;      // mode 3: ... and taken whether or not the device is busy
    bool inputs_ready;    // modes 1-3: the caller's promise that sources and keys are complete when the call is made
    bool on_device;       // device-resident sources
    bool device_busy;     // an evaluation queued earlier is still running
};
enum WalkKind { WALK_ONE_STREAM = 0, WALK_SPLIT = 1, WALK_PIPELINED = 2 };
struct WalkPlan {
    int walk;                     // WalkKind
    bool main_waits_high_ready;   // main stream: behind the second stream's last writer of this buffer (its previous walk's high_ready)
    bool side_waits_last_use;     // second stream: behind the buffer's last reader (the evaluation that marked it)
    bool side_waits_main;         // second stream: behind everything queued on the main stream so far (ev_main_)
    bool consumes_last_use;       // the mark is spent: only an evaluation of THESE powers sets it again
};
inline WalkPlan plan_walk(const WalkState &s)
{
    WalkPlan p{};
    const bool split = s.split_ok && !s.prof_on && (s.split_mode < 0 || s.split_mode == 1);
    // a pooled buffer is ordered behind its LAST READER only when an evaluation has read it since it was last written; a buffer that
    // was computed and given back without one keeps no mark: its writers may still be queued on EITHER stream
    const bool had_last_use = s.recycled && s.last_use_set;
    const bool buffer_known = !s.recycled || had_last_use;
    const bool buffer_idle = !s.recycled || (had_last_use && s.last_use_done);
    const bool pipe = s.pipe_cp && (s.device_busy || s.force_pipe) && (buffer_idle || (s.force_pipe && buffer_known)) && split && s.inputs_ready && s.on_device;
    p.main_waits_high_ready = s.recycled && s.high_async;
    p.consumes_last_use = true;
    if (pipe) {
        p.walk = WALK_PIPELINED;
        p.side_waits_last_use = had_last_use;
    } else if (!split) {
        p.walk = WALK_ONE_STREAM;
    } else {
        p.walk = WALK_SPLIT;
        // with device-resident inputs the caller has declared complete the second stream waits for the buffer's last reader, not for
        // everything the main stream has queued; a pooled buffer whose reader left no mark: wait for all
        const bool early = s.inputs_ready && s.on_device && buffer_known;
        p.side_waits_main = !early;
        p.side_waits_last_use = early && had_last_use;
    }
    return p;
}

} // namespace apsu_he
