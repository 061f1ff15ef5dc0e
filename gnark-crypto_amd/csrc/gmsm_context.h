// Per-device context, workspace buffers and error plumbing shared by the engine (C ABI) and the per-group
// translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gmsm.h"

namespace gmsm {

int fail(int code, const std::string &msg);  // records the thread-local error text, returns code
void clear_last_error();                     // this thread's text: a failure that was handled inside a successful call

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return ::gmsm::fail(GMSM_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                                                     std::to_string(__LINE__) + ")");                        \
    } while (0)

// ------------------------------------------------------------------ per-device context
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return GMSM_OK;
        if (ptr) HIP_TRY(hipFree(ptr));
        ptr = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8;  // grow-only with slack
        HIP_TRY(hipMalloc(&ptr, want));
        cap = want;
        return GMSM_OK;
    }
    size_t release() {  // caller has made sure nothing on the device still uses the buffer; returns the bytes given back
        const size_t had = cap;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        return had;
    }
};

// Process-wide switches (gmsm_set_option / gmsm_get_option, include/gmsm.h). The two a deployment may want - the window
// width and the window-table policy - take their initial value from the environment ONCE, when the library is loaded
// (GMSM_C, GMSM_TABLES); nothing on a call path reads the environment. The other two exist for the tests of the
// point-range splits.
struct Options {
    std::atomic<unsigned> window_bits{0};  // GMSM_OPT_WINDOW_BITS: 0 = the measured table (preferred_c), 2..20 forced
    std::atomic<unsigned> tables{1};       // GMSM_OPT_TABLES: 0 never, 1 the measured call sizes, 2 every size
    std::atomic<unsigned> max_run{0};      // GMSM_OPT_MAX_RUN: lower the 2^27-point cap of one pipeline run (0 = off)
    std::atomic<unsigned> host_ranges{0};  // GMSM_OPT_HOST_RANGES: force the point ranges of a host-buffer call (0 = off)
    std::atomic<unsigned> fixed_base_bits{0};  // GMSM_OPT_FIXED_BASE_BITS: table width of the fixed-base batch (0 = by size)
    std::atomic<unsigned> spin_wait_us{0};     // GMSM_OPT_SPIN_WAIT_US: poll a call's stream this long before blocking on it (0 = park at once)
    std::atomic<unsigned> small_bits{0};       // GMSM_OPT_SMALL_BITS: the fused small-n kernel: 0 = on, width by size; 1 = off; 2..7 = on, this width
    std::atomic<unsigned> split{0};            // GMSM_OPT_SPLIT (experiment): two window groups, the first group's fix-up + reduction beside the second's accumulation
    std::atomic<unsigned> small_max{0};        // GMSM_OPT_SMALL_MAX: largest call the fused kernel takes (0 = the measured default)
    std::atomic<unsigned> glv{1};              // GMSM_OPT_GLV: half scalars (gmsm_glv.h): 0 never, 1 the fused small-n kernel, 2 the sorted pipeline too
    std::atomic<unsigned> small_quad{0};       // GMSM_OPT_SMALL_QUAD: bucket phase of the fused kernel on lane quads: 0 by size, 1 never (narrow types), 2 always
};
Options &options();

extern std::atomic<unsigned long> g_small_runs;  // calls served by the fused small-n kernel (gmsm_debug_small_runs)
extern std::atomic<unsigned long> g_table_runs;  // pipeline runs that went through window tables (gmsm_debug_table_runs)

// Bases rewritten once into the lazy Montgomery domain and kept in HBM (gmsm_bases_register): the resident-SRS path.
// Shared ownership (std::shared_ptr): the handle table holds one reference, every call and every outstanding ticket
// holds another for as long as it may touch the buffers; gmsm_bases_release only drops the table's reference and the
// last owner frees the device memory.
struct ResidentBases {
    int group = -1;
    int device = -1;
    size_t n = 0;
    DeviceBuffer upoints, skip;
    // window tables (gmsm_bases_precompute; Group::precompute_tables): slab w = 2^(tab_c w) P_i, same packed layout as
    // upoints, tab_nw slabs of n points; tab_c == 0: none
    // tab_c is the publication point: written last (release) by precompute_tables, so a MultiExp running on another
    // thread either sees 0 and takes the plain path or sees the width with complete tables behind it. tab_mu serialises
    // concurrent gmsm_bases_precompute calls on one handle.
    std::atomic<unsigned> tab_c{0};
    unsigned tab_nw = 0;
    std::mutex tab_mu;
    DeviceBuffer tables;
    // narrow tables for the fused small-n kernel (Group::precompute_tables builds them next to the wide ones): slab w =
    // 2^(small_c w) P_i for the first small_m bases, small_nw slabs; small_c is their publication point (as tab_c)
    std::atomic<unsigned> small_c{0};
    unsigned small_nw = 0;
    size_t small_m = 0;
    DeviceBuffer small_tables;
    ResidentBases() = default;
    ResidentBases(const ResidentBases &) = delete;
    ResidentBases &operator=(const ResidentBases &) = delete;
    ~ResidentBases() {
        if (!upoints.ptr && !skip.ptr && !tables.ptr && !small_tables.ptr) return;
        int prev = 0;
        (void)hipGetDevice(&prev);
        if (device >= 0) (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();  // an enqueue-only call may still be reading the bases
        if (upoints.ptr) (void)hipFree(upoints.ptr);
        if (skip.ptr) (void)hipFree(skip.ptr);
        if (tables.ptr) (void)hipFree(tables.ptr);
        if (small_tables.ptr) (void)hipFree(small_tables.ptr);
        (void)hipSetDevice(prev);
    }
};


// One fft.Domain (ecc/bn254/fr/fft/domain.go:24-60) resident on a device: the constants on the host (Montgomery limbs),
// the twiddle and coset tables in HBM. Shared ownership like ResidentBases.
struct FftDomain {
    int group = -1;
    int device = -1;
    unsigned log2n = 0;
    std::vector<uint64_t> generator, generator_inv, cardinality_inv, shift, shift_inv;
    // Device tables. "Lazy domain" = the canonical element 2^(L*W-32N) * factor (gmsm_fft_lazy.h): what the lazy-limb
    // butterflies multiply by.
    DeviceBuffer twiddles_lz, twiddles_inv_lz;  // w^t, w^-t for t < n/2, lazy domain
    DeviceBuffer twiddles_rev_lz, twiddles_inv_rev_lz;  // the same in bit-reversed order (top-down passes), on first use
    bool rev_ready = false;
    DeviceBuffer coset, coset_inv_scaled;       // u^i and u^-i / n for i < n (lazy domain), built by the first coset transform
    std::vector<uint64_t> cardinality_inv_lz;   // 1/n, lazy domain
    bool coset_ready = false;
    std::mutex mu;                            // serialises the lazy coset build and transforms that share the tables
    FftDomain() = default;
    FftDomain(const FftDomain &) = delete;
    FftDomain &operator=(const FftDomain &) = delete;
    ~FftDomain() {
        DeviceBuffer *bufs[] = {&twiddles_lz, &twiddles_inv_lz, &twiddles_rev_lz, &twiddles_inv_rev_lz, &coset, &coset_inv_scaled};
        bool any = false;
        for (auto *b : bufs) any = any || b->ptr;
        if (!any) return;
        int prev = 0;
        (void)hipGetDevice(&prev);
        if (device >= 0) (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        for (auto *b : bufs)
            if (b->ptr) (void)hipFree(b->ptr);
        (void)hipSetDevice(prev);
    }
};

// Everything one in-flight MultiExp needs on the device: scratch buffers, a pinned host buffer for the window totals, a
// stream of its own (used when the caller gives none) and the stage events. A context owns three of them so that the
// asynchronous entry points (gmsm_multiexp_bases_submit / _collect) can keep two MultiExp calls in flight: the sort and
// accumulation of call i+1 overlap the latency-bound reduction, the copy-back and the host fold of call i.
struct Workspace {
    hipStream_t stream = nullptr;
    DeviceBuffer upoints, skip;  // bases rewritten into the lazy domain + infinity flags (non-resident calls)
    DeviceBuffer seg_lvl;        // hierarchical chain fixup: level partials, flags, long-chain flags
    DeviceBuffer seg_partials, seg_flags, seg_bucket;  // split-bucket partial sums of the segmented accumulation
    DeviceBuffer parted;         // coarse-partitioned references (two-level grouping)
    DeviceBuffer heavy;          // oversized partitions of the fine sort: their list, the count per window, sub-run bucket counts
    DeviceBuffer long_pieces;    // long chains of the fix-up: per-chain piece counters, the pieces' sums
    DeviceBuffer small_sums, small_done;  // fused small-n kernel: the slices' window totals, per-window arrival counters (kept zero)
    DeviceBuffer digits, sorted, blockhist, counts, starts, buckets, partials, totals;
    DeviceBuffer red_pre;        // per-thread (S, W) of the bucket reduction (k_reduce_serial -> k_combine_q)
    DeviceBuffer carry;          // running bucket sums of a multi-range host call (k_merge_buckets)
    hipStream_t mstream = nullptr;                         // merges + the one reduction of a multi-range call
    hipEvent_t ev_buckets = nullptr, ev_merged = nullptr;  // a range's buckets are complete / have been merged
    hipStream_t cstream = nullptr;                         // the base rewrite of an unregistered call, beside its scalar pipeline
    hipEvent_t ev_fork = nullptr, ev_conv = nullptr;       // inputs are ready on the call's stream / the rewrite is complete
    bool conv_pending = false;  // a forked base rewrite was launched on cstream and its join has not been enqueued (error path): begin_use joins it
    hipEvent_t events[12] = {nullptr};  // stage boundaries of the call in flight when profiling is on
    bool timed = false;                 // events[] were recorded by the last enqueue
    int timed_level = 0;                // ... at this profiling level (1: every stage, 2: the accumulation kernel only)
    void *pinned = nullptr;             // pinned host buffer for the window totals
    size_t pinned_cap = 0;
    DeviceBuffer h2d_points, h2d_scalars;  // staging of the host-pointer entries
    DeviceBuffer raw_bytes, flagword;      // point ingest: wire-format bytes, first-offender word
    bool busy = false;  // leased to a call (Context::acquire / release)
    bool ticket = false;  // ... by gmsm_multiexp_bases_submit: only gmsm_multiexp_collect ends that lease
    // state of a submitted, not yet collected call
    bool pending = false;
    int pending_group = -1;
    unsigned pending_c = 0;
    uint32_t pending_nw = 0;
    uint32_t pending_gen = 0;   // ticket generation (drawn from a process-wide counter: unique across gmsm_shutdown): a stale or repeated ticket is refused
    std::shared_ptr<ResidentBases> bases_ref;  // keeps the registered bases of the call in flight on this workspace alive
    hipEvent_t dep = nullptr;   // orders the workspace stream after the caller's stream (scalars produced there)
    bool uncollected = false;          // stage events of an enqueue-only call not yet added to the profile
    hipEvent_t last_use = nullptr;     // recorded after the last call enqueued on this workspace ...
    hipStream_t last_stream = nullptr; // ... on this stream: a call on another stream waits for it first
    bool pending_timed = false;
    // gmsm_trim / gmsm_shutdown: give back every scratch buffer larger than `keep` bytes (the caller holds the lease and
    // has synchronised the workspace's streams). Returns the device bytes released.
    size_t trim(size_t keep) {
        DeviceBuffer *all[] = {&upoints, &skip, &seg_lvl, &seg_partials, &seg_flags, &seg_bucket, &parted, &heavy, &long_pieces, &small_sums, &small_done, &digits, &sorted,
                               &blockhist, &counts, &starts, &buckets, &partials, &totals, &red_pre, &carry, &h2d_points,
                               &h2d_scalars, &raw_bytes, &flagword};
        size_t freed = 0;
        for (DeviceBuffer *b : all)
            if (b->cap > keep) freed += b->release();
        if (pinned && pinned_cap > keep) {
            (void)hipHostFree(pinned);
            pinned = nullptr;
            pinned_cap = 0;
        }
        return freed;
    }
    void destroy() {  // gmsm_shutdown: streams and events too
        (void)trim(0);
        hipEvent_t *evs[] = {&ev_buckets, &ev_merged, &dep, &last_use, &ev_fork, &ev_conv};
        for (hipEvent_t *e : evs) {
            if (*e) (void)hipEventDestroy(*e);
            *e = nullptr;
        }
        for (auto &e : events) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        if (stream) (void)hipStreamDestroy(stream);
        if (mstream) (void)hipStreamDestroy(mstream);
        if (cstream) (void)hipStreamDestroy(cstream);
        stream = mstream = cstream = nullptr;
        last_stream = nullptr;
        bases_ref.reset();
    }
    // Streams and events appear with the first lease of the workspace, the two side streams with their first use: creating a
    // stream costs 10-60 ms on this runtime (tools/hip_floor.hip), and a context used to create nine of them before its
    // first MultiExp - a small call needs one (profiles/r06_first_call.log: first 2^10 call 83 -> see INTEGRATION.md §4).
    int ensure_base() {
        if (stream) return GMSM_OK;
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ev_buckets, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_merged, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_conv, hipEventDisableTiming));
        return GMSM_OK;
    }
    int side_stream(hipStream_t &slot) {  // mstream / cstream on first use
        if (!slot) HIP_TRY(hipStreamCreateWithFlags(&slot, hipStreamNonBlocking));
        return GMSM_OK;
    }
    int ensure_pinned(size_t bytes) {
        if (bytes <= pinned_cap) return GMSM_OK;
        if (pinned) HIP_TRY(hipHostFree(pinned));
        pinned = nullptr;
        HIP_TRY(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
        pinned_cap = bytes;
        return GMSM_OK;
    }
};

// One per device. Calls lease one of the workspaces for their duration; the context lock protects nothing but the
// lease table, so two callers (two goroutines calling MultiExp, or the submit/collect pair) really overlap on the GPU:
// each runs on its workspace's stream, and the host-side wait, copy-back and fold of one happen while the other computes.
// THREE workspaces, of which submitted tickets may hold at most two (MAX_TICKETS): a blocking entry therefore always finds
// a workspace that only blocking callers cycle through, and never has to wait for somebody's gmsm_multiexp_collect.
// (Round 3 had two workspaces and a 2 s give-up timer for the case "both are uncollected tickets"; a slow collector then
// turned into a spurious error for an unrelated caller.) Scratch is allocated on first use, so an unused third
// workspace costs two streams and two events.
struct Context {
    static constexpr int NUM_WS = 3, MAX_TICKETS = 2;
    std::mutex mu;
    std::condition_variable cv;
    int device = -1;
    Workspace ws[NUM_WS];
    int num_cus = 256;
    // gmsm_shutdown never deletes a Context (a caller may be parked on its condition variable): it destroys the
    // workspaces' device resources, clears `live` and bumps `epoch`; waiters that wake into another epoch give up
    // (acquire returns nullptr, *why = 3) and the next get_context_for() runs init() again.
    uint64_t epoch = 0;
    std::atomic<bool> live{false};
    int init(int dev) {
        device = dev;
        HIP_TRY(hipSetDevice(dev));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        num_cus = prop.multiProcessorCount;
        // (the workspaces' streams and events: Workspace::ensure_base, with the first lease)
        std::lock_guard<std::mutex> lk(mu);
        live = true;
        return GMSM_OK;
    }
    // gmsm_shutdown, after it has leased and destroyed every workspace: the leases are dropped, waiters are told to give up
    void retire() {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (auto &w : ws) {
                w.busy = false;
                w.ticket = false;
                w.pending = false;
            }
            live = false;
            ++epoch;
        }
        cv.notify_all();
    }
    // Kernels that need more than the default 64 KiB of dynamic LDS: raise the limit once per kernel on this device.
    std::mutex attr_mu;
    std::vector<const void *> lds_allowed;
    int allow_lds(const void *kernel, int bytes) {
        std::lock_guard<std::mutex> lk(attr_mu);
        for (const void *k : lds_allowed)
            if (k == kernel) return GMSM_OK;
        HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        lds_allowed.push_back(kernel);
        return GMSM_OK;
    }
    // for_ticket (gmsm_multiexp_bases_submit): nullptr at once when MAX_TICKETS tickets are outstanding (*why = 1: only a
    // collect can help). With fewer tickets out a workspace is merely leased for the moment - a blocking caller, or
    // gmsm_trim walking the workspaces one at a time - and the submit waits for it like a blocking entry would (what
    // include/gmsm.h says of submit). Otherwise wait = false: nullptr when every workspace is leased (*why = 2);
    // wait = true: blocks until one is free - which always happens, because at least one workspace is never held by a
    // ticket. A waiter that wakes after gmsm_shutdown retired the context gives up (*why = 3).
    Workspace *acquire(bool wait, bool for_ticket = false, int *why = nullptr) {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t epoch0 = epoch;
        for (;;) {
            if (epoch != epoch0) {
                if (why) *why = 3;
                return nullptr;
            }
            if (for_ticket) {
                int out = 0;
                for (auto &w : ws) out += (w.busy && w.ticket) ? 1 : 0;
                if (out >= MAX_TICKETS) {
                    if (why) *why = 1;
                    return nullptr;
                }
            }
            for (auto &w : ws)
                if (!w.busy) {
                    if (w.ensure_base() != GMSM_OK) {  // the caller's device is current; the error text is set
                        if (why) *why = 4;
                        return nullptr;
                    }
                    w.busy = true;
                    w.ticket = for_ticket;
                    return &w;
                }
            if (!wait && !for_ticket) {
                if (why) *why = 2;
                return nullptr;
            }
            cv.wait(lk);
        }
    }
    // workspace i, waiting for a blocking caller to finish with it; nullptr when a ticket holds it (gmsm_shutdown)
    Workspace *acquire_unless_ticket(int i) {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t epoch0 = epoch;
        for (;;) {
            if (epoch != epoch0) return nullptr;  // a concurrent gmsm_shutdown got there first
            if (ws[i].busy && ws[i].ticket) return nullptr;
            if (!ws[i].busy) {
                ws[i].busy = true;
                ws[i].ticket = false;
                return &ws[i];
            }
            cv.wait(lk);
        }
    }
    Workspace *acquire_this(int i) {  // workspace i if it is free right now (gmsm_trim)
        std::lock_guard<std::mutex> lk(mu);
        if (ws[i].busy) return nullptr;
        ws[i].busy = true;
        ws[i].ticket = false;
        return &ws[i];
    }
    void release(Workspace *w) {
        {
            std::lock_guard<std::mutex> lk(mu);
            w->busy = false;
            w->ticket = false;
            w->pending = false;
        }
        cv.notify_all();
    }
};

struct Lease {
    Context &ctx;
    Workspace *w;
    // A workspace can still be busy with work that an enqueue-only call (gmsm_window_sums_enqueue) left in flight on the
    // caller's stream after giving its lease back: whoever leases it next orders the workspace's own stream behind that
    // work before touching any scratch buffer (entries that run on another stream do the same through begin_use).
    // A reference to registered bases that an enqueue-only call parked on the workspace is dropped here, outside the
    // context lock (the last owner's destructor synchronises the device before it frees the SRS).
    Lease(Context &c, bool wait = true, bool for_ticket = false, int *why = nullptr) : ctx(c), w(c.acquire(wait, for_ticket, why)) {
        if (!w) return;
        if (w->last_use && w->last_stream != w->stream) (void)hipStreamWaitEvent(w->stream, w->last_use, 0);
        std::shared_ptr<ResidentBases> parked;
        parked.swap(w->bases_ref);
    }
    ~Lease() {
        if (w) ctx.release(w);
    }
    Workspace *keep() {  // the lease outlives this object (submit -> collect)
        Workspace *x = w;
        w = nullptr;
        return x;
    }
    Lease(const Lease &) = delete;
    Lease &operator=(const Lease &) = delete;
};

#define GMSM_LEASE_OR_FAIL(name, context)                                                                    \
    Lease name(context);                                                                                     \
    if (!name.w) return fail(GMSM_ERR_DEVICE, "no workspace: gmsm_shutdown ran while this call waited, or its stream could not be created")

// Waits for a call's stream. hipStreamSynchronize parks the thread and wakes it when the stream has drained; with
// GMSM_OPT_SPIN_WAIT_US = t the stream is polled for up to t microseconds first. Measured (profiles/r04_spin_wait.log, same
// process, alternating): 2^16 0.548 -> 0.542 ms, 2^20 1.787 -> 1.781 ms per call - the runtime's own wait already spins
// briefly, so the gain is 5-6 us for a core kept busy for the whole call: OFF by default, there for callers who want it.
static inline hipError_t wait_stream(hipStream_t s) {
    const unsigned spin_us = options().spin_wait_us.load(std::memory_order_relaxed);
    if (spin_us) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t e = hipStreamQuery(s);
            if (e != hipErrorNotReady) return e;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    return hipStreamSynchronize(s);
}

// Orders the workspace's private stream after everything queued so far on the caller's stream (NULL = the device's
// default stream): inputs produced there are complete before the pipeline reads them.
static inline int order_after(Workspace &ws, hipStream_t caller) {
    if (!ws.dep) HIP_TRY(hipEventCreateWithFlags(&ws.dep, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ws.dep, caller));
    HIP_TRY(hipStreamWaitEvent(ws.stream, ws.dep, 0));
    return GMSM_OK;
}

// Scratch of a workspace is reused by every call that leases it; calls may run on different streams (the workspace's
// own, or the caller's for the enqueue-only entry). begin_use orders `stream` behind the last work enqueued on the
// workspace, end_use publishes the end of this call's work.
static inline int begin_use(Workspace &ws, hipStream_t stream) {
    if (ws.last_use && ws.last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, ws.last_use, 0));
    if (ws.conv_pending && ws.cstream) {  // a call failed between forking its base rewrite and joining it: the rewrite may still be writing ws.upoints
        HIP_TRY(hipStreamSynchronize(ws.cstream));
        ws.conv_pending = false;
    }
    return GMSM_OK;
}
static inline int end_use(Workspace &ws, hipStream_t stream) {
    if (!ws.last_use) HIP_TRY(hipEventCreateWithFlags(&ws.last_use, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ws.last_use, stream));
    ws.last_stream = stream;
    return GMSM_OK;
}

// Per-stage device timing (HIP events on the stream the kernels are launched on). Off by default; bench.py switches
// it on to obtain the dominant kernel's duration for the roofline record.
enum Stage {  // ABI index (gmsm_get_stage_times)
    STAGE_DECOMPOSE = 0,
    STAGE_HIST,
    STAGE_SCAN,
    STAGE_SCATTER,
    STAGE_ACCUMULATE,  // k_accumulate_seg alone: the dominant kernel of the roofline record
    STAGE_FIXUP,
    STAGE_REDUCE,
    STAGE_RESERVED,    // (round 2: wait of a window group for the previous one; always 0 now)
    STAGE_COUNT
};
enum TimedEvent { T_DECOMPOSE = 0, T_HIST, T_SCAN, T_SCATTER, T_ACCUMULATE, T_FIXUP, T_REDUCE, T_END };
static constexpr int T_TO_STAGE[T_END] = {STAGE_DECOMPOSE, STAGE_HIST, STAGE_SCAN, STAGE_SCATTER,
                                          STAGE_ACCUMULATE, STAGE_FIXUP, STAGE_REDUCE};

int profiling_level();  // 0 off, 1 every stage, 2 the accumulation kernel only (two events per pipeline run)
// adds one call's stage durations (and the number of kernel instances behind each) to the process-wide accumulators
void record_stage_times(const float *ms, const unsigned *launches);

struct StageTimer {
    Workspace &ws;
    int level;
    bool on;
    explicit StageTimer(Workspace &w, bool enabled = true) : ws(w), level(enabled ? profiling_level() : 0), on(level != 0) {
        if (on && !ws.events[0])
            for (int i = 0; i <= T_END; ++i) (void)hipEventCreate(&ws.events[i]);
        ws.timed_level = level;
    }
    void mark(int ev, hipStream_t stream) {
        if (level == 1 || (level == 2 && (ev == T_ACCUMULATE || ev == T_FIXUP))) (void)hipEventRecord(ws.events[ev], stream);
    }
    // adds the stage times of the run whose events ws holds to ms[] / launches[] (call after the stream has been synchronised)
    static void collect_add(Workspace &ws, float *ms, unsigned *launches) {
        for (int i = T_DECOMPOSE; i < T_END; ++i) {
            if (ws.timed_level == 2 && i != T_ACCUMULATE) continue;
            float t = 0.f;
            if (hipEventElapsedTime(&t, ws.events[i], ws.events[i + 1]) != hipSuccess) continue;
            ms[T_TO_STAGE[i]] += t;
            ++launches[T_TO_STAGE[i]];
        }
    }
    static void collect(Workspace &ws) {  // one pipeline run = one call of the profile
        float ms[STAGE_COUNT] = {0};
        unsigned launches[STAGE_COUNT] = {0};
        collect_add(ws, ms, launches);
        record_stage_times(ms, launches);
    }
};

int get_context(Context **out);  // context of the calling thread's device (gmsm_set_device), created on first use
int get_context_for(int device, Context **out);
// Context of the device that owns the device pointer `p` (the calling thread's device when p is null or unknown to
// the runtime): the device-pointer entries do not depend on which OS thread the caller happens to run on.
int get_context_of_pointer(const void *p, Context **out);


// ------------------------------------------------------------------ point ingest status (gmsm_ingest.h)
enum PointStatus : uint32_t {
    PT_OK = 0,
    PT_BAD_FLAG = 1,        // metadata bits of the other encoding, or undefined ones
    PT_BAD_INFINITY = 2,    // infinity flag with non-zero payload (ErrInvalidInfinityEncoding, marshal.go:36)
    PT_NOT_CANONICAL = 3,   // a coordinate >= q (SetBytesCanonical, fp/element.go)
    PT_NOT_ON_CURVE = 4,
    PT_NOT_IN_SUBGROUP = 5,
    PT_NO_SQRT = 6          // compressed X with no Y on the curve (marshal.go:921-923)
};

static inline const char *point_status_text(uint32_t s) {
    switch (s) {
        case PT_BAD_FLAG: return "invalid point encoding (flag bits this entry does not take: the raw entries read RawBytes() output, the compressed ones Bytes() output)";
        case PT_BAD_INFINITY: return "invalid infinity point encoding";
        case PT_NOT_CANONICAL: return "invalid fp.Element encoding (coordinate not below the modulus)";
        case PT_NOT_ON_CURVE: return "invalid point: not on the curve";
        case PT_NOT_IN_SUBGROUP: return "invalid point: subgroup check failed";
        case PT_NO_SQRT: return "invalid compressed coordinate: square root doesn't exist";
        default: return "ok";
    }
}

// ------------------------------------------------------------------ window geometry
static inline unsigned num_windows(unsigned fr_bits, unsigned c) { return (fr_bits + c - 1) / c; }  // multiexp.go:681
static inline unsigned last_c(unsigned fr_bits, unsigned c) {                                         // multiexp.go:690
    unsigned avail = num_windows(fr_bits, c) * c - fr_bits;
    return c + 1 - avail;
}

// Host threads this process may keep busy: the hardware thread count capped by the cgroup v2 CPU quota (the GPU boxes
// expose 256 hardware threads to a container that may use 16 CPUs).
static inline unsigned usable_cpus() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        unsigned long period = 0;
        if (fscanf(f, "%31s %lu", quota, &period) == 2 && period > 0 && quota[0] != 'm') {
            const unsigned long q = strtoul(quota, nullptr, 10);
            if (q > 0) n = std::min<unsigned>(n, (unsigned)std::max<unsigned long>(1, q / period));
        }
        fclose(f);
    }
    return n;
}

// Run-time switches of the shipped library: `Options` above (gmsm_set_option) and the device list (gmsm_set_devices,
// GMSM_DEVICES). env_uint is what the library's one-time initialisation and the -DGMSM_EXPERIMENTS builds read the
// environment with; everything that was a knob while the engine was being tuned is a compile-time constant unless the
// library is built with -DGMSM_EXPERIMENTS (A/B builds, tools/build_ab.sh): tune_uint() then reads GMSM_<NAME>.
static inline unsigned env_uint(const char *name, unsigned dflt) {
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    return (unsigned)strtoul(v, nullptr, 10);
}
#ifdef GMSM_EXPERIMENTS
static inline unsigned tune_uint(const char *name, unsigned dflt) { return env_uint(name, dflt); }
#else
static inline constexpr unsigned tune_uint(const char *, unsigned dflt) { return dflt; }
#endif

// Window width. The affine result does not depend on it (the reference asserts exactly that for c in 2..16,
// multiexp_test.go:95-126), so it is purely a cost choice - and the cost depends on the element type: a bucket of a
// wide type costs far more to reduce than to fill, so those groups want fewer buckets than BN254 G1 does.
// Measured on MI355X for every group and size 2^10..2^21/2^26 (profiles/r02_window_sweeps.log), indexed by
// floor(log2 n):
//   BN254 G1       < 2^13: 8   < 2^15: 13   < 2^17: 15   < 2^21: 16   from 2^21: 17 (15 windows of 2^16 buckets: one
//                  accumulation pass less for 0.15 ms more reduction: 2^21 3.68 -> 3.58 ms, 2^22 6.72 -> 6.52, 2^24
//                  24.1 -> 22.9, 2^26 97.1 -> 91.2; c = 20 gains less - 23.7 at 2^24 - because 13 x 2^19 buckets cost
//                  2.1 ms to reduce)
//   BN254 G2       < 2^15: 8   2^15: 13   2^16: 15   < 2^22: 16   from 2^22: 17 (2^22: 20.2 -> 19.7 ms)
//   BLS12-381 G1   < 2^15: 8   2^15: 13   < 2^24: 16   from 2^24: 17 (2^24: 45.2 -> 43.2 ms; at 2^22 17 is slower, 13.1
//                  against 12.5; c = 15 gives 17 full windows and an 18th for the carry: always worse)
//   BLS12-381 G2   < 2^15: 8   2^15: 10   2^16: 12   2^17, 2^18: 13   then 16   (2^16: 4.93 against 7.27 ms with c = 15)
//   BW6-761 G1/G2  < 2^17: 9   2^17..2^20: 14   then 16   (2^13: 4.1 against 5.6 ms, 2^18: 9.7 against 13.0, 2^20:
//                  23.3 against 25.3; c = 10, 15 and 17 leave a top window of a few bits and are far slower)
// Below ~2^17 points the pipeline is latency-bound (~0.6-1 ms for BN254 G1 whatever c). The entries were measured with
// the width forced, so they include what a narrow top window costs (long chains of partial sums for k_fixup_long, one
// crowded sort partition): widths whose top window holds only a few bits simply never won. GMSM_OPT_WINDOW_BITS (GMSM_C) overrides.
static inline unsigned preferred_c(unsigned fr_bits, size_t aff_bytes, size_t n) {
    unsigned lg = 0;
    while (lg < 63 && ((size_t)2 << lg) <= n) ++lg;  // floor(log2 n), n >= 1
    // Round 6: the band 2^13..2^18 re-measured with every width forced (tools/glv_width_sweep.py, profiles/r06_glv_width_sweep.log;
    // its glv0 columns): BN254 G2 2^16 1.49 -> 1.14 ms (12 instead of 15), 2^17 1.55 -> 1.36 (13), BW6-761 2^16 3.14 -> 2.59 (12),
    // BLS12-381 G1 2^16 0.96 -> 0.86 (12), 2^18 1.41 -> 1.36 (13), BN254 G1 2^13 0.50 -> 0.41 (12).
    if (fr_bits > 320) return lg < 13 ? 9u : lg <= 14 ? 10u : lg <= 16 ? 12u : lg <= 20 ? 14u : 16u;   // BW6-761
    if (fr_bits == 255 && aff_bytes > 96) return lg < 13 ? 8u : lg <= 15 ? 10u : lg == 16 ? 12u : lg <= 18 ? 13u : 16u;  // BLS12-381 G2
    if (fr_bits == 255) return lg < 13 ? 8u : lg <= 14 ? 12u : lg == 15 ? 13u : lg == 16 ? 12u : lg == 18 ? 13u : lg < 24 ? 16u : 17u;  // BLS12-381 G1
    if (aff_bytes > 64) return lg < 13 ? 8u : lg <= 14 ? 12u : lg == 15 ? 13u : lg == 16 ? 12u : lg <= 18 ? 13u : lg < 22 ? 16u : 17u;  // BN254 G2
    return lg < 13 ? 8u : lg < 15 ? 12u : lg < 17 ? 15u : lg < 21 ? 16u : 17u;                       // BN254 G1
}
// GLV half scalars in the sorted pipeline (gmsm_glv.h; bases taken anew): the window width of the 2 n-entry call for the sizes
// where it was measured AHEAD of the best plain width, 0 elsewhere (same sweep, glv2 against glv0 columns, resident ms):
//   BN254 G1      2^13 0.41 -> 0.33, 2^15 0.49 -> 0.44, 2^16 0.53 -> 0.48, 2^17 0.62 -> 0.54, 2^18 0.78 -> 0.71, 2^19 1.11 -> 1.06,
//                 2^20 1.776 -> 1.725; 2^21 3.22 -> 3.28 and beyond: slower (2^22 + 8 %, 2^24 + 13 %: the gather works on twice the
//                 bases, profiles/r06_glv_large.log)
//   BN254 G2      2^13 0.79 -> 0.62, 2^14 0.85 -> 0.69, 2^16 1.14 -> 0.98, 2^17 1.36 -> 1.18, 2^18 1.92 -> 1.69, 2^19 2.89 -> 2.62,
//                 2^20 4.77 -> 4.51, 2^21 8.45 -> 8.38
//   BLS12-381 G1  2^13 0.58 -> 0.52, 2^14 0.64 -> 0.58, 2^16 0.85 -> 0.80, 2^19 2.15 -> 2.07; level or behind elsewhere
//   BLS12-381 G2  2^13 1.99 -> 1.69, 2^14 2.00 -> 1.76, 2^15 1.94 -> 1.81, 2^16 2.17 -> 1.92, 2^17 2.60 -> 2.29, 2^18 4.02 -> 3.65,
//                 2^19 6.49 -> 6.29; 2^20 10.56 -> 10.71: behind
//   BW6-761       2^13 2.20 -> 1.99, 2^14 2.27 -> 2.10, 2^15 2.45 -> 2.24, 2^16 2.59 -> 2.39, 2^17 3.94 -> 3.61, 2^18 6.17 -> 5.82,
//                 2^19 10.4 -> 9.9, 2^20 18.4 -> 17.4, 2^21 33.6 -> 32.0
// What it buys is half the bucket sets to reduce and half the host fold; what it costs is a gather over twice as many bases.
static inline unsigned glv_preferred_c(unsigned fr_bits, size_t aff_bytes, size_t n) {
    unsigned lg = 0;
    while (lg < 63 && ((size_t)2 << lg) <= n) ++lg;
    if (lg < 11 || lg > 21) return 0u;
    if (fr_bits > 320) return lg <= 14 ? 10u : lg <= 16 ? 12u : lg == 17 ? 13u : lg == 18 ? 14u : 16u;      // BW6-761
    if (fr_bits == 255 && aff_bytes > 96)                                                                    // BLS12-381 G2
        return lg <= 14 ? 10u : lg <= 16 ? 12u : lg <= 18 ? 13u : lg == 19 ? 15u : 0u;
    if (lg < 13) return 0u;  // the narrow types: the fused kernel's sizes
    if (fr_bits == 255) return lg <= 14 ? 13u : lg == 16 ? 12u : lg == 19 ? 16u : 0u;                       // BLS12-381 G1
    if (aff_bytes > 64) return lg == 13 ? 12u : lg == 14 ? 13u : lg <= 16 ? 12u : lg == 17 ? 13u : 16u;     // BN254 G2
    return lg == 13 ? 12u : lg == 14 ? 13u : lg <= 20 ? 16u : 0u;                                            // BN254 G1
}
static inline unsigned choose_c(unsigned fr_bits, size_t aff_bytes, size_t n) {
    const unsigned forced = options().window_bits.load(std::memory_order_relaxed);
    if (forced >= 2 && forced <= 20) return forced;  // the range gmsm.h documents (gmsm_window_sums_*)
    return preferred_c(fr_bits, aff_bytes, n ? n : 1);
}

struct GroupVTable {
    unsigned fr_bits;
    size_t aff_bytes, scalar_bytes, jac_bytes, xyzz_bytes;
    int (*multiexp_host)(const uint64_t *points, size_t n_points, const uint64_t *scalars, size_t n_scalars, int nb_tasks,
                         uint64_t *out_jac);
    int (*multiexp_device)(Context &ctx, const void *d_points, const void *d_scalars, size_t n, hipStream_t stream,
                           uint64_t *out_jac, const ResidentBases *resident);
    int (*window_sums)(Context &ctx, const void *d_points, const void *d_scalars, size_t n, unsigned c, unsigned win_first,
                       unsigned win_stride, hipStream_t stream, uint64_t *out_xyzz, const ResidentBases *resident);
    void (*fold)(const uint64_t *xyzz_windows, unsigned c, uint64_t *out_jac);
    void (*jac_to_affine)(const uint64_t *jac, uint64_t *out_affine);
    int (*debug_decompose)(const uint64_t *scalars, size_t n, unsigned c, uint32_t *out_digits);
    int (*debug_field_op)(int field, int op, const uint64_t *a, const uint64_t *b, size_t count, uint64_t *out);
    int (*debug_group_op)(int op, const uint64_t *acc, const uint64_t *other, size_t count, uint64_t *out);
    void (*generate_points)(const uint64_t *base, const uint64_t *k0, const uint64_t *k1, int klimbs, size_t n, int nthreads,
                            uint64_t *out);
    int (*register_bases)(Context &ctx, const void *d_points, size_t n, hipStream_t stream, ResidentBases *out);
    int (*submit)(Context &ctx, Workspace &ws, const void *d_scalars, size_t n, const ResidentBases *resident);
    int (*collect)(Workspace &ws, uint64_t *out_jac);
    int (*window_sums_enqueue)(Context &ctx, const void *d_points, const void *d_scalars, size_t n, unsigned c,
                               unsigned win_first, unsigned win_stride, hipStream_t stream, void *d_out_xyzz,
                               const std::shared_ptr<ResidentBases> &resident);
    void (*fold_sets)(const uint64_t *xyzz_sets, unsigned nsets, unsigned c, uint64_t *out_jac);
    void (*fold_powers)(const uint64_t *coeff, size_t n, uint64_t *out_scalars);  // 1, g, g^2, ... (Fold, multiexp.go:331)
    int (*multiexp_bases_host)(Context &ctx, const uint64_t *scalars, size_t n, uint64_t *out_jac,
                               const ResidentBases *resident);
    int (*batch_scalar_mul)(Context &ctx, const uint64_t *base, const uint64_t *scalars, const void *d_scalars, size_t n,
                            hipStream_t caller_stream, uint64_t *out, void *d_out);
    int (*batch_jac_to_affine)(Context &ctx, const uint64_t *jac, size_t n, uint64_t *out);
    int (*decode_raw)(Workspace &ws, const void *d_raw, size_t n, int level, void *d_out, long long *bad_index,
                      uint32_t *status);
    int (*validate_points)(Workspace &ws, const void *d_points, size_t n, int level, long long *bad_index, uint32_t *status);
    // the Encoder's default (compressed) format: gmsm_decompress.h
    int (*decode_compressed)(Workspace &ws, const void *d_comp, size_t n, int level, void *d_out, long long *bad_index,
                             uint32_t *status);
    int (*encode_compressed)(Workspace &ws, const void *d_points, size_t n, void *d_comp);
    // fr/fft over the group's scalar field (gmsm_fft.h)
    int (*fft_domain_new)(Context &ctx, hipStream_t stream, unsigned log2n, FftDomain *out);
    int (*fft_run)(hipStream_t stream, FftDomain *d, void *d_a, bool inverse, bool dif, bool coset);
    int (*fft_bit_reverse)(hipStream_t stream, void *d_a, size_t n);
    // one rank's piece of a MultiExp that the library shards over several devices (Group::shard_piece)
    int (*precompute_tables)(Context &ctx, Workspace &ws, ResidentBases *rb, unsigned c);  // window tables of registered bases
    bool (*tables_serve)(size_t n_registered, size_t n_call);  // call sizes that run through the tables (Group::use_tables)
    int (*shard_piece)(Context &ctx, const uint64_t *points, const ResidentBases *resident, size_t resident_base,
                       const uint64_t *scalars, size_t n, unsigned c, unsigned win_first, unsigned win_stride,
                       uint64_t *out_xyzz);
    unsigned (*host_piece_ranges)(size_t n, bool with_points);  // point ranges a host-buffer piece of n points runs as
    int (*debug_glv_split)(const uint64_t *scalars, size_t n, uint32_t *out);  // test hook of gmsm_glv.h
    void (*plan_info)(size_t n, unsigned *c, unsigned *nwin, unsigned *entries_per_point, unsigned *fused);  // gmsm_default_plan
};

}  // namespace gmsm
