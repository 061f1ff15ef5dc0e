// libgmsm.so -- C ABI (include/gmsm.h), per-device contexts and dispatch to the per-group pipelines (gmsm_group.h).
// There is no CPU fallback: every compute entry needs a usable gfx950 device and fails loudly otherwise.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <sys/stat.h>

#include "gmsm_context.h"

namespace gmsm {

// Error text: per thread (like errno), plus the most recent one of the process - a cgo caller's goroutine may be moved to
// another OS thread between the failing call and gmsm_last_error().
static thread_local std::string g_last_error;
static std::mutex g_any_error_mu;
static std::string g_any_error;
// Device selection: gmsm_set_device sets the calling thread's device AND the process-wide default that threads which
// never called it inherit (a goroutine that selected a device and was then moved to a fresh OS thread still gets it).
// Entries that receive device pointers use the device that owns the pointer instead.
static thread_local int g_device = -1;
static std::atomic<int> g_default_device{0};
std::atomic<unsigned long> g_table_runs{0};
std::atomic<unsigned long> g_small_runs{0};
static std::atomic<uint32_t> g_ticket_gen{0};  // ticket generations: process-wide and monotonic, so a ticket of before a gmsm_shutdown never matches one issued after it

// Process-wide switches; GMSM_C and GMSM_TABLES are read here ONCE (first use), never on a call path.
std::string g_env_error;  // a malformed GMSM_C / GMSM_TABLES: reported by the next drop-in entry (shard_devices), not dropped
Options &options() {
    static Options *o = [] {
        Options *x = new Options();
        const unsigned c = env_uint("GMSM_C", 0);
        x->window_bits.store(c >= 2 && c <= 20 ? c : 0);
        const unsigned t = env_uint("GMSM_TABLES", 1);
        x->tables.store(std::min(2u, t));
        // an out-of-range initial value is not dropped silently: the next MultiExp entry reports it (g_env_error)
        if (c != 0 && (c < 2 || c > 20)) g_env_error = "GMSM_C=" + std::to_string(c) + ": the window width must be 2..20 (or unset)";
        else if (t > 2) g_env_error = "GMSM_TABLES=" + std::to_string(t) + ": 0 never, 1 the measured call sizes, 2 every call size";
        return x;
    }();
    return *o;
}

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    {
        std::lock_guard<std::mutex> lk(g_any_error_mu);
        g_any_error = msg;
    }
    return code;
}

void clear_last_error() { g_last_error.clear(); }

static std::mutex g_prof_mu;
static std::atomic<int> g_profiling{0};
static double g_stage_ms[STAGE_COUNT] = {0};
static unsigned long g_stage_launches[STAGE_COUNT] = {0};
static unsigned long g_stage_calls = 0;

int profiling_level() { return g_profiling.load(std::memory_order_relaxed); }
void record_stage_times(const float *ms, const unsigned *launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < STAGE_COUNT; ++i) {
        g_stage_ms[i] += ms[i];
        g_stage_launches[i] += launches[i];
    }
    ++g_stage_calls;
}

static std::mutex g_ctx_mu;
static std::vector<Context *> g_ctx;

int get_context_for(int device, Context **out) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(GMSM_ERR_DEVICE, std::string("no usable HIP device (hipGetDeviceCount: ") + hipGetErrorString(e) +
                                         "); libgmsm has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(GMSM_ERR_DEVICE, "gmsm_set_device: device index out of range");
    if ((int)g_ctx.size() < ndev) g_ctx.resize(ndev, nullptr);
    if (!g_ctx[device]) g_ctx[device] = new Context();  // never deleted: gmsm_shutdown retires it (Context::retire)
    if (!g_ctx[device]->live) {  // first use, or first use after a gmsm_shutdown
        int rc = g_ctx[device]->init(device);
        if (rc != GMSM_OK) return rc;
    }
    *out = g_ctx[device];
    return GMSM_OK;
}

int get_context(Context **out) { return get_context_for(g_device >= 0 ? g_device : g_default_device.load(), out); }

int get_context_of_pointer(const void *p, Context **out) {
    if (p) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeDevice)
            return get_context_for(attr.device, out);
        (void)hipGetLastError();  // not a pointer the runtime knows: fall back to the thread's device
    }
    return get_context(out);
}

const GroupVTable *gmsm_vtable_bn254_g1();
const GroupVTable *gmsm_vtable_bn254_g2();
const GroupVTable *gmsm_vtable_bls12_381_g1();
const GroupVTable *gmsm_vtable_bls12_381_g2();
const GroupVTable *gmsm_vtable_bw6_761_g1();
const GroupVTable *gmsm_vtable_bw6_761_g2();

static const GroupVTable *vtable(int group) {
    switch (group) {
        case GMSM_BN254_G1: return gmsm_vtable_bn254_g1();
        case GMSM_BN254_G2: return gmsm_vtable_bn254_g2();
        case GMSM_BLS12_381_G1: return gmsm_vtable_bls12_381_g1();
        case GMSM_BLS12_381_G2: return gmsm_vtable_bls12_381_g2();
        case GMSM_BW6_761_G1: return gmsm_vtable_bw6_761_g1();
        case GMSM_BW6_761_G2: return gmsm_vtable_bw6_761_g2();
        default: return nullptr;
    }
}

}  // namespace gmsm

using namespace gmsm;

#define GMSM_EXPORT __attribute__((visibility("default")))
#define VT_OR_FAIL(group)                          \
    const GroupVTable *vt = vtable(group);         \
    if (!vt) return fail(GMSM_ERR_ARG, "unknown group id")

// ------------------------------------------------------------------ one MultiExp over several devices
// The reference spreads one MultiExp over the cores of a machine: _innerMsmG1 starts a worker per c-bit window and
// collects one g1JacExtended per window on a channel (ecc/bn254/multiexp.go:148-209), and above a size threshold the call
// is first split in two halves of the points whose results are added (multiexp.go:98-140, AddAssign). Here the workers are
// the GPUs of the node and the split happens inside the library, below the C ABI, so that a Go caller of
// gmsm_<curve>_g1_multiexp gets all of them without knowing: one host thread per logical rank (a persistent pool, two
// workers per device like the two workspaces), each running its piece on its device through Group::shard_piece;
//   points   rank r owns points [r n/G, (r+1) n/G) and computes every window of its slice; window w of the whole MultiExp is
//            the sum of the ranks' totals for w. Each device pulls only its slice over its own PCIe link; default.
//   windows  rank r owns windows r, r+G, ... over all points (the reference's per-window workers); every device needs
//            all bases, so this only pays with bases registered on every device and few points.
// The exchange is the ranks' window totals - at most 64 extended-Jacobian points per rank, a few KB - copied from each
// device's pinned result buffer by its own host thread; an RCCL all-gather would move the same bytes device-to-device
// first and then still have to cross to the host for the fold, so the single-process form has no collective. (The
// one-process-per-GPU form of the same decomposition, with one RCCL all-gather over xGMI, is gnark-crypto_amd/sharding.py.)
// Then ONE fold (msmReduceChunk, multiexp.go:302-315) on the calling thread.
namespace gmsm {

struct ShardPool {
    struct Dev {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<std::function<void()>> q;
        int workers = 0;
    };
    std::mutex mu;
    std::map<int, Dev *> devs;  // never freed: the workers outlive every static destructor
    void post(int device, std::function<void()> fn) {
        Dev *d;
        {
            std::lock_guard<std::mutex> lk(mu);
            Dev *&slot = devs[device];
            if (!slot) slot = new Dev();
            d = slot;
        }
        {
            std::lock_guard<std::mutex> lk(d->mu);
            // two workers per device; a thread that cannot be created (std::system_error) leaves through the caller's
            // handler with nothing queued - and nothing is ever queued for a device that has no worker at all
            for (; d->workers < 2; ++d->workers) {
                try {
                    std::thread([d] {
                        for (;;) {
                            std::function<void()> job;
                            {
                                std::unique_lock<std::mutex> lk2(d->mu);
                                d->cv.wait(lk2, [d] { return !d->q.empty(); });
                                job = std::move(d->q.front());
                                d->q.pop_front();
                            }
                            job();
                        }
                    }).detach();
                } catch (...) {
                    if (d->workers == 0) throw;
                    break;  // one worker serves the device
                }
            }
            d->q.push_back(std::move(fn));
        }
        d->cv.notify_one();
    }
};
static ShardPool &shard_pool() {
    static ShardPool *p = new ShardPool();
    return *p;
}

// Devices the drop-in entries spread a MultiExp over, one entry per logical rank (a device may appear more than once).
// OPT-IN (round 4; the advisor's finding): a process that configured nothing runs every drop-in call on ONE device - the
// calling thread's (gmsm_set_device) or device 0 - and creates no context, stream or worker thread anywhere else: the
// usual deployment is one prover process per GPU, and a library that silently took all eight would compete with its
// neighbours. Spreading is asked for with gmsm_set_devices(list) or GMSM_DEVICES ("0,1,2,3", or "all"), read once.
// Resolution: gmsm_set_devices; else a gmsm_set_device call pins the process to that one device; else GMSM_DEVICES;
// else no spreading. The explicit entries (gmsm_multiexp_sharded, gmsm_bases_register_sharded) with devices = NULL use the
// configured list or, when there is none, every visible device - their caller asked for several devices by name.
static std::mutex g_devices_mu;
static std::vector<int> g_devices;
static bool g_devices_explicit = false;
static std::atomic<bool> g_pinned{false};  // gmsm_set_device was called

// GMSM_DEVICES, parsed once. A malformed or out-of-range entry is an error the next drop-in call reports (it does not
// silently shrink the list): err is non-empty then.
struct EnvDevices {
    std::vector<int> list;
    std::string err;
};
static const EnvDevices &env_devices() {
    static const EnvDevices *e = [] {
        EnvDevices *x = new EnvDevices();
        const char *v = getenv("GMSM_DEVICES");
        if (!v || !*v) return x;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
            x->err = std::string("GMSM_DEVICES=\"") + v + "\": the device count could not be obtained (hipGetDeviceCount failed)";
            return x;
        }
        if (strcmp(v, "all") == 0) {
            for (int d = 0; d < ndev; ++d) x->list.push_back(d);
            return x;
        }
        for (const char *p = v; *p;) {
            char *end = nullptr;
            const long d = strtol(p, &end, 10);
            if (end == p || (*end && *end != ',') || (*end == ',' && !end[1])) {
                x->err = std::string("GMSM_DEVICES=\"") + v + "\": expected a comma-separated list of device indices or \"all\"";
                break;
            }
            if (d < 0 || d >= ndev) {
                x->err = std::string("GMSM_DEVICES=\"") + v + "\": device " + std::to_string(d) + " does not exist (" +
                         std::to_string(ndev) + " visible)";
                break;
            }
            x->list.push_back((int)d);
            p = *end ? end + 1 : end;
        }
        if (!x->err.empty()) x->list.clear();
        return x;
    }();
    return *e;
}

// The configured list; empty = nothing configured (or pinned). rc != 0: GMSM_DEVICES is malformed.
static int shard_devices(std::vector<int> &out) {
    out.clear();
    {
        std::lock_guard<std::mutex> lk(g_devices_mu);
        if (g_devices_explicit) {
            out = g_devices;
            return GMSM_OK;
        }
    }
    (void)options();
    if (!g_env_error.empty()) return fail(GMSM_ERR_ARG, g_env_error);
    if (g_pinned.load()) return GMSM_OK;
    const EnvDevices &e = env_devices();
    if (!e.err.empty()) return fail(GMSM_ERR_ARG, e.err);
    out = e.list;
    return GMSM_OK;
}

constexpr size_t SHARD_MIN_SLICE = (size_t)1 << 16;  // fewer points per rank than this are not worth a second device

// mode: 0 auto, 1 points, 2 windows. `replicas[d]` = bases registered on device d (full copies), or empty with `points`.
static int multiexp_sharded_run(int group, const uint64_t *points, const std::map<int, std::shared_ptr<ResidentBases>> *replicas,
                                const uint64_t *scalars, size_t n, const std::vector<int> &devs, int mode, uint64_t *out_jac) {
    const GroupVTable *vt = vtable(group);
    size_t G = devs.size();
    if (G == 0) return fail(GMSM_ERR_ARG, "sharded MultiExp: empty device list");
    if (mode == 0) mode = 1;
    if (mode != 1 && mode != 2) return fail(GMSM_ERR_ARG, "sharded MultiExp: mode must be 0 (auto), 1 (points) or 2 (windows)");
    unsigned c;
    uint32_t nwin;
    if (mode == 1) {
        G = std::max<size_t>(1, std::min(G, n / SHARD_MIN_SLICE));
        const size_t slice = (n + G - 1) / G;  // the largest slice decides c: the ranks' totals must line up
        c = choose_c(vt->fr_bits, vt->aff_bytes, slice);
        if (replicas && !replicas->empty()) {  // window tables on the replicas (gmsm_bases_precompute): their width
            const ResidentBases *rb0 = replicas->begin()->second.get();
            const unsigned forced = options().window_bits.load();
            bool all = rb0->tab_c.load() != 0;
            for (const auto &kv : *replicas) all = all && kv.second->tab_c.load() == rb0->tab_c.load();
            if (all && !(forced >= 2 && forced <= 20 && forced != rb0->tab_c.load()) && options().tables.load() != 0 &&
                vt->tables_serve(rb0->n, n / G) && vt->tables_serve(rb0->n, slice))
                c = rb0->tab_c.load();
        }
        nwin = num_windows(vt->fr_bits, c);
    } else {
        c = choose_c(vt->fr_bits, vt->aff_bytes, n);
        nwin = num_windows(vt->fr_bits, c);
        G = std::min<size_t>(G, nwin);
    }
    const size_t xl = vt->xyzz_bytes / 8;
    const uint32_t rows = mode == 1 ? nwin : (uint32_t)((nwin + G - 1) / G);
    std::vector<uint64_t> sets(G * rows * xl, 0);
    struct Result {
        int rc = GMSM_OK;
        std::string err;
    };
    std::vector<Result> res(G);
    std::mutex done_mu;
    std::condition_variable done_cv;
    size_t done = 0;
    auto piece_body = [&](size_t r) {
        Result &out = res[r];
        const int dev = devs[r];
        Context *ctx = nullptr;
        out.rc = get_context_for(dev, &ctx);
        if (out.rc == GMSM_OK && hipSetDevice(dev) != hipSuccess) out.rc = fail(GMSM_ERR_DEVICE, "hipSetDevice failed");
        if (out.rc == GMSM_OK) {  // the piece must run where its context lives: say so loudly instead of computing on a neighbour
            int cur = -1;
            if (hipGetDevice(&cur) != hipSuccess || cur != dev || ctx->device != dev)
                out.rc = fail(GMSM_ERR_DEVICE, "sharded MultiExp: rank " + std::to_string(r) + " expected device " + std::to_string(dev) +
                                                   ", the worker thread is on device " + std::to_string(cur) + " (context of device " +
                                                   std::to_string(ctx->device) + ")");
        }
        if (out.rc == GMSM_OK) {
            const ResidentBases *rb = nullptr;
            if (replicas) {
                auto it = replicas->find(dev);
                rb = it == replicas->end() ? nullptr : it->second.get();
                if (!rb) out.rc = fail(GMSM_ERR_ARG, "sharded MultiExp: the bases are not registered on device " + std::to_string(dev));
            }
            if (out.rc == GMSM_OK) {
                const size_t lo = mode == 1 ? r * n / G : 0, hi = mode == 1 ? (r + 1) * n / G : n;
                out.rc = vt->shard_piece(*ctx, points ? points + lo * (vt->aff_bytes / 8) : nullptr, rb, lo,
                                         scalars + lo * (vt->scalar_bytes / 8), hi - lo, c, mode == 1 ? 0u : (unsigned)r,
                                         mode == 1 ? 1u : (unsigned)G, sets.data() + r * rows * xl);
            }
        }
        if (out.rc != GMSM_OK) out.err = gmsm_last_error();
    };
    // Nothing may leave a piece but a return code: the workers write into this frame (sets, res) and a C caller is above us.
    auto piece = [&](size_t r) noexcept {
        try {
            piece_body(r);
        } catch (const std::exception &e) {
            res[r].rc = GMSM_ERR_DEVICE;
            res[r].err = std::string("exception in a shard piece: ") + e.what();
        } catch (...) {
            res[r].rc = GMSM_ERR_DEVICE;
            res[r].err = "unknown exception in a shard piece";
        }
    };
    size_t posted = 0;
    struct WaitAll {  // the frame outlives every worker that was handed a reference to it, whatever happens below
        std::mutex &mu;
        std::condition_variable &cv;
        size_t &done;
        const size_t &posted;
        ~WaitAll() {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done == posted; });
        }
    };
    int prev_dev = 0;
    (void)hipGetDevice(&prev_dev);
    {
        WaitAll wait_all{done_mu, done_cv, done, posted};
        try {
            for (size_t r = 1; r < G; ++r) {
                shard_pool().post(devs[r], [&, r] {
                    piece(r);
                    std::lock_guard<std::mutex> lk(done_mu);  // the notify stays under the lock: the waiter owns these objects
                    ++done;
                    done_cv.notify_one();
                });
                ++posted;
            }
        } catch (const std::exception &e) {  // std::bad_alloc / std::system_error out of the pool: the ranks not posted fail
            for (size_t r = posted + 1; r < G; ++r) {
                res[r].rc = GMSM_ERR_DEVICE;
                res[r].err = std::string("cannot start the worker: ") + e.what();
            }
        }
        piece(0);
    }
    (void)hipSetDevice(prev_dev);
    for (size_t r = 0; r < G; ++r)
        if (res[r].rc != GMSM_OK) return fail(res[r].rc, "rank " + std::to_string(r) + " (device " + std::to_string(devs[r]) + "): " + res[r].err);
    if (mode == 1) {
        vt->fold_sets(sets.data(), (unsigned)G, c, out_jac);
    } else {
        std::vector<uint64_t> totals((size_t)nwin * xl);
        for (uint32_t w = 0; w < nwin; ++w)
            memcpy(totals.data() + (size_t)w * xl, sets.data() + ((w % G) * rows + w / G) * xl, vt->xyzz_bytes);
        vt->fold(totals.data(), c, out_jac);
    }
    return GMSM_OK;
}

// argument checks of (*G1Jac).MultiExp (multiexp.go:61-71) + the empty input, shared by the sharded entries
static int multiexp_precheck(const GroupVTable *vt, size_t n_points, size_t n_scalars, int nb_tasks, uint64_t *out_jac, bool *done) {
    *done = true;
    if (n_points != n_scalars) return fail(GMSM_ERR_LEN, "len(points) != len(scalars)");
    if (nb_tasks > 1024) return fail(GMSM_ERR_CONFIG, "invalid config: config.NbTasks > 1024");
    if (n_points == 0) return vt->multiexp_host(nullptr, 0, nullptr, 0, nb_tasks, out_jac);  // infinity (Z = 0); still needs a device
    *done = false;
    return GMSM_OK;
}

}  // namespace gmsm

extern "C" {

GMSM_EXPORT int gmsm_multiexp_sharded(int group, const uint64_t *points, size_t n_points, const uint64_t *scalars,
                                      size_t n_scalars, int nb_tasks, const int *devices, int n_devices, int mode,
                                      uint64_t *out_jac) {
    VT_OR_FAIL(group);
    bool done;
    int rc = multiexp_precheck(vt, n_points, n_scalars, nb_tasks, out_jac, &done);
    if (rc || done) return rc;
    std::vector<int> devs;
    if (devices && n_devices > 0) devs.assign(devices, devices + n_devices);
    else if ((rc = shard_devices(devs))) return rc;
    if (devs.empty()) {  // nothing configured: the caller asked for a sharded call, so every visible device
        int ndev = gmsm_device_count();
        if (ndev <= 0) return fail(GMSM_ERR_DEVICE, "no usable HIP device; libgmsm has no CPU fallback");
        for (int d = 0; d < ndev; ++d) devs.push_back(d);
    }
    const int ndev = gmsm_device_count();
    for (int d : devs)
        if (d < 0 || d >= ndev)
            return fail(GMSM_ERR_DEVICE, "sharded MultiExp: device " + std::to_string(d) + " does not exist (" + std::to_string(ndev) + " visible)");
    return multiexp_sharded_run(group, points, nullptr, scalars, n_points, devs, mode, out_jac);
}

GMSM_EXPORT int gmsm_multiexp(int group, const uint64_t *points, size_t n_points, const uint64_t *scalars,
                              size_t n_scalars, int nb_tasks, uint64_t *out_jac) {
    VT_OR_FAIL(group);
    // more than one device CONFIGURED (gmsm_set_devices / GMSM_DEVICES - spreading is opt-in) and enough points for two
    // slices: the call is spread over them
    if (n_points == n_scalars && nb_tasks <= 1024 && n_points >= 2 * SHARD_MIN_SLICE) {
        std::vector<int> devs;
        int rc = shard_devices(devs);
        if (rc) return rc;
        if (devs.size() > 1) return multiexp_sharded_run(group, points, nullptr, scalars, n_points, devs, 0, out_jac);
    }
    return vt->multiexp_host(points, n_points, scalars, n_scalars, nb_tasks, out_jac);
}

#define GMSM_DROPIN(name, id)                                                                                \
    GMSM_EXPORT int gmsm_##name##_multiexp(const uint64_t *points, size_t n_points, const uint64_t *scalars, \
                                           size_t n_scalars, int nb_tasks, uint64_t *out_jac) {              \
        return gmsm_multiexp(id, points, n_points, scalars, n_scalars, nb_tasks, out_jac);                   \
    }
GMSM_DROPIN(bn254_g1, GMSM_BN254_G1)
GMSM_DROPIN(bn254_g2, GMSM_BN254_G2)
GMSM_DROPIN(bls12_381_g1, GMSM_BLS12_381_G1)
GMSM_DROPIN(bls12_381_g2, GMSM_BLS12_381_G2)
GMSM_DROPIN(bw6_761_g1, GMSM_BW6_761_G1)
GMSM_DROPIN(bw6_761_g2, GMSM_BW6_761_G2)

GMSM_EXPORT int gmsm_multiexp_affine(int group, const uint64_t *points, size_t n_points, const uint64_t *scalars,
                                     size_t n_scalars, int nb_tasks, uint64_t *out_affine) {
    VT_OR_FAIL(group);
    std::vector<uint64_t> jac(vt->jac_bytes / 8);
    int rc = gmsm_multiexp(group, points, n_points, scalars, n_scalars, nb_tasks, jac.data());
    if (rc) return rc;
    vt->jac_to_affine(jac.data(), out_affine);
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_fold(int group, const uint64_t *points, size_t n_points, const uint64_t *combination_coeff,
                          int nb_tasks, uint64_t *out_jac) {
    VT_OR_FAIL(group);
    if (!combination_coeff) return fail(GMSM_ERR_ARG, "combination_coeff is null");
    if (!out_jac || (n_points && !points)) return fail(GMSM_ERR_ARG, "gmsm_fold: null argument");
    if (nb_tasks > 1024) return fail(GMSM_ERR_CONFIG, "invalid config: config.NbTasks > 1024");
    if (n_points >= ((size_t)1 << 31)) return fail(GMSM_ERR_ARG, "n must be < 2^31");
    std::vector<uint64_t> powers;  // 1, g, g^2, ... (multiexp.go:331-337)
    try {
        powers.resize(n_points * (vt->scalar_bytes / 8));
    } catch (const std::exception &) {
        return fail(GMSM_ERR_DEVICE, "gmsm_fold: out of host memory for the powers of the coefficient");
    }
    vt->fold_powers(combination_coeff, n_points, powers.data());
    return gmsm_multiexp(group, points, n_points, powers.data(), n_points, nb_tasks, out_jac);
}

GMSM_EXPORT int gmsm_multiexp_device(int group, const void *d_points, const void *d_scalars, size_t n, void *hip_stream,
                                     uint64_t *out_jac) {
    VT_OR_FAIL(group);
    Context *ctx;
    int rc = get_context_of_pointer(d_scalars ? d_scalars : d_points, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    return vt->multiexp_device(*ctx, d_points, d_scalars, n, (hipStream_t)hip_stream, out_jac, nullptr);
}

// ------------------------------------------------------------------ resident bases
using BasesRef = std::shared_ptr<ResidentBases>;
static std::mutex g_bases_mu;
static std::vector<BasesRef> g_bases;  // handle = index + 1

// The returned reference keeps the bases alive for the caller's whole call (or ticket) even if another thread releases
// the handle meanwhile.
static BasesRef lookup_bases(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g_bases_mu);
    if (handle == 0 || handle > g_bases.size()) return nullptr;
    return g_bases[handle - 1];
}

GMSM_EXPORT int gmsm_bases_register(int group, const uint64_t *points, const void *d_points, size_t n,
                                    uint64_t *out_handle) {
    VT_OR_FAIL(group);
    if (!out_handle) return fail(GMSM_ERR_ARG, "gmsm_bases_register: out_handle is null");
    if ((points == nullptr) == (d_points == nullptr) && n)
        return fail(GMSM_ERR_ARG, "gmsm_bases_register: give exactly one of points (host) / d_points (device)");
    Context *ctx;
    int rc = get_context_of_pointer(d_points, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    Workspace &ws = *lease.w;
    const void *src = d_points;
    if (points && n) {
        if ((rc = ws.h2d_points.ensure(n * vt->aff_bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(ws.h2d_points.ptr, points, n * vt->aff_bytes, hipMemcpyHostToDevice, ws.stream));
        src = ws.h2d_points.ptr;
    } else if (n) {
        HIP_TRY(hipDeviceSynchronize());  // d_points may still be being written on a stream we do not know
    }
    BasesRef rb = std::make_shared<ResidentBases>();
    rb->group = group;
    rb->device = ctx->device;
    rc = vt->register_bases(*ctx, src, n, ws.stream, rb.get());
    if (rc) return rc;  // ~ResidentBases frees whatever was allocated
    std::lock_guard<std::mutex> lk2(g_bases_mu);
    g_bases.push_back(rb);
    *out_handle = g_bases.size();
    return GMSM_OK;
}

// ------------------------------------------------------------------ point ingest
static int point_error(const char *what, long long index, uint32_t status) {
    return fail(GMSM_ERR_POINT, std::string(what) + ": point " + std::to_string(index) + ": " + point_status_text(status));
}

static int register_device_points(const GroupVTable *vt, Context *ctx, Workspace &ws, int group, const void *d_points,
                                  size_t n, uint64_t *out_handle) {
    BasesRef rb = std::make_shared<ResidentBases>();
    rb->group = group;
    rb->device = ctx->device;
    int rc = vt->register_bases(*ctx, d_points, n, ws.stream, rb.get());
    if (rc) return rc;
    std::lock_guard<std::mutex> lk2(g_bases_mu);
    g_bases.push_back(rb);
    *out_handle = g_bases.size();
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_points_from_raw(int group, const uint8_t *raw, size_t n, int check, uint64_t *out_affine,
                                     void *d_out_affine, int64_t *bad_index) {
    VT_OR_FAIL(group);
    if (!bad_index) return fail(GMSM_ERR_ARG, "gmsm_points_from_raw: bad_index is null");
    *bad_index = -1;
    if (n && (!raw || (!out_affine && !d_out_affine))) return fail(GMSM_ERR_ARG, "gmsm_points_from_raw: null argument");
    if (n == 0) return GMSM_OK;
    Context *ctx;
    int rc = get_context_of_pointer(d_out_affine, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    Workspace &ws = *lease.w;
    const size_t bytes = n * vt->aff_bytes;
    if ((rc = ws.raw_bytes.ensure(bytes))) return rc;
    void *d_out = d_out_affine;
    if (!d_out) {
        if ((rc = ws.h2d_points.ensure(bytes))) return rc;
        d_out = ws.h2d_points.ptr;
    }
    HIP_TRY(hipMemcpyAsync(ws.raw_bytes.ptr, raw, bytes, hipMemcpyHostToDevice, ws.stream));
    long long bad = -1;
    uint32_t status = 0;
    if ((rc = vt->decode_raw(ws, ws.raw_bytes.ptr, n, check, d_out, &bad, &status))) return rc;
    if (out_affine) {
        HIP_TRY(hipMemcpyAsync(out_affine, d_out, bytes, hipMemcpyDeviceToHost, ws.stream));
        HIP_TRY(hipStreamSynchronize(ws.stream));
    }
    if (bad >= 0) {
        *bad_index = bad;
        return point_error("gmsm_points_from_raw", bad, status);
    }
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_points_validate(int group, const uint64_t *points, const void *d_points, size_t n, int check,
                                     int64_t *bad_index) {
    VT_OR_FAIL(group);
    if (!bad_index) return fail(GMSM_ERR_ARG, "gmsm_points_validate: bad_index is null");
    *bad_index = -1;
    if (n && (points == nullptr) == (d_points == nullptr))
        return fail(GMSM_ERR_ARG, "gmsm_points_validate: give exactly one of points (host) / d_points (device)");
    if (n == 0 || check <= 0) return GMSM_OK;
    Context *ctx;
    int rc = get_context_of_pointer(d_points, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    Workspace &ws = *lease.w;
    const void *src = d_points;
    if (points) {
        if ((rc = ws.h2d_points.ensure(n * vt->aff_bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(ws.h2d_points.ptr, points, n * vt->aff_bytes, hipMemcpyHostToDevice, ws.stream));
        src = ws.h2d_points.ptr;
    } else {
        HIP_TRY(hipDeviceSynchronize());  // d_points may still be being written on a stream we do not know
    }
    long long bad = -1;
    uint32_t status = 0;
    if ((rc = vt->validate_points(ws, src, n, check, &bad, &status))) return rc;
    if (bad >= 0) {
        *bad_index = bad;
        return point_error("gmsm_points_validate", bad, status);
    }
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_bases_register_raw(int group, const uint8_t *raw, size_t n, int check, uint64_t *out_handle,
                                        int64_t *bad_index) {
    VT_OR_FAIL(group);
    if (!out_handle || !bad_index) return fail(GMSM_ERR_ARG, "gmsm_bases_register_raw: null argument");
    *bad_index = -1;
    if (n && !raw) return fail(GMSM_ERR_ARG, "gmsm_bases_register_raw: raw is null");
    Context *ctx;
    int rc = get_context(&ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    Workspace &ws = *lease.w;
    const size_t bytes = n * vt->aff_bytes;
    if (n) {
        if ((rc = ws.raw_bytes.ensure(bytes))) return rc;
        if ((rc = ws.h2d_points.ensure(bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(ws.raw_bytes.ptr, raw, bytes, hipMemcpyHostToDevice, ws.stream));
        long long bad = -1;
        uint32_t status = 0;
        if ((rc = vt->decode_raw(ws, ws.raw_bytes.ptr, n, check, ws.h2d_points.ptr, &bad, &status))) return rc;
        if (bad >= 0) {
            *bad_index = bad;
            return point_error("gmsm_bases_register_raw", bad, status);
        }
    }
    return register_device_points(vt, ctx, ws, group, ws.h2d_points.ptr, n, out_handle);
}

// The Encoder's default (compressed) format: X and the flag of Y's half (gmsm_decompress.h). Same shape as the raw entries.
GMSM_EXPORT int gmsm_points_from_compressed(int group, const uint8_t *comp, size_t n, int check, uint64_t *out_affine,
                                            void *d_out_affine, int64_t *bad_index) {
    VT_OR_FAIL(group);
    if (!bad_index) return fail(GMSM_ERR_ARG, "gmsm_points_from_compressed: bad_index is null");
    *bad_index = -1;
    if (n && (!comp || (!out_affine && !d_out_affine))) return fail(GMSM_ERR_ARG, "gmsm_points_from_compressed: null argument");
    if (n == 0) return GMSM_OK;
    Context *ctx;
    int rc = get_context_of_pointer(d_out_affine, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    Workspace &ws = *lease.w;
    const size_t bytes = n * vt->aff_bytes, cbytes = bytes / 2;
    if ((rc = ws.raw_bytes.ensure(cbytes))) return rc;
    void *d_out = d_out_affine;
    if (!d_out) {
        if ((rc = ws.h2d_points.ensure(bytes))) return rc;
        d_out = ws.h2d_points.ptr;
    }
    HIP_TRY(hipMemcpyAsync(ws.raw_bytes.ptr, comp, cbytes, hipMemcpyHostToDevice, ws.stream));
    long long bad = -1;
    uint32_t status = 0;
    if ((rc = vt->decode_compressed(ws, ws.raw_bytes.ptr, n, check, d_out, &bad, &status))) return rc;
    if (out_affine) {
        HIP_TRY(hipMemcpyAsync(out_affine, d_out, bytes, hipMemcpyDeviceToHost, ws.stream));
        HIP_TRY(hipStreamSynchronize(ws.stream));
    }
    if (bad >= 0) {
        *bad_index = bad;
        return point_error("gmsm_points_from_compressed", bad, status);
    }
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_points_compress(int group, const uint64_t *points, const void *d_points, size_t n, uint8_t *out_comp) {
    VT_OR_FAIL(group);
    if (n && ((points == nullptr) == (d_points == nullptr) || !out_comp))
        return fail(GMSM_ERR_ARG, "gmsm_points_compress: give exactly one of points (host) / d_points (device), and out_comp");
    if (n == 0) return GMSM_OK;
    Context *ctx;
    int rc = get_context_of_pointer(d_points, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    Workspace &ws = *lease.w;
    const size_t bytes = n * vt->aff_bytes, cbytes = bytes / 2;
    if ((rc = ws.raw_bytes.ensure(cbytes))) return rc;
    const void *src = d_points;
    if (points) {
        if ((rc = ws.h2d_points.ensure(bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(ws.h2d_points.ptr, points, bytes, hipMemcpyHostToDevice, ws.stream));
        src = ws.h2d_points.ptr;
    } else {
        HIP_TRY(hipDeviceSynchronize());  // d_points may still be being written on a stream we do not know
    }
    if ((rc = vt->encode_compressed(ws, src, n, ws.raw_bytes.ptr))) return rc;
    HIP_TRY(hipMemcpyAsync(out_comp, ws.raw_bytes.ptr, cbytes, hipMemcpyDeviceToHost, ws.stream));
    HIP_TRY(hipStreamSynchronize(ws.stream));
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_bases_register_compressed(int group, const uint8_t *comp, size_t n, int check, uint64_t *out_handle,
                                               int64_t *bad_index) {
    VT_OR_FAIL(group);
    if (!out_handle || !bad_index) return fail(GMSM_ERR_ARG, "gmsm_bases_register_compressed: null argument");
    *bad_index = -1;
    if (n && !comp) return fail(GMSM_ERR_ARG, "gmsm_bases_register_compressed: comp is null");
    Context *ctx;
    int rc = get_context(&ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    Workspace &ws = *lease.w;
    const size_t bytes = n * vt->aff_bytes;
    if (n) {
        if ((rc = ws.raw_bytes.ensure(bytes / 2))) return rc;
        if ((rc = ws.h2d_points.ensure(bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(ws.raw_bytes.ptr, comp, bytes / 2, hipMemcpyHostToDevice, ws.stream));
        long long bad = -1;
        uint32_t status = 0;
        if ((rc = vt->decode_compressed(ws, ws.raw_bytes.ptr, n, check, ws.h2d_points.ptr, &bad, &status))) return rc;
        if (bad >= 0) {
            *bad_index = bad;
            return point_error("gmsm_bases_register_compressed", bad, status);
        }
    }
    return register_device_points(vt, ctx, ws, group, ws.h2d_points.ptr, n, out_handle);
}

GMSM_EXPORT int gmsm_bases_register_dump(int group, const char *path, uint64_t offset, int expect_marker, size_t max_points,
                                         int check, uint64_t *out_handle, size_t *out_n, int64_t *bad_index) {
    VT_OR_FAIL(group);
    if (!path || !out_handle || !out_n || !bad_index) return fail(GMSM_ERR_ARG, "gmsm_bases_register_dump: null argument");
    *bad_index = -1;
    *out_n = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(GMSM_ERR_ARG, std::string("gmsm_bases_register_dump: cannot open ") + path);
    struct Closer {
        FILE *f;
        ~Closer() { fclose(f); }
    } closer{f};
    if (fseeko(f, (off_t)offset, SEEK_SET) != 0) return fail(GMSM_ERR_ARG, "gmsm_bases_register_dump: bad offset");
    uint64_t word = 0;
    if (expect_marker) {  // unsafe.ReadMarker, utils/unsafe/dump_slice.go:91-103
        if (fread(&word, 8, 1, f) != 1) return fail(GMSM_ERR_ARG, "gmsm_bases_register_dump: short read (marker)");
        if (word != 0xdeadbeefull)
            return fail(GMSM_ERR_ARG, "marker mismatch: dump was not written on the same architecture");
    }
    if (fread(&word, 8, 1, f) != 1) return fail(GMSM_ERR_ARG, "gmsm_bases_register_dump: short read (length)");
    size_t n = (size_t)word;
    if (max_points && n > max_points) n = max_points;  // ReadSlice's maxElements
    if (n >= ((size_t)1 << 31)) return fail(GMSM_ERR_ARG, "gmsm_bases_register_dump: more than 2^31 points");
    {   // the length word comes from the file: believe it only as far as the file goes (before any device memory is asked for)
        const off_t here = ftello(f);
        struct stat st;
        if (here >= 0 && fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode)) {
            const uint64_t left = st.st_size > here ? (uint64_t)(st.st_size - here) : 0;
            if ((uint64_t)n * vt->aff_bytes > left)
                return fail(GMSM_ERR_ARG, "gmsm_bases_register_dump: short read (points): the file holds fewer points than its length word says");
        }
    }
    Context *ctx;
    int rc = get_context(&ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    Workspace &ws = *lease.w;
    const size_t bytes = n * vt->aff_bytes;
    if (n) {
        if ((rc = ws.h2d_points.ensure(bytes))) return rc;
        // file -> two pinned buffers -> HBM: the read of chunk k+1 runs while chunk k crosses PCIe
        constexpr size_t CHUNK = (size_t)32 << 20;
        void *pin[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        auto cleanup = [&]() {
            for (int i = 0; i < 2; ++i) {
                if (done[i]) (void)hipEventDestroy(done[i]);
                if (pin[i]) (void)hipHostFree(pin[i]);
            }
        };
        for (int i = 0; i < 2; ++i) {
            if (hipHostMalloc(&pin[i], CHUNK, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) {
                cleanup();
                return fail(GMSM_ERR_DEVICE, "gmsm_bases_register_dump: cannot allocate pinned staging buffers");
            }
        }
        size_t pos = 0;
        for (unsigned k = 0; pos < bytes; ++k) {
            const size_t len = std::min(CHUNK, bytes - pos);
            const int b = (int)(k & 1u);
            if (k >= 2) (void)hipEventSynchronize(done[b]);
            if (fread(pin[b], 1, len, f) != len) {
                (void)hipStreamSynchronize(ws.stream);
                cleanup();
                return fail(GMSM_ERR_ARG, "gmsm_bases_register_dump: short read (points)");
            }
            hipError_t e = hipMemcpyAsync((char *)ws.h2d_points.ptr + pos, pin[b], len, hipMemcpyHostToDevice, ws.stream);
            if (e == hipSuccess) e = hipEventRecord(done[b], ws.stream);
            if (e != hipSuccess) {
                (void)hipStreamSynchronize(ws.stream);
                cleanup();
                return fail(GMSM_ERR_DEVICE, std::string("gmsm_bases_register_dump: ") + hipGetErrorString(e));
            }
            pos += len;
        }
        (void)hipStreamSynchronize(ws.stream);
        cleanup();
        if (check > 0) {
            long long bad = -1;
            uint32_t status = 0;
            if ((rc = vt->validate_points(ws, ws.h2d_points.ptr, n, check, &bad, &status))) return rc;
            if (bad >= 0) {
                *bad_index = bad;
                return point_error("gmsm_bases_register_dump", bad, status);
            }
        }
    }
    if ((rc = register_device_points(vt, ctx, ws, group, ws.h2d_points.ptr, n, out_handle))) return rc;
    *out_n = n;
    return GMSM_OK;
}

// Bases registered on several devices (full copies: 1 GiB per device for 2^24 BN254 G1 points, of 288 GB): any prefix of
// them can then be cut into equal point slices - or into window sets - over the ranks, whatever its length.
struct ShardedBases {
    int group = -1;
    size_t n = 0;
    std::vector<int> devices;                                  // logical ranks
    std::map<int, std::shared_ptr<ResidentBases>> replicas;    // device -> its copy
};
constexpr uint64_t SHARDED_TAG = (uint64_t)1 << 62;
static std::vector<std::shared_ptr<ShardedBases>> g_sharded;  // handle = SHARDED_TAG | (index + 1); under g_bases_mu

static std::shared_ptr<ShardedBases> lookup_sharded(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g_bases_mu);
    const uint64_t i = handle & ~SHARDED_TAG;
    if (!(handle & SHARDED_TAG) || i == 0 || i > g_sharded.size()) return nullptr;
    return g_sharded[i - 1];
}

GMSM_EXPORT int gmsm_bases_register_sharded(int group, const uint64_t *points, size_t n, const int *devices, int n_devices,
                                            uint64_t *out_handle) {
    VT_OR_FAIL(group);
    if (!out_handle) return fail(GMSM_ERR_ARG, "gmsm_bases_register_sharded: out_handle is null");
    if (n && !points) return fail(GMSM_ERR_ARG, "gmsm_bases_register_sharded: points is null");
    auto sb = std::make_shared<ShardedBases>();
    sb->group = group;
    sb->n = n;
    if (devices && n_devices > 0) sb->devices.assign(devices, devices + n_devices);
    else if (int rc0 = shard_devices(sb->devices)) return rc0;
    const int ndev = gmsm_device_count();
    if (ndev <= 0) return fail(GMSM_ERR_DEVICE, "no usable HIP device; libgmsm has no CPU fallback");
    if (sb->devices.empty())
        for (int d = 0; d < ndev; ++d) sb->devices.push_back(d);
    for (int d : sb->devices)
        if (d < 0 || d >= ndev)
            return fail(GMSM_ERR_DEVICE, "gmsm_bases_register_sharded: device " + std::to_string(d) + " does not exist (" +
                                             std::to_string(ndev) + " visible)");
    // one upload per distinct device, all of them at once (every device has its own PCIe link)
    std::vector<int> distinct;
    for (int d : sb->devices)
        if (std::find(distinct.begin(), distinct.end(), d) == distinct.end()) distinct.push_back(d);
    std::vector<uint64_t> handles(distinct.size(), 0);
    std::vector<int> rcs(distinct.size(), GMSM_OK);
    std::vector<std::string> errs(distinct.size());
    std::vector<std::thread> th;
    for (size_t i = 0; i < distinct.size(); ++i)
        th.emplace_back([&, i] {
            g_device = distinct[i];  // the worker's own device: gmsm_bases_register uploads to the calling thread's
            rcs[i] = gmsm_bases_register(group, points, nullptr, n, &handles[i]);
            if (rcs[i]) errs[i] = gmsm_last_error();
        });
    for (auto &t : th) t.join();
    int rc = GMSM_OK;
    for (size_t i = 0; i < distinct.size(); ++i) {
        if (rcs[i] != GMSM_OK && rc == GMSM_OK) rc = fail(rcs[i], "device " + std::to_string(distinct[i]) + ": " + errs[i]);
        if (handles[i]) {
            sb->replicas[distinct[i]] = lookup_bases(handles[i]);
            (void)gmsm_bases_release(handles[i]);  // the per-device table entry goes; `replicas` keeps the bases alive
        }
    }
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_bases_mu);
    g_sharded.push_back(sb);
    *out_handle = SHARDED_TAG | (uint64_t)g_sharded.size();
    return GMSM_OK;
}

// Window tables of registered bases: slab w = 2^(c w) P_i for every window of the c-bit decomposition, nwin copies of
// the bases in HBM; MultiExp calls over the handle then fill ONE bucket set (Group::precompute_tables). c = 0: the
// library's width for this many bases.
static int precompute_on(const BasesRef &rb, unsigned c) {
    const GroupVTable *vt = vtable(rb->group);
    std::lock_guard<std::mutex> only_one(rb->tab_mu);
    if (rb->tab_c.load() != 0) {
        if (c == 0 || c == rb->tab_c.load()) return GMSM_OK;
        return fail(GMSM_ERR_ARG, "gmsm_bases_precompute: the handle already has tables of another width");
    }
    Context *ctx;
    int rc = get_context_for(rb->device, &ctx);
    if (rc) return rc;
    struct RestoreDevice {  // every exit path leaves the calling thread on the device it came with
        int prev = 0;
        RestoreDevice() { (void)hipGetDevice(&prev); }
        ~RestoreDevice() { (void)hipSetDevice(prev); }
    } restore;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    return vt->precompute_tables(*ctx, *lease.w, rb.get(), c);
}

GMSM_EXPORT int gmsm_bases_precompute(uint64_t handle, unsigned c) {
    if (handle & SHARDED_TAG) {
        std::shared_ptr<ShardedBases> sb = lookup_sharded(handle);
        if (!sb) return fail(GMSM_ERR_ARG, "unknown bases handle");
        std::vector<std::thread> th;
        std::vector<int> rcs(sb->replicas.size(), GMSM_OK);
        std::vector<std::string> errs(sb->replicas.size());
        size_t i = 0;
        for (auto &kv : sb->replicas) {
            BasesRef rb = kv.second;
            th.emplace_back([&, rb, i] {
                rcs[i] = precompute_on(rb, c);
                if (rcs[i]) errs[i] = gmsm_last_error();
            });
            ++i;
        }
        for (auto &t : th) t.join();
        for (size_t k = 0; k < rcs.size(); ++k)
            if (rcs[k]) return fail(rcs[k], errs[k]);
        return GMSM_OK;
    }
    BasesRef rb = lookup_bases(handle);
    if (!rb) return fail(GMSM_ERR_ARG, "unknown bases handle");
    return precompute_on(rb, c);
}

GMSM_EXPORT unsigned long gmsm_debug_table_runs(void) { return g_table_runs.load(); }
GMSM_EXPORT unsigned long gmsm_debug_small_runs(void) { return g_small_runs.load(); }

// 0 = no tables; else the window width of the handle's tables
GMSM_EXPORT unsigned gmsm_bases_table_bits(uint64_t handle) {
    if (handle & SHARDED_TAG) {
        std::shared_ptr<ShardedBases> sb = lookup_sharded(handle);
        if (!sb || sb->replicas.empty()) return 0;
        return sb->replicas.begin()->second->tab_c.load();
    }
    BasesRef rb = lookup_bases(handle);
    return rb ? rb->tab_c.load() : 0;
}

GMSM_EXPORT int gmsm_bases_release(uint64_t handle) {
    BasesRef rb;
    std::shared_ptr<ShardedBases> sb;
    {
        std::lock_guard<std::mutex> lk(g_bases_mu);
        if (handle & SHARDED_TAG) {
            const uint64_t i = handle & ~SHARDED_TAG;
            if (i == 0 || i > g_sharded.size() || !g_sharded[i - 1]) return fail(GMSM_ERR_ARG, "unknown bases handle");
            sb.swap(g_sharded[i - 1]);
        } else {
            if (handle == 0 || handle > g_bases.size() || !g_bases[handle - 1]) return fail(GMSM_ERR_ARG, "unknown bases handle");
            rb.swap(g_bases[handle - 1]);
        }
    }
    // Calls and tickets that are still using the bases hold their own references; the memory goes with the last one. A
    // reference that an enqueue-only call parked on an idle workspace would otherwise live until that workspace is leased
    // again: drop it here once its work has finished (outside the context lock - the destructor synchronises the device).
    std::vector<Context *> ctxs;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        ctxs = g_ctx;
    }
    for (Context *c : ctxs) {
        if (!c) continue;
        BasesRef parked[Context::NUM_WS];
        {
            std::lock_guard<std::mutex> lk(c->mu);
            for (int i = 0; i < Context::NUM_WS; ++i) {
                Workspace &w = c->ws[i];
                if (w.busy || !w.bases_ref) continue;
                const bool mine = w.bases_ref == rb || (sb && sb->replicas.count(c->device) && sb->replicas[c->device] == w.bases_ref);
                if (mine && (!w.last_use || hipEventQuery(w.last_use) == hipSuccess)) parked[i].swap(w.bases_ref);
            }
        }
    }
    return GMSM_OK;
}

static int multiexp_bases_impl(uint64_t handle, const uint64_t *scalars, const void *d_scalars, size_t n, int nb_tasks,
                               void *hip_stream, uint64_t *out_jac) {
    BasesRef rb = lookup_bases(handle);
    if (!rb) return fail(GMSM_ERR_ARG, "unknown bases handle");
    const GroupVTable *vt = vtable(rb->group);
    if (n > rb->n) return fail(GMSM_ERR_LEN, "len(points) != len(scalars)");  // more scalars than registered bases
    if (nb_tasks > 1024) return fail(GMSM_ERR_CONFIG, "invalid config: config.NbTasks > 1024");
    Context *ctx;
    int rc = get_context_for(rb->device, &ctx);  // the bases decide the device, not the calling thread
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if (scalars && n) return vt->multiexp_bases_host(*ctx, scalars, n, out_jac, rb.get());
    return vt->multiexp_device(*ctx, nullptr, d_scalars, n, (hipStream_t)hip_stream, out_jac, rb.get());
}

static int multiexp_bases_sharded(uint64_t handle, const uint64_t *scalars, size_t n, int nb_tasks, int mode, uint64_t *out_jac) {
    std::shared_ptr<ShardedBases> sb = lookup_sharded(handle);
    if (!sb) return fail(GMSM_ERR_ARG, "unknown bases handle");
    const GroupVTable *vt = vtable(sb->group);
    if (n > sb->n) return fail(GMSM_ERR_LEN, "len(points) != len(scalars)");  // more scalars than registered bases
    bool done;
    int rc = multiexp_precheck(vt, n, n, nb_tasks, out_jac, &done);
    if (rc || done) return rc;
    if (!scalars) return fail(GMSM_ERR_ARG, "scalars is null");
    return multiexp_sharded_run(sb->group, nullptr, &sb->replicas, scalars, n, sb->devices, mode, out_jac);
}

GMSM_EXPORT int gmsm_multiexp_bases(uint64_t handle, const uint64_t *scalars, size_t n_scalars, int nb_tasks,
                                    uint64_t *out_jac) {
    if (handle & SHARDED_TAG) return multiexp_bases_sharded(handle, scalars, n_scalars, nb_tasks, 0, out_jac);
    return multiexp_bases_impl(handle, scalars, nullptr, n_scalars, nb_tasks, nullptr, out_jac);
}

GMSM_EXPORT int gmsm_multiexp_bases_sharded(uint64_t handle, const uint64_t *scalars, size_t n_scalars, int nb_tasks, int mode,
                                            uint64_t *out_jac) {
    if (!(handle & SHARDED_TAG)) return fail(GMSM_ERR_ARG, "gmsm_multiexp_bases_sharded: not a handle of gmsm_bases_register_sharded");
    return multiexp_bases_sharded(handle, scalars, n_scalars, nb_tasks, mode, out_jac);
}

GMSM_EXPORT int gmsm_multiexp_bases_device(uint64_t handle, const void *d_scalars, size_t n_scalars, void *hip_stream,
                                           uint64_t *out_jac) {
    return multiexp_bases_impl(handle, nullptr, d_scalars, n_scalars, 0, hip_stream, out_jac);
}

// ------------------------------------------------------------------ two MultiExp calls in flight
// ticket = device << 40 | generation << 8 | (slot + 1)
GMSM_EXPORT int gmsm_multiexp_bases_submit(uint64_t handle, const void *d_scalars, size_t n_scalars, void *hip_stream,
                                           uint64_t *out_ticket) {
    BasesRef rb = lookup_bases(handle);
    if (!rb) return fail(GMSM_ERR_ARG, "unknown bases handle");
    if (!out_ticket) return fail(GMSM_ERR_ARG, "out_ticket is null");
    const GroupVTable *vt = vtable(rb->group);
    if (n_scalars > rb->n) return fail(GMSM_ERR_LEN, "len(points) != len(scalars)");
    Context *ctx;
    int rc = get_context_for(rb->device, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    int why = 0;
    Lease lease(*ctx, /*wait=*/false, /*for_ticket=*/true, &why);
    Workspace *ws = lease.w;
    if (!ws)
        return why == 1 ? fail(GMSM_ERR_ARG, "two submitted MultiExp calls are outstanding on this device: collect one first")
                        : fail(GMSM_ERR_DEVICE, "gmsm_shutdown ran while this submit waited for a workspace");
    // the scalars are produced on the caller's stream (NULL = the default stream): order our stream behind it
    if ((rc = order_after(*ws, (hipStream_t)hip_stream))) return rc;
    if ((rc = vt->submit(*ctx, *ws, d_scalars, n_scalars, rb.get()))) return rc;
    lease.keep();  // released by gmsm_multiexp_collect
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ws->bases_ref = rb;  // the ticket owns a reference until it is collected
        ws->pending = true;
        ws->pending_group = rb->group;
        ws->pending_gen = g_ticket_gen.fetch_add(1, std::memory_order_relaxed) + 1;  // unique across gmsm_shutdown
    }
    *out_ticket = ((uint64_t)ctx->device << 40) | ((uint64_t)(ws->pending_gen & 0xffffffffu) << 8) | (uint64_t)(ws - ctx->ws + 1);
    return GMSM_OK;
}

// k MultiExp over the same registered bases, one scalar vector each (kzg.Commit over many polynomials with one SRS,
// ecc/bn254/kzg/kzg.go:159-176, 246-300): a single blocking call that keeps two of them in flight. With host scalars
// the copy of vector i+1 runs while vector i is being accumulated.
static int multiexp_bases_batch_on(const BasesRef &rb, const uint64_t *scalars, const void *d_scalars, size_t n, size_t k,
                                   void *hip_stream, uint64_t *out_jac);

GMSM_EXPORT int gmsm_multiexp_bases_batch(uint64_t handle, const uint64_t *scalars, const void *d_scalars, size_t n,
                                          size_t k, void *hip_stream, uint64_t *out_jac) {
    if (handle & SHARDED_TAG) {
        // Bases registered on several devices: the k vectors are independent MultiExp calls, so they need no sharding of
        // a single one - rank r (one host thread per logical rank) runs the contiguous block of vectors [r k/G, (r+1) k/G)
        // on its device's copy of the bases; no exchange at all (the replica mode of sharding.py, inside the library).
        std::shared_ptr<ShardedBases> sb = lookup_sharded(handle);
        if (!sb) return fail(GMSM_ERR_ARG, "unknown bases handle");
        if (n > sb->n) return fail(GMSM_ERR_LEN, "len(points) != len(scalars)");
        if (k == 0) return GMSM_OK;
        if (n && !scalars) return fail(GMSM_ERR_ARG, "gmsm_multiexp_bases_batch: a sharded handle takes host scalars");
        const GroupVTable *vts = vtable(sb->group);
        const size_t G = std::min<size_t>(sb->devices.size(), k), jl = vts->jac_bytes / 8, sl = vts->scalar_bytes / 8 * n;
        std::vector<int> rcs(G, GMSM_OK);
        std::vector<std::string> errs(G);
        std::vector<std::thread> th;
        for (size_t r = 0; r < G; ++r)
            th.emplace_back([&, r] {
                const size_t lo = r * k / G, hi = (r + 1) * k / G;
                const BasesRef &rb = sb->replicas.at(sb->devices[r]);
                rcs[r] = multiexp_bases_batch_on(rb, scalars ? scalars + lo * sl : nullptr, nullptr, n, hi - lo, nullptr, out_jac + lo * jl);
                if (rcs[r]) errs[r] = gmsm_last_error();
            });
        for (auto &t : th) t.join();
        for (size_t r = 0; r < G; ++r)
            if (rcs[r]) return fail(rcs[r], "rank " + std::to_string(r) + " (device " + std::to_string(sb->devices[r]) + "): " + errs[r]);
        return GMSM_OK;
    }
    BasesRef rb = lookup_bases(handle);
    if (!rb) return fail(GMSM_ERR_ARG, "unknown bases handle");
    if (n > rb->n) return fail(GMSM_ERR_LEN, "len(points) != len(scalars)");
    if (k == 0) return GMSM_OK;
    if (n && (scalars == nullptr) == (d_scalars == nullptr))
        return fail(GMSM_ERR_ARG, "gmsm_multiexp_bases_batch: give exactly one of scalars (host) / d_scalars (device)");
    return multiexp_bases_batch_on(rb, scalars, d_scalars, n, k, hip_stream, out_jac);
}

static int multiexp_bases_batch_on(const BasesRef &rb, const uint64_t *scalars, const void *d_scalars, size_t n, size_t k,
                                   void *hip_stream, uint64_t *out_jac) {
    const GroupVTable *vt = vtable(rb->group);
    Context *ctx;
    int rc = get_context_for(rb->device, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t jl = vt->jac_bytes / 8, sb = vt->scalar_bytes * n;
    if (n == 0) {
        for (size_t i = 0; i < k; ++i)
            if ((rc = vt->multiexp_device(*ctx, nullptr, nullptr, 0, nullptr, out_jac + i * jl, rb.get()))) return rc;
        return GMSM_OK;
    }
    Workspace *w[2] = {ctx->acquire(true), nullptr};
    if (!w[0]) return fail(GMSM_ERR_DEVICE, "no workspace could be leased (internal error)");
    w[1] = ctx->acquire(false);  // a concurrent caller may hold it: then this batch runs one call at a time
    const size_t nws = w[1] ? 2 : 1;
    size_t submitted = 0, collected = 0;
    rc = GMSM_OK;
    while (rc == GMSM_OK && collected < k) {
        while (rc == GMSM_OK && submitted < k && submitted - collected < nws) {
            Workspace &ws = *w[submitted % nws];
            const void *dsc;
            if (scalars) {
                if ((rc = ws.h2d_scalars.ensure(sb))) break;
                hipError_t e = hipMemcpyAsync(ws.h2d_scalars.ptr, (const char *)scalars + submitted * sb, sb,
                                              hipMemcpyHostToDevice, ws.stream);
                if (e != hipSuccess) {
                    rc = fail(GMSM_ERR_DEVICE, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
                    break;
                }
                dsc = ws.h2d_scalars.ptr;
            } else {
                if ((rc = order_after(ws, (hipStream_t)hip_stream))) break;
                dsc = (const char *)d_scalars + submitted * sb;
            }
            if ((rc = vt->submit(*ctx, ws, dsc, n, rb.get()))) break;
            ++submitted;
        }
        if (rc) break;
        rc = vt->collect(*w[collected % nws], out_jac + collected * jl);
        ++collected;
    }
    for (size_t i = 0; i < nws; ++i) {
        (void)hipStreamSynchronize(w[i]->stream);  // nothing of this call is left running on a released workspace
        ctx->release(w[i]);
    }
    return rc;
}

GMSM_EXPORT int gmsm_multiexp_collect(uint64_t ticket, uint64_t *out_jac) {
    const unsigned slot = (unsigned)(ticket & 0xff), dev = (unsigned)(ticket >> 40);
    const uint32_t gen = (uint32_t)(ticket >> 8);
    Context *ctx = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        if (dev < g_ctx.size()) ctx = g_ctx[dev];
    }
    if (!ctx || slot < 1 || slot > (unsigned)Context::NUM_WS) return fail(GMSM_ERR_ARG, "unknown MultiExp ticket");
    Workspace &ws = ctx->ws[slot - 1];
    const GroupVTable *vt;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (!ws.pending || ws.pending_gen != gen) return fail(GMSM_ERR_ARG, "unknown MultiExp ticket (already collected?)");
        ws.pending = false;  // claimed by this call; the workspace stays leased (busy) until the fold is done
        vt = vtable(ws.pending_group);
    }
    // The slot stays `pending` while we wait and fold, so nobody else touches it; the context lock is not held and the
    // other slot can be submitted to meanwhile.
    (void)hipSetDevice(ctx->device);
    int rc = vt->collect(ws, out_jac);
    ws.bases_ref.reset();
    ctx->release(&ws);
    return rc;
}

// ------------------------------------------------------------------ fr/fft
using FftRef = std::shared_ptr<FftDomain>;
static std::mutex g_fft_mu;
static std::vector<FftRef> g_fft;  // handle = index + 1

static FftRef lookup_fft(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g_fft_mu);
    if (handle == 0 || handle > g_fft.size()) return nullptr;
    return g_fft[handle - 1];
}

GMSM_EXPORT int gmsm_fft_domain_new(int group, uint64_t m, uint64_t *out_handle) {
    VT_OR_FAIL(group);
    if (!out_handle) return fail(GMSM_ERR_ARG, "gmsm_fft_domain_new: out_handle is null");
    unsigned log2n = 0;  // ecc.NextPowerOfTwo(m)
    while (log2n < 63 && ((uint64_t)1 << log2n) < m) ++log2n;
    Context *ctx;
    int rc = get_context(&ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    FftRef d = std::make_shared<FftDomain>();
    d->group = group;
    d->device = ctx->device;
    if ((rc = vt->fft_domain_new(*ctx, lease.w->stream, log2n, d.get()))) return rc;
    std::lock_guard<std::mutex> lk(g_fft_mu);
    g_fft.push_back(d);
    *out_handle = g_fft.size();
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_fft_domain_release(uint64_t handle) {
    FftRef d;
    std::lock_guard<std::mutex> lk(g_fft_mu);
    if (handle == 0 || handle > g_fft.size() || !g_fft[handle - 1]) return fail(GMSM_ERR_ARG, "unknown fft domain handle");
    d.swap(g_fft[handle - 1]);
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_fft_domain_info(uint64_t handle, uint64_t *cardinality, uint64_t *generator, uint64_t *generator_inv,
                                     uint64_t *cardinality_inv, uint64_t *fr_multiplicative_gen,
                                     uint64_t *fr_multiplicative_gen_inv) {
    FftRef d = lookup_fft(handle);
    if (!d) return fail(GMSM_ERR_ARG, "unknown fft domain handle");
    if (cardinality) *cardinality = (uint64_t)1 << d->log2n;
    auto put = [](uint64_t *dst, const std::vector<uint64_t> &src) {
        if (dst) memcpy(dst, src.data(), src.size() * 8);
    };
    put(generator, d->generator);
    put(generator_inv, d->generator_inv);
    put(cardinality_inv, d->cardinality_inv);
    put(fr_multiplicative_gen, d->shift);
    put(fr_multiplicative_gen_inv, d->shift_inv);
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_fft(uint64_t handle, uint64_t *a, void *d_a, size_t n, int inverse, int decimation, int on_coset,
                         void *hip_stream) {
    FftRef d = lookup_fft(handle);
    if (!d) return fail(GMSM_ERR_ARG, "unknown fft domain handle");
    const GroupVTable *vt = vtable(d->group);
    if (n != ((size_t)1 << d->log2n)) return fail(GMSM_ERR_LEN, "len(a) must equal the domain's cardinality");
    if ((a == nullptr) == (d_a == nullptr)) return fail(GMSM_ERR_ARG, "gmsm_fft: give exactly one of a (host) / d_a (device)");
    if (decimation != 0 && decimation != 1) return fail(GMSM_ERR_ARG, "gmsm_fft: decimation must be 0 (DIT) or 1 (DIF)");
    Context *ctx;
    int rc = get_context_for(d->device, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    Workspace &ws = *lease.w;
    const size_t bytes = n * vt->scalar_bytes;
    void *dev = d_a;
    if (a) {
        if ((rc = ws.h2d_scalars.ensure(bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(ws.h2d_scalars.ptr, a, bytes, hipMemcpyHostToDevice, ws.stream));
        dev = ws.h2d_scalars.ptr;
    } else {
        hipPointerAttribute_t attr;  // a vector on another GPU would fault (or read garbage) under this domain's tables
        if (hipPointerGetAttributes(&attr, d_a) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != d->device) {
            (void)hipGetLastError();
            return fail(GMSM_ERR_ARG, "gmsm_fft: d_a is not device memory of the domain's device");
        }
        if ((rc = order_after(ws, (hipStream_t)hip_stream))) return rc;
    }
    {
        std::lock_guard<std::mutex> lk(d->mu);
        rc = vt->fft_run(ws.stream, d.get(), dev, inverse != 0, decimation == 1, on_coset != 0);
    }
    if (rc) return rc;
    if (a) HIP_TRY(hipMemcpyAsync(a, dev, bytes, hipMemcpyDeviceToHost, ws.stream));
    HIP_TRY(hipStreamSynchronize(ws.stream));
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_fft_bit_reverse(int group, uint64_t *a, void *d_a, size_t n, void *hip_stream) {
    VT_OR_FAIL(group);
    if ((a == nullptr) == (d_a == nullptr) && n) return fail(GMSM_ERR_ARG, "gmsm_fft_bit_reverse: give exactly one of a / d_a");
    if (n == 0) return GMSM_OK;
    Context *ctx;
    int rc = get_context_of_pointer(d_a, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    Workspace &ws = *lease.w;
    const size_t bytes = n * vt->scalar_bytes;
    void *dev = d_a;
    if (a) {
        if ((rc = ws.h2d_scalars.ensure(bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(ws.h2d_scalars.ptr, a, bytes, hipMemcpyHostToDevice, ws.stream));
        dev = ws.h2d_scalars.ptr;
    } else if ((rc = order_after(ws, (hipStream_t)hip_stream))) {
        return rc;
    }
    if ((rc = vt->fft_bit_reverse(ws.stream, dev, n))) return rc;
    if (a) HIP_TRY(hipMemcpyAsync(a, dev, bytes, hipMemcpyDeviceToHost, ws.stream));
    HIP_TRY(hipStreamSynchronize(ws.stream));
    return GMSM_OK;
}

// ------------------------------------------------------------------ fixed-base batch
GMSM_EXPORT int gmsm_batch_scalar_mul(int group, const uint64_t *base_affine, const uint64_t *scalars, size_t n,
                                      uint64_t *out_affine) {
    VT_OR_FAIL(group);
    if (!base_affine || (n && (!scalars || !out_affine))) return fail(GMSM_ERR_ARG, "gmsm_batch_scalar_mul: null argument");
    Context *ctx;
    int rc = get_context(&ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    return vt->batch_scalar_mul(*ctx, base_affine, scalars, nullptr, n, nullptr, out_affine, nullptr);
}

GMSM_EXPORT int gmsm_batch_scalar_mul_device(int group, const uint64_t *base_affine, const void *d_scalars, size_t n,
                                             void *hip_stream, void *d_out_affine) {
    VT_OR_FAIL(group);
    if (!base_affine || (n && (!d_scalars || !d_out_affine)))
        return fail(GMSM_ERR_ARG, "gmsm_batch_scalar_mul_device: null argument");
    Context *ctx;
    int rc = get_context_of_pointer(d_scalars, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    return vt->batch_scalar_mul(*ctx, base_affine, nullptr, d_scalars, n, (hipStream_t)hip_stream, nullptr, d_out_affine);
}

GMSM_EXPORT int gmsm_batch_jac_to_affine(int group, const uint64_t *jac, size_t n, uint64_t *out_affine) {
    VT_OR_FAIL(group);
    if (n && (!jac || !out_affine)) return fail(GMSM_ERR_ARG, "gmsm_batch_jac_to_affine: null argument");
    Context *ctx;
    int rc = get_context(&ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    return vt->batch_jac_to_affine(*ctx, jac, n, out_affine);
}

GMSM_EXPORT unsigned gmsm_default_window_bits(int group, size_t n) {
    const GroupVTable *vt = vtable(group);
    return vt ? choose_c(vt->fr_bits, vt->aff_bytes, n) : 0;
}

GMSM_EXPORT int gmsm_default_plan(int group, size_t n, unsigned *c, unsigned *nwin, unsigned *entries_per_point, unsigned *fused) {
    VT_OR_FAIL(group);
    unsigned a = 0, b = 0, e = 1, f = 0;
    vt->plan_info(n, &a, &b, &e, &f);
    if (c) *c = a;
    if (nwin) *nwin = b;
    if (entries_per_point) *entries_per_point = e;
    if (fused) *fused = f;
    return GMSM_OK;
}

GMSM_EXPORT unsigned gmsm_num_windows(int group, unsigned c) {
    const GroupVTable *vt = vtable(group);
    return (vt && c) ? num_windows(vt->fr_bits, c) : 0;
}

GMSM_EXPORT int gmsm_window_sums_device(int group, const void *d_points, const void *d_scalars, size_t n, unsigned c,
                                        unsigned win_first, unsigned win_stride, void *hip_stream, uint64_t *out_xyzz) {
    VT_OR_FAIL(group);
    if (c < 2 || c > 20) return fail(GMSM_ERR_ARG, "c out of range (2..20)");
    Context *ctx;
    int rc = get_context_of_pointer(d_scalars, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    return vt->window_sums(*ctx, d_points, d_scalars, n, c, win_first, win_stride, (hipStream_t)hip_stream, out_xyzz, nullptr);
}

GMSM_EXPORT int gmsm_window_sums_enqueue(int group, const void *d_points, uint64_t bases_handle, const void *d_scalars,
                                         size_t n, unsigned c, unsigned win_first, unsigned win_stride, void *hip_stream,
                                         void *d_out_xyzz) {
    VT_OR_FAIL(group);
    if (c < 2 || c > 20) return fail(GMSM_ERR_ARG, "c out of range (2..20)");
    if (!d_out_xyzz) return fail(GMSM_ERR_ARG, "d_out_xyzz is null");
    BasesRef rb;
    if (bases_handle) {
        rb = lookup_bases(bases_handle);
        if (!rb || rb->group != group) return fail(GMSM_ERR_ARG, "unknown bases handle");
        if (n > rb->n) return fail(GMSM_ERR_LEN, "len(points) != len(scalars)");
    }
    Context *ctx;
    int rc = rb ? get_context_for(rb->device, &ctx) : get_context_of_pointer(d_out_xyzz, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    // hip_stream is used as given: NULL is the device's default (null) stream, which is what the consumer of d_out_xyzz
    // is ordered against when it runs there too - not the engine's private stream.
    return vt->window_sums_enqueue(*ctx, d_points, d_scalars, n, c, win_first, win_stride, (hipStream_t)hip_stream,
                                   d_out_xyzz, rb);  // the workspace keeps the reference while the work is in flight
}

GMSM_EXPORT int gmsm_fold_window_sets(int group, unsigned c, const uint64_t *xyzz_sets, unsigned nsets, uint64_t *out_jac) {
    VT_OR_FAIL(group);
    if (c < 2 || c > 24) return fail(GMSM_ERR_ARG, "c out of range");
    if (nsets == 0) return fail(GMSM_ERR_ARG, "nsets must be >= 1");
    vt->fold_sets(xyzz_sets, nsets, c, out_jac);
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_fold_windows(int group, unsigned c, const uint64_t *xyzz_windows, uint64_t *out_jac) {
    VT_OR_FAIL(group);
    if (c < 2 || c > 24) return fail(GMSM_ERR_ARG, "c out of range");
    vt->fold(xyzz_windows, c, out_jac);
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_jac_to_affine(int group, const uint64_t *jac, uint64_t *out_affine) {
    VT_OR_FAIL(group);
    vt->jac_to_affine(jac, out_affine);
    return GMSM_OK;
}

GMSM_EXPORT size_t gmsm_affine_limbs(int group) {
    const GroupVTable *vt = vtable(group);
    return vt ? vt->aff_bytes / 8 : 0;
}

GMSM_EXPORT size_t gmsm_scalar_limbs(int group) {
    const GroupVTable *vt = vtable(group);
    return vt ? vt->scalar_bytes / 8 : 0;
}

GMSM_EXPORT int gmsm_debug_decompose(int group, const uint64_t *scalars, size_t n, unsigned c, uint32_t *out_digits) {
    VT_OR_FAIL(group);
    return vt->debug_decompose(scalars, n, c, out_digits);
}

GMSM_EXPORT int gmsm_debug_glv_split(int group, const uint64_t *scalars, size_t n, uint32_t *out) {
    VT_OR_FAIL(group);
    return vt->debug_glv_split(scalars, n, out);
}

GMSM_EXPORT int gmsm_debug_field_op(int group, int field, int op, const uint64_t *a, const uint64_t *b, size_t count,
                                    uint64_t *out) {
    VT_OR_FAIL(group);
    if (count == 0) return GMSM_OK;
    return vt->debug_field_op(field, op, a, b, count, out);
}

GMSM_EXPORT int gmsm_debug_group_op(int group, int op, const uint64_t *acc, const uint64_t *other, size_t count,
                                    uint64_t *out) {
    VT_OR_FAIL(group);
    if (count == 0) return GMSM_OK;
    return vt->debug_group_op(op, acc, other, count, out);
}

GMSM_EXPORT int gmsm_generate_points(int group, const uint64_t *base_affine, const uint64_t *k0, const uint64_t *k1,
                                     int klimbs, size_t n, int nthreads, uint64_t *out_points) {
    VT_OR_FAIL(group);
    if (klimbs < 1 || klimbs > 16) return fail(GMSM_ERR_ARG, "klimbs out of range");
    if (n) vt->generate_points(base_affine, k0, k1, klimbs, n, nthreads, out_points);
    return GMSM_OK;
}

GMSM_EXPORT void gmsm_set_profiling(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_profiling.store(on == 2 ? 2 : on != 0 ? 1 : 0);
    for (int i = 0; i < STAGE_COUNT; ++i) {
        g_stage_ms[i] = 0;
        g_stage_launches[i] = 0;
    }
    g_stage_calls = 0;
}

GMSM_EXPORT int gmsm_get_stage_times(double *out_ms, int max_stages, unsigned long *out_calls) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = max_stages < (int)STAGE_COUNT ? max_stages : (int)STAGE_COUNT;
    for (int i = 0; i < n; ++i) out_ms[i] = g_stage_ms[i];
    if (out_calls) *out_calls = g_stage_calls;
    return n;
}

GMSM_EXPORT int gmsm_get_stage_launches(unsigned long *out_launches, int max_stages) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = max_stages < (int)STAGE_COUNT ? max_stages : (int)STAGE_COUNT;
    for (int i = 0; i < n; ++i) out_launches[i] = g_stage_launches[i];
    return n;
}

GMSM_EXPORT int gmsm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

GMSM_EXPORT int gmsm_set_device(int device) {
    int n = gmsm_device_count();
    if (device < 0 || device >= n) return fail(GMSM_ERR_DEVICE, "gmsm_set_device: no such device");
    g_device = device;
    g_default_device.store(device);
    g_pinned.store(true);
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_set_devices(const int *devices, int count) {
    std::vector<int> list;
    if (count > 0) {
        if (!devices) return fail(GMSM_ERR_ARG, "gmsm_set_devices: devices is null");
        const int n = gmsm_device_count();
        for (int i = 0; i < count; ++i) {
            if (devices[i] < 0 || devices[i] >= n)
                return fail(GMSM_ERR_DEVICE, "gmsm_set_devices: device " + std::to_string(devices[i]) + " does not exist (" +
                                                 std::to_string(n) + " visible)");
            list.push_back(devices[i]);
        }
    }
    std::lock_guard<std::mutex> lk(g_devices_mu);
    g_devices = list;
    g_devices_explicit = count > 0;
    if (count > 0) g_default_device.store(list[0]);  // the single-device entries follow the first of them
    else g_pinned.store(false);                      // count = 0: back to the default (GMSM_DEVICES, else no spreading)
    return GMSM_OK;
}

GMSM_EXPORT int gmsm_get_devices(int *out_devices, int max_devices) {
    std::vector<int> list;
    if (shard_devices(list) != GMSM_OK) return -1;  // malformed GMSM_DEVICES: gmsm_last_error() has the text
    if (list.empty()) {  // nothing configured (or pinned): the calling thread's device
        if (out_devices && max_devices > 0) out_devices[0] = g_device >= 0 ? g_device : g_default_device.load();
        return 1;
    }
    if (out_devices)
        for (int i = 0; i < (int)list.size() && i < max_devices; ++i) out_devices[i] = list[i];
    return (int)list.size();
}

// ------------------------------------------------------------------ switches and lifecycle
GMSM_EXPORT int gmsm_set_option(int key, unsigned value) {
    Options &o = options();
    switch (key) {
        case GMSM_OPT_WINDOW_BITS:
            if (value != 0 && (value < 2 || value > 20)) return fail(GMSM_ERR_ARG, "GMSM_OPT_WINDOW_BITS: 0 (the measured table) or 2..20");
            o.window_bits.store(value);
            return GMSM_OK;
        case GMSM_OPT_TABLES:
            if (value > 2) return fail(GMSM_ERR_ARG, "GMSM_OPT_TABLES: 0 never, 1 the measured call sizes, 2 every call size");
            o.tables.store(value);
            return GMSM_OK;
        case GMSM_OPT_MAX_RUN: o.max_run.store(value); return GMSM_OK;
        case GMSM_OPT_HOST_RANGES: o.host_ranges.store(value); return GMSM_OK;
        case GMSM_OPT_FIXED_BASE_BITS:
            if (value != 0 && (value < 2 || value > 14)) return fail(GMSM_ERR_ARG, "GMSM_OPT_FIXED_BASE_BITS: 0 (by batch size) or 2..14");
            o.fixed_base_bits.store(value);
            return GMSM_OK;
        case GMSM_OPT_SMALL_BITS:
            if (value > 7) return fail(GMSM_ERR_ARG, "GMSM_OPT_SMALL_BITS: 0 (on, width by size), 1 (off) or 2..7 (on, forced width)");
            o.small_bits.store(value);
            return GMSM_OK;
        case GMSM_OPT_SMALL_MAX: o.small_max.store(value); return GMSM_OK;
        case GMSM_OPT_SPLIT: o.split.store(value ? 1 : 0); return GMSM_OK;
        case GMSM_OPT_GLV:
            if (value > 2) return fail(GMSM_ERR_ARG, "GMSM_OPT_GLV: 0 never, 1 where measured ahead, 2 every unregistered call");
            o.glv.store(value);
            return GMSM_OK;
        case GMSM_OPT_SMALL_QUAD:
            if (value > 2) return fail(GMSM_ERR_ARG, "GMSM_OPT_SMALL_QUAD: 0 by call size, 1 never, 2 always");
            o.small_quad.store(value);
            return GMSM_OK;
        case GMSM_OPT_SPIN_WAIT_US:
            if (value > 1000000) return fail(GMSM_ERR_ARG, "GMSM_OPT_SPIN_WAIT_US: at most 1000000");
            o.spin_wait_us.store(value);
            return GMSM_OK;
        default: return fail(GMSM_ERR_ARG, "gmsm_set_option: unknown key");
    }
}

GMSM_EXPORT unsigned gmsm_get_option(int key) {
    Options &o = options();
    switch (key) {
        case GMSM_OPT_WINDOW_BITS: return o.window_bits.load();
        case GMSM_OPT_TABLES: return o.tables.load();
        case GMSM_OPT_MAX_RUN: return o.max_run.load();
        case GMSM_OPT_HOST_RANGES: return o.host_ranges.load();
        case GMSM_OPT_FIXED_BASE_BITS: return o.fixed_base_bits.load();
        case GMSM_OPT_SPIN_WAIT_US: return o.spin_wait_us.load();
        case GMSM_OPT_SMALL_BITS: return o.small_bits.load();
        case GMSM_OPT_SMALL_MAX: return o.small_max.load();
        case GMSM_OPT_SPLIT: return o.split.load();
        case GMSM_OPT_GLV: return o.glv.load();
        case GMSM_OPT_SMALL_QUAD: return o.small_quad.load();
        default: return 0;
    }
}

// Scratch of the idle workspaces back to the device (the reference's buffers are per call and garbage-collected,
// ecc/bn254/multiexp.go:148-176; ours are grow-only so that a steady stream of calls never allocates - one 2^26 call
// would otherwise pin tens of GB for the life of the process). Workspaces that are leased right now are left alone.
GMSM_EXPORT int gmsm_trim(size_t keep_bytes, size_t *out_freed) {
    if (out_freed) *out_freed = 0;
    std::vector<Context *> ctxs;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        ctxs = g_ctx;
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    size_t freed = 0;
    for (Context *c : ctxs) {
        if (!c || !c->live) continue;
        (void)hipSetDevice(c->device);
        // one workspace at a time - lease, drain, trim, release - so that a submit or a blocking entry arriving meanwhile
        // always finds the other workspaces (holding all of them at once turned "trim under load" into spurious
        // "already in flight" errors for unrelated callers)
        for (int i = 0; i < Context::NUM_WS; ++i) {
            Workspace *w = c->acquire_this(i);
            if (!w) continue;  // leased by a running call or a ticket: its scratch is in use
            // an enqueue-only call (gmsm_window_sums_enqueue) may have left work on the caller's stream
            if (w->last_use) (void)hipEventSynchronize(w->last_use);
            if (w->stream) (void)hipStreamSynchronize(w->stream);
            if (w->mstream) (void)hipStreamSynchronize(w->mstream);
            if (w->cstream) (void)hipStreamSynchronize(w->cstream);
            w->conv_pending = false;
            std::shared_ptr<ResidentBases> parked;
            parked.swap(w->bases_ref);
            freed += w->trim(keep_bytes);
            c->release(w);
        }
    }
    (void)hipSetDevice(prev);
    if (out_freed) *out_freed = freed;
    return GMSM_OK;
}

// Everything back: registered bases, FFT domains, workspaces, streams, events, contexts. The library is usable again
// afterwards (contexts reappear on first use). No other call may be running or start while this one runs; outstanding
// tickets are refused (collect them first).
GMSM_EXPORT int gmsm_shutdown(void) {
    std::vector<Context *> ctxs;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        ctxs = g_ctx;
    }
    // Every workspace of every context is leased FIRST - blocking callers still inside finish, new ones wait behind the
    // lease - and the check for uncollected tickets happens under the same lock as each lease, so no submit can slip in
    // between the check and the teardown (it used to: shutdown then waited for ever on a ticket nobody would collect).
    std::vector<std::pair<Context *, Workspace *>> held;
    for (Context *c : ctxs) {
        if (!c || !c->live) continue;
        for (int i = 0; i < Context::NUM_WS; ++i) {
            Workspace *w = c->acquire_unless_ticket(i);
            if (!w) {
                for (auto &h : held) h.first->release(h.second);
                return fail(GMSM_ERR_ARG, "gmsm_shutdown: a submitted MultiExp has not been collected");
            }
            held.emplace_back(c, w);
        }
    }
    {
        // The handle tables keep their length (a handle is index + 1: an old handle must stay unknown, not come to name
        // a later registration); the memory goes with the last reference, at the end of this block.
        std::vector<BasesRef> bases;
        std::vector<std::shared_ptr<ShardedBases>> sharded;
        {
            std::lock_guard<std::mutex> lk(g_bases_mu);
            for (auto &b : g_bases) bases.push_back(std::move(b)), b = nullptr;
            for (auto &b : g_sharded) sharded.push_back(std::move(b)), b = nullptr;
        }
        std::vector<FftRef> ffts;
        {
            std::lock_guard<std::mutex> lk(g_fft_mu);
            for (auto &d : g_fft) ffts.push_back(std::move(d)), d = nullptr;
        }
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (Context *c : ctxs) {
        if (!c || !c->live) continue;
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        for (auto &w : c->ws) w.destroy();  // all of them are in `held`
        // the leases end here: a caller that was parked on this context's condition variable (a submit or a blocking entry
        // that arrived while everything was leased) wakes into another epoch and returns GMSM_ERR_DEVICE; the Context
        // object itself stays (advisor, round 5: deleting it left such a waiter on a destroyed mutex) and init() runs
        // again on the next use
        c->retire();
    }
    (void)hipSetDevice(prev);
    return GMSM_OK;
}

GMSM_EXPORT const char *gmsm_last_error(void) {
    if (g_last_error.empty()) {  // this thread never failed: hand out the process-wide text (copied, so it stays valid)
        std::lock_guard<std::mutex> lk(g_any_error_mu);
        g_last_error = g_any_error;
    }
    return g_last_error.c_str();
}
GMSM_EXPORT const char *gmsm_version(void) { return "gmsm 0.6 (gfx950)"; }

}  // extern "C"
