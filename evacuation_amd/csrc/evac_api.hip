// Host side of the libevac C ABI (include/evac.h): argument checking, kernel dispatch by env size,
// error reporting.  No device allocation, no synchronisation: every call only enqueues work on the
// caller's stream, so the step can be captured into a hipGraph by the caller.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "evac_device.h"
#include "evac_subwave.h"
#include "evac_team.h"
#include "evac_gather.h"

namespace {

thread_local std::string g_create_error;

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

int validate(const evac_config_t* c, std::string& err) {
    if (!c) { err = "cfg is NULL"; return EVAC_ERR_INVALID_ARGUMENT; }
    if (c->number_of_pedestrians < 1 || c->number_of_pedestrians > EVAC_MAX_PEDESTRIANS) {
        err = "number_of_pedestrians must be in [1, " + std::to_string(EVAC_MAX_PEDESTRIANS) + "]";
        return EVAC_ERR_INVALID_ARGUMENT;
    }
    if (!(c->width > 0.f) || !(c->height > 0.f) || !(c->step_size > 0.f)) { err = "width, height, step_size must be > 0"; return EVAC_ERR_INVALID_ARGUMENT; }
    if (c->max_timesteps < 1) { err = "max_timesteps must be >= 1"; return EVAC_ERR_INVALID_ARGUMENT; }
    if (c->positions < EVAC_POS_ABS || c->positions > EVAC_POS_GRAV) { err = "positions must be abs|rel|grav"; return EVAC_ERR_INVALID_ARGUMENT; }
    if (c->statuses < EVAC_STAT_NO || c->statuses > EVAC_STAT_CAT) { err = "statuses must be no|ohe|cat"; return EVAC_ERR_INVALID_ARGUMENT; }
    if (c->type != EVAC_TYPE_DICT && c->type != EVAC_TYPE_BOX) { err = "type must be Dict|Box"; return EVAC_ERR_INVALID_ARGUMENT; }   // wrappers/config.py:81-82 ValueError
    if (c->positions == EVAC_POS_GRAV && c->type == EVAC_TYPE_BOX) {
        err = "positions='grav' with type='Box' is not implemented (reference wrappers/config.py:79-80 raises NotImplementedError)";
        return EVAC_ERR_UNSUPPORTED;
    }
    return EVAC_OK;
}

int64_t obs_dim_of(const evac_config_t* c) {
    const int64_t n = c->number_of_pedestrians;
    if (c->positions == EVAC_POS_GRAV) return 6;
    const int64_t sc = c->statuses == EVAC_STAT_OHE ? 4 : (c->statuses == EVAC_STAT_CAT ? 1 : 0);
    if (c->type == EVAC_TYPE_BOX) return (n + 2) * (2 + sc);
    return 4 + 2 * n + sc * n;
}

}  // namespace

struct evac_handle {
    evac_config_t cfg;
    evac::Params p;
    int device;
    bool bound;
    int sub_lanes;      // 0: one wave (or more) per env; 16 / 32: sub-wave kernels (evac_subwave.h)
    bool cells;         // workgroup-per-env kernels with the cell list (N > 64) instead of all pairs
    bool cu_wide;       // rollouts of one-wave envs in CU-wide workgroups (the batch fills every CU with 16 envs)
    bool cu_wide4;      // the same for four-wave envs (4 envs per CU-wide workgroup)
    bool default_cfg;   // the configuration the specialised rollout kernels assume (k_rollout_default_config)
    int team_k;         // rollouts of 513..1024-pedestrian envs by teams of 2 / 4 / 8 / 16 workgroups per env (0: one workgroup per env)
    int32_t* sched;     // inside the caller's workspace (evac_bind_workspace): moving[2][E] | perm[2][E], or NULL
    int sched_gen;      // rollout launches under the schedule so far (< 0: no deal yet): launch g reads perm[g & 1], leaves its loads
                        // in moving[g & 1] and deals perm[(g + 1) & 1] from moving[(g - 1) & 1] (rollout_body)
    bool team_bound;    // the workspace holds the teams' exchange areas
    int team_fit;       // -1: not checked yet; 1 / 0: the team grid fits the device at once (occupancy x CUs >= workgroups) or not
    size_t team_xchg_bytes;   // the teams' exchange area (records + tile slots), filled with 0xff (the tag of no round) before every team launch
    bool team_coop;     // EVAC_TEAM_COOP=1 (and the device supports it): team kernels are launched with hipLaunchCooperativeKernel
    int cus;            // compute units of the device
    bool team_fault;    // EVAC_TEAM_FAULT=1 (tests): launch the team grid one workgroup short
    // The teams' error word: 64 bytes of host-mapped memory owned by the handle.  A team that lost a member raises it from the
    // kernel (system-scope store); every later evac_* call of the handle reads it on the host, without synchronising.
    volatile unsigned* team_flag_host;
    unsigned* team_flag_dev;
    std::string err;
    std::string variant[4];   // evac_step | evac_rollout with one workgroup (or less) per env | evac_rollout by teams | ... as two parts
    // evac_options_t.parts = 2: the handle's rollouts go out as two half-batch kernels on two streams it owns (include/evac.h,
    // evac_join).  part[k] is a complete handle of its own over envs [k E / 2, (k + 1) E / 2) of THIS handle's buffers (state
    // pointers and workspace slices offset, Params::slab_envs = E, env_id_offset + k E / 2: the same global env ids), so a part
    // launches exactly what a handle of that size launches -- schedule, in-kernel deal and generation counter of its own.
    int n_parts;              // 1 or 2
    evac_handle* part[2];
    hipStream_t part_stream[2];
    hipEvent_t part_done[2], fork_ev;
    bool parts_pending;       // the part streams hold work the caller's stream has not been made to wait for (evac_join)
    bool forked;              // the own streams have been put behind the caller's stream since the last join (evac_rollout forks once per join)
    evac_options_t opt;       // as resolved at creation (evac_get_options)
    // evac_options_t.chain = 1: rollout launch g goes to part_stream[g & 1] and waits PER ENV for launch g - 1 on the device
    // (include/evac.h, evac_common.h ChainArgs).  The schedule is four deep here: launch g reads perm[g & 3], leaves its loads in
    // moving[g & 3] and deals perm[(g + 2) & 3] -- read by the next launch of ITS stream -- from moving[(g - 2) & 3], the last
    // launch of its stream: everything a launch reads was written by a launch its queue has completed.
    bool chain;               // (requested and possible; used only with the workspace bound)
    bool chain_bound;
    bool persist;             // evac_options_t.chain = 2: one persistent rollout kernel per join, every evac_rollout call a command of its ring
    bool persist_running;     // the kernel is resident (on part_stream[0]) and reads commands
    int persist_seq;          // the next command's index (its sequence number is index + 1; never reset)
    int persist_first;        // the first command of the kernel that is running
    bool chain_small;         // the chain's launches are the 256-thread workgroups of one-wave envs (four envs each: no deal, no pace keeping)
    int chain_gen;            // rollout launches of the chain so far = the generation the next launch waits for
    int chain_start;          // the launch at which the chain (re)started: deals begin two launches later
    bool chain_restart;       // the state was written by something else than the chain's last launch: fill the generation words first
    int32_t* chain_sched;     // moving[4][E] | perm[4][E]
    size_t chain_xchg_bytes;  // (what the pool handed out: given back with it)
    char* chain_xchg;         // [E] exchange records (evac_common.h): the state between the chain's launches
    unsigned* chain_abort;    // device word: a wait timed out
    bool chain_dirty;         // the records are ahead of the caller's state arrays (k_chain_export at the next join)
    unsigned long long chain_wgs;   // workgroups of the chain's launches enqueued since its last restart (the gate's target)
    hipEvent_t chain_ev;
};

namespace {

int fail(evac_handle_t h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

// A team rollout of this handle lost a member (evac_team.h): the outputs of that launch are void.  Sticky until
// evac_team_clear_error; the handle runs the one-workgroup-per-env kernels from then on.
int team_aborted(evac_handle_t h, const char* what) {
    if (h->team_flag_host && *h->team_flag_host != 0u && h->chain) {
        h->chain = false;
        return fail(h, EVAC_ERR_TEAM_ABORTED,
                    std::string(what) + ": a chained rollout launch waited in vain for an env's state (a launch of the chain was lost); the "
                    "outputs since are void -- call evac_team_clear_error(), then reset or restore the batch; the handle issues plain launches from now on");
    }
    if (h->team_flag_host && *h->team_flag_host != 0u) {
        h->team_k = 0;
        return fail(h, EVAC_ERR_TEAM_ABORTED,
                    std::string(what) + ": an earlier team rollout lost a member (the workgroups of a team were not resident together); "
                    "the outputs of that launch are void and the env it carried kept its pre-launch state -- call "
                    "evac_team_clear_error(), then reset or restore the batch; the handle uses one workgroup per env from now on");
    }
    return EVAC_OK;
}

int check_launch(evac_handle_t h, const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, EVAC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return EVAC_OK;
}

int waves_per_env(int n_ped) { return n_ped <= 64 ? 1 : (n_ped <= 128 ? 2 : (n_ped <= 256 ? 4 : (n_ped <= 512 ? 8 : 16))); }

// launch KERNEL<F, GRAV> for a Wave / Cells family F
#define EVAC_LAUNCH_F(h, KERNEL, F_, GRAV_, s_, ...)                                                                  \
    hipLaunchKernelGGL((evac::KERNEL<F_, GRAV_>), dim3((unsigned)(((h)->p.n_envs + F_::kEnvsPerBlock - 1) / F_::kEnvsPerBlock)), \
                       dim3(F_::kBlock), 0, s_, __VA_ARGS__)
#define EVAC_LAUNCH_SUB(h, KERNEL, G_, GRAV_, s_, ...)                                                              \
    hipLaunchKernelGGL((evac::KERNEL##_sub<G_, GRAV_>),                                                             \
                       dim3((unsigned)(((h)->p.n_envs + evac::Sub<G_>::kEnvsPerBlock - 1) / evac::Sub<G_>::kEnvsPerBlock)), \
                       dim3(evac::Sub<G_>::kBlock), 0, s_, __VA_ARGS__)
#define EVAC_LAUNCH_WPE(h, KERNEL, WPE_, s_, ...)                                                                   \
    do {                                                                                                            \
        if (WPE_ > 1 && (h)->cells) {                                                                               \
            using FC_ = evac::Cells<(WPE_ > 1 ? WPE_ : 2)>;                                                         \
            if (grav_) EVAC_LAUNCH_F(h, KERNEL, FC_, true, s_, __VA_ARGS__);                                        \
            else EVAC_LAUNCH_F(h, KERNEL, FC_, false, s_, __VA_ARGS__);                                             \
        } else {                                                                                                    \
            using FW_ = evac::Wave<WPE_>;                                                                           \
            if (grav_) EVAC_LAUNCH_F(h, KERNEL, FW_, true, s_, __VA_ARGS__);                                        \
            else EVAC_LAUNCH_F(h, KERNEL, FW_, false, s_, __VA_ARGS__);                                             \
        }                                                                                                           \
    } while (0)

// kernel variant = family (lanes or waves per env, all pairs | cell list) x (gravity observation | generic
// positions/statuses observation)
#define EVAC_DISPATCH(h, KERNEL, stream, ...)                                         \
    do {                                                                              \
        const int wpe_ = waves_per_env((h)->p.n_ped);                                 \
        const bool grav_ = (h)->p.obs_pos == EVAC_POS_GRAV;                           \
        hipStream_t s_ = (hipStream_t)(stream);                                       \
        if ((h)->sub_lanes == 16) {                                                   \
            if (grav_) EVAC_LAUNCH_SUB(h, KERNEL, 16, true, s_, __VA_ARGS__);         \
            else EVAC_LAUNCH_SUB(h, KERNEL, 16, false, s_, __VA_ARGS__);              \
        } else if ((h)->sub_lanes == 32) {                                            \
            if (grav_) EVAC_LAUNCH_SUB(h, KERNEL, 32, true, s_, __VA_ARGS__);         \
            else EVAC_LAUNCH_SUB(h, KERNEL, 32, false, s_, __VA_ARGS__);              \
        } else if (wpe_ == 1) EVAC_LAUNCH_WPE(h, KERNEL, 1, s_, __VA_ARGS__);         \
        else if (wpe_ == 2) EVAC_LAUNCH_WPE(h, KERNEL, 2, s_, __VA_ARGS__);           \
        else if (wpe_ == 4) EVAC_LAUNCH_WPE(h, KERNEL, 4, s_, __VA_ARGS__);           \
        else if (wpe_ == 8) EVAC_LAUNCH_WPE(h, KERNEL, 8, s_, __VA_ARGS__);           \
        else EVAC_LAUNCH_WPE(h, KERNEL, 16, s_, __VA_ARGS__);                         \
    } while (0)


// ---- team rollouts (evac_team.h): which kernel, how many workgroups, and do they all fit the device at once ----
// ONE team grid at a time per device, process-wide.  The members of a team wait for each other, and team_grid_fits only
// checks that ONE grid fits the device: two grids in flight (two handles on two streams -- train and eval envs --, a graph replay
// beside a live launch) can each be dispatched in part, every resident member polling in place on a CU the other grid's missing
// members need -- both time out.  So every team launch first waits (on the device, hipStreamWaitEvent) for the event the
// previous team launch on that device recorded behind itself, whatever stream or handle it came from.  A launch under stream
// capture cannot join the chain (an event recorded outside the capture); a graph holding team launches must not be replayed
// beside another team launch.  Other PROCESSES on the same GPU are out of reach: they need EVAC_TEAM=0.
constexpr int kMaxDevices = 64;
std::mutex g_team_chain_lock;
hipEvent_t g_team_chain[kMaxDevices] = {};

const void* team_kernel(const evac_handle* h) {
    const bool grav = h->p.obs_pos == EVAC_POS_GRAV, dflt = h->default_cfg;
#define EVAC_TEAM_FN(K_)                                                                                                          \
    (dflt ? (grav ? (const void*)evac::k_rollout_default_config<evac::Team<K_>, true> : (const void*)evac::k_rollout_default_config<evac::Team<K_>, false>) \
          : (grav ? (const void*)evac::k_rollout<evac::Team<K_>, true> : (const void*)evac::k_rollout<evac::Team<K_>, false>))
    return h->team_k == 16 ? EVAC_TEAM_FN(16) : (h->team_k == 8 ? EVAC_TEAM_FN(8) : (h->team_k == 4 ? EVAC_TEAM_FN(4) : EVAC_TEAM_FN(2)));
#undef EVAC_TEAM_FN
}
const void* team_persist_kernel(const evac_handle* h) {
    const bool grav = h->p.obs_pos == EVAC_POS_GRAV, dflt = h->default_cfg;
#define EVAC_TEAM_FN(K_)                                                                                                          \
    (dflt ? (grav ? (const void*)evac::k_rollout_persist_default_config<evac::Team<K_>, true> : (const void*)evac::k_rollout_persist_default_config<evac::Team<K_>, false>) \
          : (grav ? (const void*)evac::k_rollout_persist<evac::Team<K_>, true> : (const void*)evac::k_rollout_persist<evac::Team<K_>, false>))
    return h->team_k == 16 ? EVAC_TEAM_FN(16) : (h->team_k == 8 ? EVAC_TEAM_FN(8) : (h->team_k == 4 ? EVAC_TEAM_FN(4) : EVAC_TEAM_FN(2)));
#undef EVAC_TEAM_FN
}
unsigned team_grid(const evac_handle* h) {
    // (EVAC_TEAM_FAULT=1, fault injection for tests/test_gpu_team.py: the last workgroup is never launched, so the team it
    // belongs to loses a member and must time out, flag the error and leave its env's state alone)
    return (unsigned)((h->p.n_envs + 7) / 8 * 8 * h->team_k) - (h->team_fault ? 1u : 0u);
}
// The members of a team wait for each other, so every workgroup of the grid needs a CU of its own at the same time.
bool team_grid_fits(evac_handle* h) {
    if (h->team_fit < 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, team_kernel(h), 1024, 0) != hipSuccess) {
            (void)hipGetLastError();
            per_cu = 0;
        }
        h->team_fit = (long long)per_cu * h->cus >= (long long)team_grid(h) ? 1 : 0;
    }
    return h->team_fit == 1;
}

}  // namespace

extern "C" {

int evac_version(void) { return EVAC_VERSION; }

const char* evac_status_string(int s) {
    switch (s) {
        case EVAC_OK: return "ok";
        case EVAC_ERR_INVALID_ARGUMENT: return "invalid argument";
        case EVAC_ERR_NOT_BOUND: return "state buffers not bound";
        case EVAC_ERR_UNSUPPORTED: return "unsupported configuration";
        case EVAC_ERR_HIP: return "HIP error";
        case EVAC_ERR_NO_DEVICE: return "no HIP device";
        case EVAC_ERR_TEAM_ABORTED: return "a team rollout lost a member";
        default: return "unknown status";
    }
}

const char* evac_last_error(evac_handle_t h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int evac_config_validate(const evac_config_t* cfg) {
    std::string err;
    const int rc = validate(cfg, err);
    g_create_error = err;
    return rc;
}

int64_t evac_config_obs_dim(const evac_config_t* cfg) {
    std::string err;
    if (validate(cfg, err) != EVAC_OK) { g_create_error = err; return -1; }
    return obs_dim_of(cfg);
}

}  // extern "C"

namespace {
// A create-time option: the diagnostic environment variable wins when it is set (A/B runs of an unmodified caller), then the
// caller's evac_options_t field, then -1 = automatic.
int option_value(const char* env_name, int option) {
    const char* v = std::getenv(env_name);
    if (v && v[0]) return std::atoi(v);
    return option;
}

// ---- What a handle allocates from the device ITSELF -- its two streams, the chain's uncached exchange records, the host-mapped error
// word -- is taken from pools that live as long as the process and goes back to them when the handle is destroyed; nothing of it is
// ever given back to HIP.  Creating and freeing these per handle re-used device addresses (and hardware queues) at a high rate under
// whatever else the process runs, and on some MI355X boxes a process that did so read stale memory through OTHER allocations for a
// while afterwards (tests/test_gpu_parity.py's chained cases, in their first iterations only, serialised launches included:
// DESIGN.md "the pools").  A process that creates a thousand handles now makes one pair of streams.
struct PooledPair { int device; hipStream_t s[2]; };
struct PooledBuf { int device; size_t bytes; void* p; };
struct PooledWord { int device; void* host; void* dev; };
std::mutex g_pool_mu;
std::vector<PooledPair> g_pool_pairs;
std::vector<PooledBuf> g_pool_bufs;
std::vector<PooledWord> g_pool_words;
bool pools_on() { static const bool on = option_value("EVAC_DIAG_NO_POOLS", 0) == 0; return on; }

// 64 bytes of host memory the device can write (the error word of the team kernels and of the chain), zeroed
bool take_error_word(int device, void** host, void** dev) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool_words.size(); ++i)
            if (g_pool_words[i].device == device) {
                *host = g_pool_words[i].host; *dev = g_pool_words[i].dev;
                g_pool_words.erase(g_pool_words.begin() + (long)i);
                std::memset(*host, 0, 64);
                return true;
            }
    }
    *host = *dev = nullptr;
    if (hipHostMalloc(host, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(dev, *host, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (*host) (void)hipHostFree(*host);
        *host = *dev = nullptr;
        return false;
    }
    std::memset(*host, 0, 64);
    return true;
}
void give_error_word(int device, void* host, void* dev) {
    if (!host) return;
    if (!pools_on()) { (void)hipHostFree(host); return; }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_words.push_back(PooledWord{device, host, dev});
}
// uncached device memory of at least `bytes` (the caller fills it)
void* take_uncached(int device, size_t bytes, size_t* got) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t best = g_pool_bufs.size();
        for (size_t i = 0; i < g_pool_bufs.size(); ++i)
            if (g_pool_bufs[i].device == device && g_pool_bufs[i].bytes >= bytes && g_pool_bufs[i].bytes <= 2 * bytes + (1u << 16) &&
                (best == g_pool_bufs.size() || g_pool_bufs[i].bytes < g_pool_bufs[best].bytes)) best = i;
        if (best < g_pool_bufs.size()) {
            void* p = g_pool_bufs[best].p;
            *got = g_pool_bufs[best].bytes;
            g_pool_bufs.erase(g_pool_bufs.begin() + (long)best);
            return p;
        }
    }
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *got = bytes;
    return p;
}
void give_uncached(int device, void* p, size_t bytes) {
    if (!p) return;
    if (!pools_on()) { (void)hipFree(p); return; }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_bufs.push_back(PooledBuf{device, bytes, p});
}

int create_impl(const evac_config_t* cfg, int32_t num_envs, int32_t device, uint64_t seed, uint64_t env_id_offset,
                const evac_options_t& o, evac_handle_t* out) {
    if (!out) { g_create_error = "out is NULL"; return EVAC_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    std::string err;
    const int rc = validate(cfg, err);
    if (rc != EVAC_OK) { g_create_error = err; return rc; }
    if (num_envs < 1) { g_create_error = "num_envs must be >= 1"; return EVAC_ERR_INVALID_ARGUMENT; }
    if (env_id_offset + (uint64_t)num_envs > 0xffffffffull) { g_create_error = "env_id_offset + num_envs exceeds 2^32"; return EVAC_ERR_INVALID_ARGUMENT; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        g_create_error = "no HIP device visible: libevac has no CPU path";
        return EVAC_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) { g_create_error = "device index out of range"; return EVAC_ERR_INVALID_ARGUMENT; }
    evac_handle* h = new (std::nothrow) evac_handle();
    if (!h) { g_create_error = "out of host memory"; return EVAC_ERR_INVALID_ARGUMENT; }
    h->cfg = *cfg;
    h->device = device;
    h->bound = false;
    h->n_parts = 1;
    h->part[0] = h->part[1] = nullptr;
    h->part_stream[0] = h->part_stream[1] = nullptr;
    h->part_done[0] = h->part_done[1] = h->fork_ev = nullptr;
    h->parts_pending = false;
    h->forked = false;
    h->chain = h->chain_bound = h->chain_small = false;
    h->persist = h->persist_running = false;
    h->persist_seq = h->persist_first = 0;
    h->chain_gen = h->chain_start = 1;          // (never 0: a zero-filled workspace must not look like a published generation)
    h->chain_restart = true;
    h->chain_sched = nullptr;
    h->chain_xchg = nullptr;
    h->chain_xchg_bytes = 0;
    h->chain_abort = nullptr;
    h->chain_dirty = false;
    h->chain_wgs = 0;
    h->chain_ev = nullptr;
    const int o_subwave = option_value("EVAC_SUBWAVE", o.subwave), o_cells = option_value("EVAC_CELLS", o.cells);
    const int o_cu_wide = option_value("EVAC_CU_WIDE", o.cu_wide), o_team = option_value("EVAC_TEAM", o.team);
    const int o_specialize = option_value("EVAC_SPECIALIZE", o.specialize);
    const int o_team_coop = option_value("EVAC_TEAM_COOP", o.team_coop), o_team_fault = option_value("EVAC_TEAM_FAULT", o.team_fault);
    {   // small rooms share a wave: 4 envs per wave for N <= 16, 2 for N <= 32 (subwave = 0 disables, for A/B tests)
        const bool allow = o_subwave != 0;
        const int n = cfg->number_of_pedestrians;
        h->sub_lanes = !allow ? 0 : (n <= 16 ? 16 : (n <= 32 ? 32 : 0));
        // rooms of more than 512 pedestrians use the cell list; cells = 1 / 0 forces it on (for every room of more
        // than one wave) / off, for A/B tests
        h->cells = n > evac::kWave && (o_cells == 1 ? true : (o_cells == 0 ? false : n > 512));
    }
    evac::Params& p = h->p;
    std::memset(&p, 0, sizeof(p));
    p.n_envs = num_envs;
    p.slab_envs = num_envs;
    p.n_ped = cfg->number_of_pedestrians;
    p.width = cfg->width;
    p.height = cfg->height;
    p.step_size = cfg->step_size;
    p.noise_coef = cfg->noise_coef;
    p.eps = cfg->eps;
    p.ens = cfg->enslaving_degree;
    p.one_minus_ens = (float)(1.0 - (double)cfg->enslaving_degree);   // area.py:141 computes (1. - e) in double
    p.init_reward = cfg->init_reward_each_step;
    p.intrinsic_coef = cfg->intrinsic_reward_coef;
    p.flags = (cfg->is_new_exiting_reward ? evac::kFlagNewExitingReward : 0u) |
              (cfg->is_new_followers_reward ? evac::kFlagNewFollowersReward : 0u) |
              (cfg->is_termination_agent_wall_collision ? evac::kFlagTermOnWall : 0u) |
              (cfg->nan_guard ? evac::kFlagNanGuard : 0u) | (cfg->clip_action ? evac::kFlagClipAction : 0u);
    p.max_timesteps = cfg->max_timesteps;
    p.inv_n = (float)(1.0 / (double)cfg->number_of_pedestrians);
    p.inv_200n = (float)(1.0 / (200.0 * (double)cfg->number_of_pedestrians));
    p.obs_pos = cfg->positions;
    p.obs_stat = cfg->statuses;
    p.obs_box = cfg->type == EVAC_TYPE_BOX;
    p.obs_dim = (int32_t)obs_dim_of(cfg);
    p.alpha = cfg->alpha;
    p.neg_alpha = -cfg->alpha;
    p.grav_pow = cfg->alpha + 2.0f;
    const float gp = cfg->alpha + 2.0f;
    p.grav_pow_int = (gp == std::floor(gp) && gp >= 1.0f && gp <= 63.0f) ? (int)gp : 0;
    {
        const float half = std::fabs(cfg->noise_coef) * 0.5f;
        p.small_noise = half <= 0.2f ? 2 : (half <= 0.78539816f ? 1 : 0);
    }
    {   // cell list: 16 x 16 cells over [-max(W,1), max(W,1)] x [-max(H,1), max(H,1)] (reset draws positions in +-1
        // whatever the room, pedestrians.py:17): a cell is >= 0.125 wide, the pedestrian radius is 0.1
        const float wb = std::fmax(cfg->width, 1.0f), hb = std::fmax(cfg->height, 1.0f);
        p.cell_ox = wb;
        p.cell_oy = hb;
        p.cell_inv_hx = (float)evac::kCellsX / (2.0f * wb);
        p.cell_inv_hy = (float)evac::kCellsY / (2.0f * hb);
        // integer heading = rint(heading * head_scale): it must fit v_mad_i32_i24 (< 2^23) and N of them an int32
        // (a unit-heading component can exceed 1 by a few ulp of v_rsq: 16 units of slack cover it)
        const int64_t cap = (int64_t)0x7fffffff / cfg->number_of_pedestrians - 16;
        p.head_scale = (float)(cap < 0x7ffff0 ? cap : (int64_t)0x7ffff0);     // exact in f32 (< 2^24)
        // Rooms of 513..1024 pedestrians (team kernels and Cells<16>, which agree bit for bit): the team kernels sum up to
        // kTeamExactBatch = 8 such headings at a time in packed f32 arithmetic before the partial sum goes to an integer accumulator
        // (evac_team.h) -- exact below 2^24, so the scale is capped at 2^21 - 16 there (N = 1024 has 2^21 - 17 anyway).
        if (cfg->number_of_pedestrians > 512 && p.head_scale > (float)((1 << 24) / evac::kTeamExactBatch - 16))
            p.head_scale = (float)((1 << 24) / evac::kTeamExactBatch - 16);
    }
    {   // CU-wide rollout workgroups pay off once every CU gets its 16 one-wave envs, and as long as the launch is a few
        // rounds deep: a 16-wave workgroup needs a whole CU, so in a long launch every CU idles while the last waves of its
        // workgroup finish (524 288 envs: 1.44e9 against 1.62e9 env-steps/s with 4-wave workgroups; 8 192 envs: 1.75e9
        // against 1.59e9).  EVAC_CU_WIDE=1 / 0 forces, for tests.
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        h->cus = cus;
        const bool one_wave = h->sub_lanes == 0 && cfg->number_of_pedestrians <= evac::kWave;
        h->cu_wide = one_wave && (o_cu_wide == 1 ? true : (o_cu_wide == 0 ? false : (num_envs >= 16 * cus && num_envs <= 64 * cus)));
        // priority rotation (rollout_body) evens out the waves of a SIMD in launches of one or two rounds; deeper launches
        // even out by themselves and run ~2 % faster without it
        const int wpe = waves_per_env(cfg->number_of_pedestrians);
        const long long waves = h->sub_lanes ? ((long long)num_envs * h->sub_lanes + 63) / 64 : (long long)num_envs * wpe;
        h->p.fair = waves <= 2ll * 16 * cus ? 1 : 0;
        // four-wave envs (N = 129..256, all pairs): 4 envs per CU-wide workgroup with per-env LDS barriers, pace and schedule
        h->cu_wide4 = h->sub_lanes == 0 && wpe == 4 && !h->cells &&
                      (o_cu_wide == 1 ? true : (o_cu_wide == 0 ? false : (num_envs >= 4 * cus && num_envs <= 16 * cus)));
        h->default_cfg = o_specialize != 0;                  // specialize = 0: always the generic kernels (A/B runs)
        const bool obs_default = cfg->positions == EVAC_POS_GRAV
                                     ? p.grav_pow_int == 5
                                     : (cfg->positions == EVAC_POS_REL && cfg->statuses == EVAC_STAT_OHE && cfg->type == EVAC_TYPE_BOX);
        h->default_cfg = h->default_cfg && obs_default && p.small_noise == 2 && p.ens == 1.0f &&
                         p.one_minus_ens == 0.0f &&
                         (p.flags & (evac::kFlagTermOnWall | evac::kFlagNanGuard)) == 0;
        h->sched = nullptr;
        h->sched_gen = -1;
        // teams: as many CUs per env as the batch leaves free -- all members must be resident together (one 1024-thread
        // workgroup per CU), teams are laid out in rows of 8 (one per XCD).  EVAC_TEAM=0 disables, 2 / 4 / 8 / 16 forces a size.
        h->team_k = 0;
        h->team_bound = false;
        h->team_fit = -1;
        h->team_coop = false;
        h->team_fault = false;
        h->team_flag_host = nullptr;
        h->team_flag_dev = nullptr;
        if (cfg->number_of_pedestrians > 512) {
            const int want = o_team;
            const int rows = (num_envs + 7) / 8 * 8;
            for (int k = 16; k >= 2; k >>= 1)
                if ((want < 0 || want == k) && rows * k <= cus) { h->team_k = k; break; }
            if (want == 0 || (want < 0 && o_cells >= 0)) h->team_k = 0;   // an A/B run of the one-workgroup families
        }
        if (h->team_k) {
            DeviceGuard g(device);
            void* host = nullptr;
            void* dev = nullptr;
            if (!take_error_word(device, &host, &dev)) {
                h->team_k = 0;                       // no error word, no teams
            } else {
                h->team_flag_host = (volatile unsigned*)host;
                h->team_flag_dev = (unsigned*)dev;
                int coop = 0;
                if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device) != hipSuccess) coop = 0;
                h->team_coop = coop != 0 && o_team_coop == 1;        // opt-in: see evac_rollout
                h->team_fault = o_team_fault == 1;
            }
        }
    }
    p.seed_lo = (uint32_t)(seed & 0xffffffffull);
    p.seed_hi = (uint32_t)(seed >> 32);
    p.env_id_offset = (uint32_t)env_id_offset;
    {
        const int wpe = waves_per_env(p.n_ped);
        const bool grav = p.obs_pos == EVAC_POS_GRAV;
        std::string fam = h->sub_lanes == 16 ? evac::Sub<16>::kName : h->sub_lanes == 32 ? evac::Sub<32>::kName :
                          wpe == 1 ? evac::Wave<1>::kName :
                          h->cells ? (wpe == 2 ? evac::Cells<2>::kName : wpe == 4 ? evac::Cells<4>::kName : wpe == 8 ? evac::Cells<8>::kName : evac::Cells<16>::kName)
                                   : (wpe == 2 ? evac::Wave<2>::kName : wpe == 4 ? evac::Wave<4>::kName : wpe == 8 ? evac::Wave<8>::kName : evac::Wave<16>::kName);
        const char* kind = h->default_cfg ? "_default_config<" : "<";     // (the names rocprofv3 shows)
        h->variant[0] = std::string("k_step") + kind + fam + (grav ? ", grav obs>" : ", generic obs>");
        if (h->cu_wide) fam = evac::Wave<1, 1024>::kName;
        if (h->cu_wide4) fam = evac::Wave<4, 1024>::kName;
        h->variant[1] = std::string("k_rollout") + kind + fam + (grav ? ", grav obs>" : ", generic obs>");
        if (h->team_k) fam = h->team_k == 16 ? evac::Team<16>::kName : (h->team_k == 8 ? evac::Team<8>::kName : (h->team_k == 4 ? evac::Team<4>::kName : evac::Team<2>::kName));
        h->variant[2] = std::string("k_rollout") + kind + fam + (grav ? ", grav obs>" : ", generic obs>");
    }
    if (!h->team_k) h->team_fault = o_team_fault == 1;       // (chained launches: fault injection of their own, evac_rollout)
    h->opt = evac_options_t{h->sub_lanes ? 1 : 0, h->cells ? 1 : 0, (h->cu_wide || h->cu_wide4) ? 1 : 0, h->team_k, h->default_cfg ? 1 : 0,
                            1, h->team_coop ? 1 : 0, h->team_fault ? 1 : 0, 0};
    *out = h;
    return EVAC_OK;
}

// ---- evac_options_t.parts = 2: two streams of the handle's own that really run BESIDE each other.  HIP deals its streams onto a
// handful of hardware queues (four by default) in the order of their first use, and two streams that share a queue execute
// strictly one after the other (round 4 found the gather stream on the rollout's queue).  So every candidate is timed
// against the first stream with two one-thread spin kernels of ~100 us: a pair that takes about as long as one of them overlaps.
__global__ void k_spin_100us(unsigned long long ticks, unsigned* sink) {
    unsigned long long t0, t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    unsigned n = 0;
    do {                                           // (100 MHz constant clock; bounded whatever the clock reads)
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    } while (t - t0 < ticks && ++n < (1u << 22));
    if (sink) *sink = n;
}
float spin_pair_ms(hipStream_t a, hipStream_t b, hipEvent_t e0, hipEvent_t e1, hipEvent_t eb) {
    float ms = -1.0f;
    (void)hipEventRecord(e0, a);
    hipLaunchKernelGGL(k_spin_100us, dim3(1), dim3(1), 0, a, 10000ull, (unsigned*)nullptr);
    if (b) {
        hipLaunchKernelGGL(k_spin_100us, dim3(1), dim3(1), 0, b, 10000ull, (unsigned*)nullptr);
        (void)hipEventRecord(eb, b);
        (void)hipStreamWaitEvent(a, eb, 0);
    }
    (void)hipEventRecord(e1, a);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { (void)hipGetLastError(); return -1.0f; }
    return ms;
}
bool make_part_streams(evac_handle* h) {
    DeviceGuard g(h->device);
    hipEvent_t e0 = nullptr, e1 = nullptr, eb = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool_pairs.size(); ++i)
            if (g_pool_pairs[i].device == h->device) {        // a pair an earlier handle found to overlap, idle since that handle was destroyed
                h->part_stream[0] = g_pool_pairs[i].s[0];
                h->part_stream[1] = g_pool_pairs[i].s[1];
                g_pool_pairs.erase(g_pool_pairs.begin() + (long)i);
                break;
            }
    }
    if (h->part_stream[0]) {
        bool ok = true;
        for (int k = 0; ok && k < 2; ++k) ok = hipEventCreateWithFlags(&h->part_done[k], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        return ok;
    }
    bool ok = hipStreamCreateWithFlags(&h->part_stream[0], hipStreamNonBlocking) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
              hipEventCreate(&e1) == hipSuccess && hipEventCreate(&eb) == hipSuccess;
    if (ok) {
        (void)spin_pair_ms(h->part_stream[0], nullptr, e0, e1, eb);                 // (first launch: the code object is loaded)
        const float one = spin_pair_ms(h->part_stream[0], nullptr, e0, e1, eb);
        hipStream_t dropped[8];
        int n_dropped = 0;
        for (int c = 0; ok && c < 8; ++c) {
            hipStream_t cand = nullptr;
            if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) { ok = false; break; }
            (void)spin_pair_ms(h->part_stream[0], cand, e0, e1, eb);                // (first use: the stream gets its queue here)
            const float pair = spin_pair_ms(h->part_stream[0], cand, e0, e1, eb);
            if (c == 7 || (one > 0.0f && pair > 0.0f && pair < 1.5f * one)) { h->part_stream[1] = cand; break; }
            dropped[n_dropped++] = cand;          // (kept until the search ends: its queue assignment stays used up, which moves the next candidate on)
        }
        for (int k = 0; k < n_dropped; ++k) (void)hipStreamDestroy(dropped[k]);
        ok = ok && h->part_stream[1] != nullptr;
    }
    for (int k = 0; ok && k < 2; ++k) ok = hipEventCreateWithFlags(&h->part_done[k], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming) == hipSuccess;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (eb) (void)hipEventDestroy(eb);
    if (!ok) (void)hipGetLastError();
    return ok;
}
// a command for the resident kernel: the payload, a store fence, then the sequence number in the same 64-byte segment (through the BAR)
void post_command(evac_handle* h, int n_steps, const void* slab, const void* stats, const void* actions) {
    volatile evac::PersistCmd* c = (volatile evac::PersistCmd*)(h->chain_xchg + (size_t)(h->persist_seq & (evac::kPersistRing - 1)) * 64);
    c->slab = (unsigned long long)(uintptr_t)slab;
    c->stats = (unsigned long long)(uintptr_t)stats;
    c->actions = (unsigned long long)(uintptr_t)actions;
    c->n_steps = n_steps;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    c->seq = (unsigned)h->persist_seq + 1u;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    h->persist_seq += 1;
}
void destroy_parts(evac_handle* h) {
    DeviceGuard g(h->device);
    if (h->persist && h->persist_running) {            // (a handle destroyed without a join: the resident kernel is told to end)
        post_command(h, 0, nullptr, nullptr, nullptr);     //  -- and would leave by itself for lack of commands anyway
        h->persist_running = false;
    }
    for (int k = 0; k < 2; ++k)
        if (h->part_stream[k]) (void)hipStreamSynchronize(h->part_stream[k]);
    if (h->part_stream[0] && h->part_stream[1] && pools_on()) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_pool_pairs.push_back(PooledPair{h->device, {h->part_stream[0], h->part_stream[1]}});
        h->part_stream[0] = h->part_stream[1] = nullptr;
    }
    for (int k = 0; k < 2; ++k) {
        if (h->part_stream[k]) (void)hipStreamDestroy(h->part_stream[k]);
        if (h->part_done[k]) (void)hipEventDestroy(h->part_done[k]);
        if (h->part[k]) {
            give_error_word(h->device, (void*)h->part[k]->team_flag_host, (void*)h->part[k]->team_flag_dev);
            delete h->part[k];
        }
        h->part_stream[k] = nullptr; h->part_done[k] = nullptr; h->part[k] = nullptr;
    }
    if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
    if (h->chain_ev) (void)hipEventDestroy(h->chain_ev);
    give_uncached(h->device, h->chain_xchg, h->chain_xchg_bytes);
    h->chain_xchg = nullptr;
    h->fork_ev = h->chain_ev = nullptr;
    h->chain = false;
    h->persist = false;
    h->n_parts = 1;
    h->parts_pending = false;
}
// `stream` waits for everything the part streams have been given so far (evac_join; implied by every call that is not a plain rollout)
void deal_now(evac_handle_t h, hipStream_t s, bool both);
// The persistent rollout kernel on the handle's stream.  fresh: the first kernel after a join -- every env starts at command
// h->persist_seq, workgroup 0 deals the next kernel's envs as it starts; resume: every env at the command it had reached when the kernel
// before left; stop_at: the index of a STOP command that is already in the ring (the finisher of a join), else INT_MAX.
int launch_persistent(evac_handle* h, int resume, int stop_at, bool fresh) {
    using FW = evac::Wave<1, 1024>;
    using FW4 = evac::Wave<4, 1024>;
    hipStream_t S = h->part_stream[0];
    const int E = h->p.n_envs;
    if (h->sched && h->sched_gen < 0) deal_now(h, S, true);
    const int g_ = h->sched_gen;
    const bool dealt = h->sched && g_ >= 0;
    const bool deals = dealt && fresh;
    const int32_t* perm = dealt ? h->sched + (2 + (g_ & 1)) * E : nullptr;
    int32_t* moving = h->sched ? h->sched + (dealt ? (g_ & 1) : 0) * E : nullptr;
    const int32_t* deal_loads = deals ? h->sched + ((g_ + 1) & 1) * E : nullptr;       // (workgroup 0 deals the NEXT fresh kernel's envs as it starts)
    int32_t* deal_perm = deals ? h->sched + (2 + ((g_ + 1) & 1)) * E : nullptr;
    if (deals) h->sched_gen = g_ + 1;
    evac::ChainArgs ca{h->chain_xchg, h->persist_seq, nullptr, nullptr, 0, nullptr, resume, stop_at};
#define EVAC_PERSIST_ARGS h->p, (const int*)perm, (int*)moving, (const int*)deal_loads, (int*)deal_perm, ca
    if (h->team_k) {
        // a team grid (evac_team.h): the exchange area starts with the tag of no round, the teams' verdicts at zero; team grids of one
        // device take turns (g_team_chain), whatever handle or stream they come from
        int32_t* decision = (int32_t*)(h->chain_xchg + (size_t)evac::kPersistRing * 64 + 128) + E;
        if (hipMemsetAsync(h->p.team_rec, 0xff, h->team_xchg_bytes, S) != hipSuccess || hipMemsetAsync(decision, 0, 4 * (size_t)E, S) != hipSuccess) {
            (void)hipGetLastError();
            return fail(h, EVAC_ERR_HIP, "evac_rollout (persistent team kernel): hipMemsetAsync failed");
        }
        const dim3 grid(team_grid(h)), block(1024);
        const int* np_ = nullptr;
        int* nq_ = nullptr;
        void* argv[] = {(void*)&h->p, (void*)&np_, (void*)&nq_, (void*)&np_, (void*)&nq_, (void*)&ca};
        const bool chained = h->device >= 0 && h->device < kMaxDevices;
        std::unique_lock<std::mutex> chain(g_team_chain_lock, std::defer_lock);
        if (chained) {
            chain.lock();
            hipEvent_t& ev = g_team_chain[h->device];
            if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev = nullptr; }
            if (ev && hipStreamWaitEvent(S, ev, 0) != hipSuccess) (void)hipGetLastError();
        }
        const hipError_t le = hipLaunchKernel(team_persist_kernel(h), grid, block, argv, 0, S);
        if (chained && g_team_chain[h->device] && le == hipSuccess && hipEventRecord(g_team_chain[h->device], S) != hipSuccess) (void)hipGetLastError();
        if (chained) chain.unlock();
        if (le != hipSuccess) return fail(h, EVAC_ERR_HIP, std::string("evac_rollout (persistent team kernel): ") + hipGetErrorString(le));
    } else if (h->cu_wide4) {
        const dim3 grid((unsigned)(E / FW4::kEnvsPerBlock));
        if (h->default_cfg && h->p.obs_pos == EVAC_POS_GRAV)
            hipLaunchKernelGGL((evac::k_rollout_persist_default_config<FW4, true>), grid, dim3(FW4::kBlock), 0, S, EVAC_PERSIST_ARGS);
        else if (h->default_cfg)
            hipLaunchKernelGGL((evac::k_rollout_persist_default_config<FW4, false>), grid, dim3(FW4::kBlock), 0, S, EVAC_PERSIST_ARGS);
        else if (h->p.obs_pos == EVAC_POS_GRAV)
            hipLaunchKernelGGL((evac::k_rollout_persist<FW4, true>), grid, dim3(FW4::kBlock), 0, S, EVAC_PERSIST_ARGS);
        else
            hipLaunchKernelGGL((evac::k_rollout_persist<FW4, false>), grid, dim3(FW4::kBlock), 0, S, EVAC_PERSIST_ARGS);
    } else {
        const dim3 grid((unsigned)(E / FW::kEnvsPerBlock));
        if (h->default_cfg && h->p.obs_pos == EVAC_POS_GRAV)
            hipLaunchKernelGGL((evac::k_rollout_persist_default_config<FW, true>), grid, dim3(FW::kBlock), 0, S, EVAC_PERSIST_ARGS);
        else if (h->default_cfg)
            hipLaunchKernelGGL((evac::k_rollout_persist_default_config<FW, false>), grid, dim3(FW::kBlock), 0, S, EVAC_PERSIST_ARGS);
        else if (h->p.obs_pos == EVAC_POS_GRAV)
            hipLaunchKernelGGL((evac::k_rollout_persist<FW, true>), grid, dim3(FW::kBlock), 0, S, EVAC_PERSIST_ARGS);
        else
            hipLaunchKernelGGL((evac::k_rollout_persist<FW, false>), grid, dim3(FW::kBlock), 0, S, EVAC_PERSIST_ARGS);
    }
#undef EVAC_PERSIST_ARGS
    if (const int lc = check_launch(h, "evac_rollout (persistent kernel)"); lc != EVAC_OK) return lc;
    if (hipEventRecord(h->chain_ev, S) != hipSuccess) { (void)hipGetLastError(); return fail(h, EVAC_ERR_HIP, "evac_rollout: event record behind the persistent kernel failed"); }
    return EVAC_OK;
}
// STOP, and behind the resident kernel a FINISHER: a kernel that takes up every env that has not yet run everything up to the STOP (the
// resident kernel may have left, or be leaving, for lack of commands) and ends at once where there is nothing to do.
int stop_persistent(evac_handle* h) {
    const int stop_index = h->persist_seq;
    post_command(h, 0, nullptr, nullptr, nullptr);
    h->persist_running = false;
    return launch_persistent(h, /*resume=*/1, stop_index, /*fresh=*/false);
}
int join_parts(evac_handle* h, hipStream_t stream) {
    if (!h->part_stream[0] || !h->parts_pending) return EVAC_OK;
    DeviceGuard g(h->device);
    if (h->persist && h->persist_running)              // STOP: the waves store their state and the kernel ends (+ the finisher)
        if (const int rc = stop_persistent(h); rc != EVAC_OK) return rc;
    for (int k = 0; k < 2; ++k)
        if (hipEventRecord(h->part_done[k], h->part_stream[k]) != hipSuccess || hipStreamWaitEvent(stream, h->part_done[k], 0) != hipSuccess) {
            (void)hipGetLastError();
            return fail(h, EVAC_ERR_HIP, "evac_join: event record / wait failed");
        }
    h->parts_pending = false;
    h->forked = false;
    if (h->chain && h->chain_dirty && h->chain_xchg) {     // the chain's launches kept the state in the exchange records: back to the caller's arrays
        const int wpe = h->cu_wide4 ? 4 : 1;
        hipLaunchKernelGGL(evac::k_chain_export, dim3((unsigned)((h->p.n_envs * wpe + 3) / 4)), dim3(256), 0, stream, h->p, (const char*)h->chain_xchg, wpe);
        h->chain_dirty = false;
        if (hipGetLastError() != hipSuccess) return fail(h, EVAC_ERR_HIP, "evac_join: export of the chain's state failed");
    }
    return EVAC_OK;
}
}  // namespace

extern "C" {

int evac_create(const evac_config_t* cfg, int32_t num_envs, int32_t device, uint64_t seed, uint64_t env_id_offset,
                evac_handle_t* out) {
    return evac_create_ex(cfg, num_envs, device, seed, env_id_offset, nullptr, out);
}

int evac_create_ex(const evac_config_t* cfg, int32_t num_envs, int32_t device, uint64_t seed, uint64_t env_id_offset,
                   const evac_options_t* options, evac_handle_t* out) {
    evac_options_t o = EVAC_OPTIONS_AUTO;
    if (options) o = *options; else o.parts = 1;         // (evac_create: the stream contract existing callers rely on)
    const int32_t* f = &o.subwave;
    for (int k = 0; k < (int)(sizeof(o) / sizeof(int32_t)); ++k)
        if (f[k] < -1 || f[k] > 16) { g_create_error = "evac_options_t: every field must be -1 (automatic) or a small non-negative value"; if (out) *out = nullptr; return EVAC_ERR_INVALID_ARGUMENT; }
    if (!options) o.chain = 0;
    if (o.parts == 0 || o.parts > 2) { g_create_error = "evac_options_t.parts must be -1, 1 or 2"; if (out) *out = nullptr; return EVAC_ERR_INVALID_ARGUMENT; }
    if (o.chain > 2) { g_create_error = "evac_options_t.chain must be -1, 0, 1 or 2"; if (out) *out = nullptr; return EVAC_ERR_INVALID_ARGUMENT; }
    const int rc = create_impl(cfg, num_envs, device, seed, env_id_offset, o, out);
    if (rc != EVAC_OK) return rc;
    evac_handle* h = *out;
    // Chained launches (include/evac.h): the CU-wide kernels of one-wave envs, whole workgroups only (a wave that gives up must not
    // leave others at a barrier: that family has none), and a host-mapped error word like the teams'.  Wins over parts.
    const int chain_opt = option_value("EVAC_CHAIN", o.chain);
    int can_wait_value = 0;
    if (hipDeviceGetAttribute(&can_wait_value, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess) { (void)hipGetLastError(); can_wait_value = 0; }
    const bool chain_one_wave = h->cu_wide && num_envs % 16 == 0 && num_envs >= 32;
    const bool chain_four_waves = h->cu_wide4 && num_envs % 4 == 0 && num_envs >= 8;     // (the CU-wide form of four-wave envs: a barrier per env in LDS)
    // ... and the 256-thread workgroups of one-wave envs (four envs each), on request only (chain = 1 with cu_wide = 0)
    const bool chain_small = chain_opt == 1 && !h->cu_wide && !h->cu_wide4 && h->sub_lanes == 0 && !h->cells && !h->team_k &&
                             waves_per_env(h->p.n_ped) == 1 && num_envs % 4 == 0 && num_envs >= 8;
    h->chain_small = chain_small;
    // chain = 2: ONE PERSISTENT KERNEL PER JOIN (evac_common.h, PersistCmd).  The CU-wide kernels only (every workgroup resident at once: the grid
    // must fit the device), and only where the host can write device memory directly (large BAR): the command ring lives in uncached device
    // memory, written by the CPU -- no stream operation of ours could run while the resident kernel holds every CU.
    if (chain_opt == 2) {
        int large_bar = 0;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess) { (void)hipGetLastError(); large_bar = 0; }
        const int per_wg = h->cu_wide4 ? 4 : 16;
        // (teams -- one env on K CUs: the grid's fit is checked where it is for plain team launches, at the call, once the exchange areas are bound)
        const bool fits = ((h->cu_wide || h->cu_wide4) && num_envs % per_wg == 0 && num_envs / per_wg <= h->cus) || (h->team_k != 0 && !h->team_fault);
        bool ok = large_bar != 0 && fits && make_part_streams(h);
        if (ok) {
            DeviceGuard g(device);
            void* host = (void*)h->team_flag_host;
            void* dev = (void*)h->team_flag_dev;
            if (!host && !take_error_word(device, &host, &dev)) ok = false;
            if (ok) { h->team_flag_host = (volatile unsigned*)host; h->team_flag_dev = (unsigned*)dev; }
            ok = ok && hipEventCreateWithFlags(&h->chain_ev, hipEventDisableTiming) == hipSuccess;      // (behind every persistent kernel: has it left?)
            void* ring = nullptr;
            const size_t rbytes = (size_t)evac::kPersistRing * 64 + 128 + 8 * (size_t)num_envs;      // the ring, a line of diagnostics, next_cmd[E], decision[E] (teams)
            if (ok) { ring = take_uncached(device, rbytes, &h->chain_xchg_bytes); ok = ring != nullptr; }
            if (ok && (hipMemset(ring, 0, rbytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) { (void)hipGetLastError(); ok = false; }
            h->chain_xchg = (char*)ring;
            if (ok) h->chain_abort = (unsigned*)((char*)ring + (size_t)evac::kPersistRing * 64);
        }
        if (ok) {
            h->persist = true;
            h->opt.chain = 2;
            h->variant[3] = h->variant[1] + ", one persistent kernel per join";
            return EVAC_OK;
        }
        if (h->part_stream[0]) destroy_parts(h);       // (not possible here: chained launches if they are, else plain ones)
    }
    if (chain_opt != 0 && can_wait_value && (chain_one_wave || chain_four_waves || chain_small)) {
        bool ok = make_part_streams(h);
        {
            DeviceGuard g(device);
            void* host = nullptr;
            void* dev = nullptr;
            ok = ok && hipEventCreateWithFlags(&h->chain_ev, hipEventDisableTiming) == hipSuccess;
            if (ok && h->team_flag_host) {          // (a handle has one error word: the teams' serves the chain too)
                host = (void*)h->team_flag_host;
                dev = (void*)h->team_flag_dev;
            } else if (ok && !take_error_word(device, &host, &dev)) {
                ok = false;
            }
            if (ok) {
                h->team_flag_host = (volatile unsigned*)host;
                h->team_flag_dev = (unsigned*)dev;
            } else {
                (void)hipGetLastError();
            }
            // The exchange records (evac_common.h) live in UNCACHED device memory, the second thing a chained handle allocates
            // itself.  In ordinary hipMalloc memory -- the caller's workspace -- a record line can sit in the L2 of an XCD that once
            // touched it with a plain access (the workspace's zero fill, the import), and another XCD's write-through store does not
            // refresh that copy: an import written with `sc1` stores over the zero-filled workspace read back as zeros in ~1.5 of a
            // record's 12 lines (tools/chain_debug.py, round 6).  A plain-store import cured that case, but the same can happen to the
            // import's own lines once a batch that fills the chip lets workgroups land on whatever XCD has room.  Memory that no L2
            // ever holds removes the question: 1.5 KB per env and launch at memory speed is nothing next to 20 steps.
            void* xchg = nullptr;
            const size_t xbytes = (size_t)num_envs * (size_t)evac::xchg_bytes(h->cu_wide4 ? 256 : 64);
            if (ok) {
                xchg = take_uncached(device, xbytes, &h->chain_xchg_bytes);
                ok = xchg != nullptr;
            }
            if (ok && (hipMemset(xchg, 0, xbytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) { (void)hipGetLastError(); ok = false; }   // (hipMemset does not wait)
            h->chain_xchg = (char*)xchg;
        }
        if (ok) {
            h->chain = true;
            h->opt.chain = 1;
            h->variant[3] = h->variant[1] + ", chained launches on 2 streams";
            (void)chain_one_wave;
            return EVAC_OK;
        }
        destroy_parts(h);                              // (no second queue: plain launches)
    }
    // Two parts: where it pays by itself (-1) -- CU-wide rollouts whose halves are whole CU-wide workgroups; the halves keep the
    // CU-wide form although each alone would not fill the device (pace keeping and the in-kernel deal are what they are to keep) --
    // or on request (2) for every handle but the teams' (their grids run one at a time: evac_rollout).
    const int per_wg = h->cu_wide4 ? 4 : 16;
    const int parts_opt = option_value("EVAC_PARTS", o.parts);
    const bool can = h->team_k == 0 && num_envs >= 2 && num_envs % 2 == 0;
    const bool pays = (h->cu_wide || h->cu_wide4) && num_envs % (2 * per_wg) == 0;
    if (!(can && (parts_opt == 2 || (parts_opt < 0 && pays)))) return EVAC_OK;
    evac_options_t po = h->opt;                        // the parts take the parent's resolved choices
    po.parts = 1;
    po.chain = 0;
    const int32_t half = num_envs / 2;
    bool ok = make_part_streams(h);
    for (int k = 0; ok && k < 2; ++k) {
        evac_handle_t c = nullptr;
        ok = create_impl(cfg, half, device, seed, env_id_offset + (uint64_t)k * (uint64_t)half, po, &c) == EVAC_OK;
        h->part[k] = c;
        if (ok) c->p.slab_envs = num_envs;
    }
    if (!ok) { destroy_parts(h); return EVAC_OK; }     // (no second queue, no parts: the handle works as one)
    h->n_parts = 2;
    h->opt.parts = 2;
    h->variant[3] = h->part[0]->variant[1] + " x 2 streams";
    return EVAC_OK;
}

int evac_get_options(evac_handle_t h, evac_options_t* out) {
    if (!h || !out) return EVAC_ERR_INVALID_ARGUMENT;
    *out = h->opt;
    return EVAC_OK;
}

int evac_join(evac_handle_t h, void* stream) {
    if (!h) return EVAC_ERR_INVALID_ARGUMENT;
    return join_parts(h, (hipStream_t)stream);
}
int evac_order_next_rollout(evac_handle_t h) {
    if (!h) return EVAC_ERR_INVALID_ARGUMENT;
    h->forked = false;
    return EVAC_OK;
}
int32_t evac_num_parts(evac_handle_t h) { return h ? h->n_parts : -1; }
int32_t evac_own_streams(evac_handle_t h) { return h ? ((h->n_parts > 1 || h->chain || h->persist) && h->part_stream[0] ? 2 : 0) : -1; }
void* evac_part_stream(evac_handle_t h, int32_t part) { return (h && evac_own_streams(h) == 2 && part >= 0 && part < 2) ? (void*)h->part_stream[part] : nullptr; }

const char* evac_kernel_variant(evac_handle_t h, int32_t rollout) {
    if (!h) return "";
    if (!rollout) return h->variant[0].c_str();
    if (h->persist) {          // (teams: the persistent form of the team kernel, once its exchange areas are bound and its grid fits)
        const bool team = h->team_k && h->team_bound && h->team_fit != 0;
        if (h->team_k && !team) return h->variant[1].c_str();
        h->variant[3] = h->variant[team ? 2 : 1] + ", one persistent kernel per join";
        return h->variant[3].c_str();
    }
    if (h->n_parts > 1 || (h->chain && h->chain_bound)) return h->variant[3].c_str();
    // the path evac_rollout takes right now: teams only with their exchange areas bound and a grid that fits the device
    return h->variant[(h->team_k && h->team_bound && h->team_fit != 0) ? 2 : 1].c_str();
}

int evac_destroy(evac_handle_t h) {
    if (h && (h->n_parts > 1 || h->part_stream[0] || h->chain)) destroy_parts(h);
    if (h && h->team_flag_host) {
        DeviceGuard g(h->device);
        give_error_word(h->device, (void*)h->team_flag_host, (void*)h->team_flag_dev);
    }
    delete h;
    return EVAC_OK;
}

int64_t evac_obs_dim(evac_handle_t h) { return h ? h->p.obs_dim : -1; }
int32_t evac_num_envs(evac_handle_t h) { return h ? h->p.n_envs : -1; }

int64_t evac_algorithmic_bytes_per_env_step(evac_handle_t h) {
    if (!h) return -1;
    // per agent-update 16 B read + 16 B write of (x,y,dx,dy); per env-step: action 8 +
    // leader pos 8 r / 8 w + now 4 r / 4 w + reward 4 + terminated 1 + truncated 1 = 38; plus the obs write.
    return 32ll * h->p.n_ped + 38ll + 4ll * h->p.obs_dim;
}

int evac_bind_state(evac_handle_t h, float* ped, uint8_t* status, float* agent, int32_t* clock, float* acc) {
    if (!h) return EVAC_ERR_INVALID_ARGUMENT;
    if (!ped || !status || !agent || !clock || !acc) return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_bind_state: NULL buffer");
    if (((uintptr_t)ped | (uintptr_t)agent | (uintptr_t)clock | (uintptr_t)acc) & 15u)
        return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_bind_state: ped/agent/clock/acc must be 16-byte aligned");
    h->p.ped = (float4*)ped;
    h->p.status = status;
    h->p.agent = (float4*)agent;
    h->p.clock = (int4*)clock;
    h->p.acc = (float4*)acc;
    h->bound = true;
    h->chain_restart = true;
    for (int k = 0; k < (h->n_parts > 1 ? h->n_parts : 0); ++k) {      // the parts: the same buffers from their first env on
        const size_t first = (size_t)k * (size_t)h->part[k]->p.n_envs, N = (size_t)h->p.n_ped;
        const int rc = evac_bind_state(h->part[k], ped + first * N * 4, status + first * N, agent + first * 4, clock + first * 4, acc + first * 4);
        if (rc != EVAC_OK) return fail(h, rc, std::string("evac_bind_state (part): ") + h->part[k]->err);
    }
    return EVAC_OK;
}

namespace {
struct WorkspaceLayout {
    size_t sched, stats, team_rec, team_tile, team_xchg_end, chain_sched, chain_abort, total;
};
WorkspaceLayout workspace_layout(const evac_handle* h) {
    const size_t E = (size_t)h->p.n_envs;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    WorkspaceLayout w{};
    size_t o = 0;
    w.sched = o; o = up(o + 4 * E * sizeof(int32_t));          // moving[2][E] | perm[2][E]
    w.stats = o; o = up(o + 64);
    if (h->chain) {                                            // moving[4][E] | perm[4][E] | the abort word + started-workgroups counter
        w.chain_sched = o; o = up(o + 8 * E * sizeof(int32_t));    // (the exchange records are NOT here: they need memory the L2s do not cache,
        w.chain_abort = o; o = up(o + 128);                        //  which the library allocates itself: evac_create_ex)
    }
    if (h->team_k) {
        w.team_rec = o; o = up(o + evac::kTeamSets * E * 32 * 16);          // (two slot sets: evac_team.h, exchange)
        w.team_tile = o; o = up(o + evac::kTeamSets * E * 1024 * 16);
        w.team_xchg_end = o;
    }
    w.total = o;
    return w;
}
}  // namespace

int64_t evac_workspace_bytes(evac_handle_t h) {
    if (!h) return -1;
    size_t total = workspace_layout(h).total, parts = 0;
    for (int k = 0; k < (h->n_parts > 1 ? h->n_parts : 0); ++k) parts += workspace_layout(h->part[k]).total;   // (the parts carve it up between them)
    return (int64_t)(parts > total ? parts : total);
}

int evac_bind_workspace(evac_handle_t h, void* workspace, int64_t bytes) {
    if (!h) return EVAC_ERR_INVALID_ARGUMENT;
    h->sched = nullptr;
    h->sched_gen = -1;
    h->team_bound = false;
    if ((h->chain && (h->parts_pending || h->chain_dirty)) || (h->persist && h->parts_pending)) {      // (launches in flight still use the old workspace; the chain's state lives in it)
        DeviceGuard g(h->device);
        (void)join_parts(h, nullptr);
        (void)hipDeviceSynchronize();
    }
    h->chain_bound = false;
    if (h->n_parts > 1) {
        // the parts schedule themselves: each gets a slice (moving[2][E/2] | perm[2][E/2] of its own); this handle's own rollouts --
        // the diagnostic face only -- run without a schedule
        if (workspace && (bytes < evac_workspace_bytes(h) || ((uintptr_t)workspace & 255u)))
            return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_bind_workspace: workspace smaller than evac_workspace_bytes() or not 256-byte aligned");
        size_t o = 0;
        for (int k = 0; k < h->n_parts; ++k) {
            const size_t t = workspace_layout(h->part[k]).total;
            const int rc = evac_bind_workspace(h->part[k], workspace ? (char*)workspace + o : nullptr, (int64_t)t);
            if (rc != EVAC_OK) return fail(h, rc, std::string("evac_bind_workspace (part): ") + h->part[k]->err);
            o += t;
        }
        return EVAC_OK;
    }
    if (!workspace) return EVAC_OK;
    const WorkspaceLayout w = workspace_layout(h);
    if (bytes < (int64_t)w.total) return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_bind_workspace: workspace smaller than evac_workspace_bytes()");
    if ((uintptr_t)workspace & 255u) return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_bind_workspace: workspace must be 256-byte aligned");
    char* base = (char*)workspace;
    h->sched = (int32_t*)(base + w.sched);
    h->chain_bound = false;
    if (h->chain) {
        h->chain_sched = (int32_t*)(base + w.chain_sched);
        h->chain_abort = (unsigned*)(base + w.chain_abort);
        h->chain_dirty = false;
        h->chain_wgs = 0;                             // (the workspace comes zero-filled: include/evac.h)
        h->chain_bound = true;
        h->chain_restart = true;
    }
    if (h->team_k) {
        h->p.team_err = h->team_flag_dev;           // (host-mapped memory of the handle, not part of the workspace)
        h->p.team_rec = base + w.team_rec;
        h->p.team_tile = base + w.team_tile;
        h->team_xchg_bytes = w.team_xchg_end - w.team_rec;
        h->team_bound = true;
        DeviceGuard g(h->device);
        (void)team_grid_fits(h);                      // (so that evac_kernel_variant names the path the first rollout will take)
    }
    return EVAC_OK;
}

namespace {
// the explicit deal (k_schedule): the permutation launch `gen` will read, from the loads launch gen - 1 left; `both`: the other
// permutation buffer too
void deal_now(evac_handle_t h, hipStream_t s, bool both) {
    const int E = h->p.n_envs;
    if (h->sched_gen < 0) h->sched_gen = 0;
    const int g = h->sched_gen;
    hipLaunchKernelGGL(evac::k_schedule, dim3(1), dim3(1024), 0, s, E, (const int*)(h->sched + ((g + 1) & 1) * E),
                       h->sched + (2 + (g & 1)) * E, both ? h->sched + (2 + ((g + 1) & 1)) * E : (int32_t*)nullptr,
                       h->cu_wide4 ? 4 : 16, h->cu_wide4 ? 4 : 1, 0);
}
}  // namespace

int evac_reschedule(evac_handle_t h, void* stream) {
    if (!h) return EVAC_ERR_INVALID_ARGUMENT;
    if (h->chain) {                                    // (the chain deals itself again when it restarts)
        h->chain_restart = true;
        return join_parts(h, (hipStream_t)stream);
    }
    if (h->n_parts > 1) {                              // (on the caller's stream, behind everything the parts have been given)
        if (const int rc = join_parts(h, (hipStream_t)stream); rc != EVAC_OK) return rc;
        for (int k = 0; k < h->n_parts; ++k)
            if (const int rc = evac_reschedule(h->part[k], stream); rc != EVAC_OK) return fail(h, rc, h->part[k]->err);
        return EVAC_OK;
    }
    if (!h->sched || !(h->cu_wide || h->cu_wide4)) return EVAC_OK;       // nothing to deal
    if (h->persist && h->parts_pending)                                   // (the resident kernel keeps the deal it started with: it ends first)
        if (const int rc = join_parts(h, (hipStream_t)stream); rc != EVAC_OK) return rc;
    DeviceGuard g(h->device);
    deal_now(h, (hipStream_t)stream, true);
    return check_launch(h, "evac_reschedule");
}

int32_t evac_schedule_generation(evac_handle_t h) {
    if (h && h->n_parts > 1) return evac_schedule_generation(h->part[0]);
    return (h && h->sched && (h->cu_wide || h->cu_wide4)) ? h->sched_gen : -1;
}

int evac_peer_gather(const float* src, int64_t rows, int32_t row_words, int32_t take_words, float* const* peer_dst, int32_t world,
                     int32_t my_rank, int32_t wgs_per_peer, void* stream) {
    if (!src || !peer_dst || rows <= 0 || row_words <= 0 || take_words <= 0 || take_words > row_words || world < 1 ||
        world > evac::kMaxPeers || my_rank < 0 || my_rank >= world)
        return EVAC_ERR_INVALID_ARGUMENT;
    if (rows * (int64_t)take_words >= (int64_t)1 << 31) return EVAC_ERR_INVALID_ARGUMENT;      // (32-bit element indices)
    evac::PeerPtrs pp{};
    for (int r = 0; r < world; ++r) {
        if (!peer_dst[r]) return EVAC_ERR_INVALID_ARGUMENT;
        pp.dst[r] = peer_dst[r];
    }
    hipPointerAttribute_t attr;                       // (no handle: the kernel must run on the device that owns the slab)
    int src_dev = -1;
    if (hipPointerGetAttributes(&attr, src) == hipSuccess) src_dev = attr.device; else (void)hipGetLastError();
    DeviceGuard g(src_dev);
    const unsigned n = (unsigned)(rows * take_words);
    int w = wgs_per_peer > 0 ? wgs_per_peer : 8;
    const int need = (int)((n + 1023u) / 1024u);                 // (no more workgroups than 1024-element pieces)
    if (w > need) w = need < 1 ? 1 : need;
    const dim3 grid((unsigned)(w * world)), block(256);
    hipStream_t s_ = (hipStream_t)stream;
    if (take_words == 6)
        hipLaunchKernelGGL((evac::k_peer_gather<6>), grid, block, 0, s_, src, n, (unsigned)row_words, 6u, pp, (int)my_rank, (int)world, n, w);
    else if (take_words == 9)
        hipLaunchKernelGGL((evac::k_peer_gather<9>), grid, block, 0, s_, src, n, (unsigned)row_words, 9u, pp, (int)my_rank, (int)world, n, w);
    else
        hipLaunchKernelGGL((evac::k_peer_gather<0>), grid, block, 0, s_, src, n, (unsigned)row_words, (unsigned)take_words, pp, (int)my_rank,
                           (int)world, n, w);
    return hipGetLastError() == hipSuccess ? EVAC_OK : EVAC_ERR_HIP;
}

int evac_team_error(evac_handle_t h, int32_t* out) {
    if (!h || !out) return EVAC_ERR_INVALID_ARGUMENT;
    *out = 0;
    if (!h->team_flag_host) return EVAC_OK;
    DeviceGuard g(h->device);
    if (hipDeviceSynchronize() != hipSuccess) return fail(h, EVAC_ERR_HIP, "evac_team_error: hipDeviceSynchronize failed");
    *out = (int32_t)*h->team_flag_host;
    return EVAC_OK;
}

int evac_team_error_nosync(evac_handle_t h, int32_t* out) {
    if (!h || !out) return EVAC_ERR_INVALID_ARGUMENT;
    *out = h->team_flag_host ? (int32_t)*h->team_flag_host : 0;
    return EVAC_OK;
}

int evac_team_clear_error(evac_handle_t h) {
    if (!h) return EVAC_ERR_INVALID_ARGUMENT;
    if (h->team_flag_host) {
        if (*h->team_flag_host != 0u) { h->team_k = 0; h->chain = false; h->persist = false; }     // the handle stays on one workgroup per env / on plain launches
        *h->team_flag_host = 0u;
    }
    return EVAC_OK;
}

// (a handle with two parts: whatever is not a plain rollout first makes the caller's stream wait for the parts' streams)
#define EVAC_REQUIRE_BOUND(h, name)                                               \
    if (!(h)) return EVAC_ERR_INVALID_ARGUMENT;                                  \
    if (!(h)->bound) return fail((h), EVAC_ERR_NOT_BOUND, name ": call evac_bind_state first"); \
    if (const int ta_ = team_aborted((h), name); ta_ != EVAC_OK) return ta_
#define EVAC_JOIN_FIRST(h, stream)                                                \
    (h)->chain_restart = true;      /* (whatever follows may write the state: the chain starts afresh behind it) */ \
    if ((h)->parts_pending)                                                       \
        if (const int jn_ = join_parts((h), (hipStream_t)(stream)); jn_ != EVAC_OK) return jn_

int evac_reset(evac_handle_t h, const uint8_t* mask, const float* draws, float* obs_out, void* stream) {
    EVAC_REQUIRE_BOUND(h, "evac_reset");
    EVAC_JOIN_FIRST(h, stream);
    if (draws && ((uintptr_t)draws & 15u)) return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_reset: draws must be 16-byte aligned");
    DeviceGuard g(h->device);
    EVAC_DISPATCH(h, k_reset, stream, h->p, mask, (const float4*)draws, obs_out);
    return check_launch(h, "evac_reset");
}

#define EVAC_STEP_ARGS h->p, (const float2*)actions, noise, obs_out, reward_out, terminated_out, truncated_out, (int)autoreset, final_obs, final_stats, na

static int step_common(evac_handle_t h, const char* name, const float* actions, const float* noise, float* obs_out,
                       float* reward_out, uint8_t* terminated_out, uint8_t* truncated_out, int32_t autoreset, float* final_obs,
                       evac_episode_stats_t* final_stats, const evac::NormArgs& na, void* stream) {
    if (!actions || !obs_out || !reward_out || !terminated_out || !truncated_out)
        return fail(h, EVAC_ERR_INVALID_ARGUMENT, std::string(name) + ": actions/obs/reward/terminated/truncated must be non-NULL");
    if ((uintptr_t)actions & 7u) return fail(h, EVAC_ERR_INVALID_ARGUMENT, std::string(name) + ": actions must be 8-byte aligned");
    DeviceGuard g(h->device);
    if (na.state && h->default_cfg) EVAC_DISPATCH(h, k_step_norm_default_config, stream, EVAC_STEP_ARGS);
    else if (na.state) EVAC_DISPATCH(h, k_step_norm, stream, EVAC_STEP_ARGS);
    else if (h->default_cfg) EVAC_DISPATCH(h, k_step_default_config, stream, EVAC_STEP_ARGS);
    else EVAC_DISPATCH(h, k_step_raw, stream, EVAC_STEP_ARGS);
    return check_launch(h, name);
}

int evac_step(evac_handle_t h, const float* actions, const float* noise, float* obs_out, float* reward_out,
              uint8_t* terminated_out, uint8_t* truncated_out, int32_t autoreset, float* final_obs,
              evac_episode_stats_t* final_stats, void* stream) {
    EVAC_REQUIRE_BOUND(h, "evac_step");
    EVAC_JOIN_FIRST(h, stream);
    return step_common(h, "evac_step", actions, noise, obs_out, reward_out, terminated_out, truncated_out, autoreset, final_obs,
                       final_stats, evac::NormArgs{nullptr, 0.f, 0.f, 0.f, 0.f}, stream);
}

int evac_step_normalized(evac_handle_t h, const float* actions, const float* noise, float* obs_out, float* reward_out,
                         uint8_t* terminated_out, uint8_t* truncated_out, int32_t autoreset, float* final_obs,
                         evac_episode_stats_t* final_stats, double* norm_state, float gamma, float obs_clip,
                         float reward_clip, float epsilon, void* stream) {
    EVAC_REQUIRE_BOUND(h, "evac_step_normalized");
    EVAC_JOIN_FIRST(h, stream);
    if (!norm_state) return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_step_normalized: norm_state is NULL");
    return step_common(h, "evac_step_normalized", actions, noise, obs_out, reward_out, terminated_out, truncated_out, autoreset,
                       final_obs, final_stats, evac::NormArgs{norm_state, gamma, obs_clip, reward_clip, epsilon}, stream);
}

int evac_rollout(evac_handle_t h, int32_t n_steps, const float* actions, float* actions_out, float* slab_out,
                 evac_episode_stats_t* final_stats, int32_t capture_envs, float* capture, const float* noise,
                 void* stream) {
    EVAC_REQUIRE_BOUND(h, "evac_rollout");
    if (n_steps < 1) return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_rollout: n_steps must be >= 1");
    if (!slab_out) return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_rollout: slab_out must be non-NULL");
    if (((uintptr_t)actions | (uintptr_t)actions_out) & 7u)
        return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_rollout: actions buffers must be 8-byte aligned");
    DeviceGuard g(h->device);
    if (capture && (capture_envs < 1 || capture_envs > h->p.n_envs))
        return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_rollout: capture_envs must be in [1, num_envs] when capture is given");
    if (h->n_parts > 1 && !(capture || actions_out || noise)) {
        // two half-batch kernels on the handle's own streams, both behind what `stream` holds so far; `stream` is NOT made to
        // wait for them (evac_join): consecutive rollout calls must not meet, or the halves would run in lock-step
        hipStream_t s_ = (hipStream_t)stream;
        // (the fork costs a barrier packet in front of each kernel -- 7-10 us on this platform whether or not the event has fired,
        // DESIGN.md 6 -- which is more than the parts gain.  So the own streams are put behind the caller's stream ONCE per join: at
        // the first rollout call after evac_join / any other call on the handle, and at every call that brings inputs (actions).
        // Asking the stream instead -- hipStreamQuery -- was tried: the query leaves a marker in the stream, the next query finds it
        // busy, and the handle falls into forking at every call: a second, 15-40 % slower mode of the same program.)
        const bool fork = !h->forked || actions != nullptr;
        if (fork) {
            if (hipEventRecord(h->fork_ev, s_) != hipSuccess) { (void)hipGetLastError(); return fail(h, EVAC_ERR_HIP, "evac_rollout: hipEventRecord failed"); }
            h->forked = true;
        }
        const size_t row = (size_t)h->p.obs_dim + 3;
        for (int k = 0; k < h->n_parts; ++k) {
            evac_handle* c = h->part[k];
            const size_t first = (size_t)k * (size_t)c->p.n_envs;
            if (fork && hipStreamWaitEvent(h->part_stream[k], h->fork_ev, 0) != hipSuccess) { (void)hipGetLastError(); return fail(h, EVAC_ERR_HIP, "evac_rollout: hipStreamWaitEvent failed"); }
            h->parts_pending = true;
            const int rc = evac_rollout(c, n_steps, actions ? actions + first * 2 : nullptr, nullptr, slab_out + first * row,
                                        final_stats ? final_stats + first : nullptr, 0, nullptr, nullptr, h->part_stream[k]);
            if (rc != EVAC_OK) return fail(h, rc, std::string("evac_rollout (part): ") + c->err);
        }
        return EVAC_OK;
    }
    if (h->persist && !(capture || actions_out || noise || actions) && (!h->team_k || (h->team_bound && team_grid_fits(h)))) {
        hipStream_t s_ = (hipStream_t)stream;
        hipStreamCaptureStatus pcap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s_, &pcap) != hipSuccess) { (void)hipGetLastError(); pcap = hipStreamCaptureStatusNone; }
        if (pcap == hipStreamCaptureStatusNone) {
            // ONE PERSISTENT KERNEL PER JOIN: the call becomes a command of the resident kernel's ring.  The kernel is started -- behind
            // what the caller's stream holds at this moment, as the parts' fork -- by the first call after a join (or after
            // evac_order_next_rollout); the calls that follow cost the host a 64-byte write through the BAR and the device nothing but the
            // steps: the state stays in registers.  A kernel that found no command for ~150 us has LEFT by itself (every env's state and place
            // in the ring stored): the call then starts one that takes every env up where it stopped.  (Given actions take the plain path
            // below: their buffer is the caller's stream's business.)
            hipStream_t S = h->part_stream[0];
            if (h->persist_running && !h->forked)            // the caller touched a buffer (evac_order_next_rollout): a new kernel behind a new fork
                if (const int rc = stop_persistent(h); rc != EVAC_OK) return rc;
            if (h->persist_running && h->persist_seq - h->persist_first >= evac::kPersistRing - 2) {
                // the ring is about to lap the slowest env: the kernel is stopped and WAITED FOR on the host (once per ~1000 calls without a join)
                if (const int rc = stop_persistent(h); rc != EVAC_OK) return rc;
                if (hipStreamSynchronize(S) != hipSuccess) { (void)hipGetLastError(); return fail(h, EVAC_ERR_HIP, "evac_rollout: the persistent kernel did not end"); }
            }
            if (!h->persist_running) {
                if (hipEventRecord(h->fork_ev, s_) != hipSuccess || hipStreamWaitEvent(S, h->fork_ev, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    return fail(h, EVAC_ERR_HIP, "evac_rollout: fork of the persistent kernel failed");
                }
                h->forked = true;
                if (const int rc = launch_persistent(h, /*resume=*/0, /*stop_at=*/0x7fffffff, /*fresh=*/true); rc != EVAC_OK) return rc;
                h->persist_running = true;
                h->persist_first = h->persist_seq;
            } else if (hipEventQuery(h->chain_ev) == hipSuccess) {
                // the kernel has left (idle): one that resumes.  (A kernel that is leaving RIGHT NOW is seen at the next call or at the join,
                // whose finisher runs whatever an env has not run yet: no command is lost, it only waits for that kernel.)
                if (const int rc = launch_persistent(h, /*resume=*/1, /*stop_at=*/0x7fffffff, /*fresh=*/false); rc != EVAC_OK) return rc;
            } else {
                (void)hipGetLastError();                         // (hipErrorNotReady: resident)
            }
            post_command(h, (int)n_steps, slab_out, final_stats, nullptr);
            h->parts_pending = true;
            return EVAC_OK;
        }
    }
    if (h->chain && h->chain_bound && !(capture || actions_out || noise)) {
        hipStream_t s_ = (hipStream_t)stream;
        hipStreamCaptureStatus ccap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s_, &ccap) != hipSuccess) { (void)hipGetLastError(); ccap = hipStreamCaptureStatusNone; }
        if (ccap == hipStreamCaptureStatusNone) {
            // CHAINED: launch g on stream g & 1, ordered per env on the device (include/evac.h).  Both streams start behind what the
            // caller's stream holds -- once per join (below): a barrier packet per launch costs more than the chain gains.
            using FW = evac::Wave<1, 1024>;
            using FW4 = evac::Wave<4, 1024>;
            using FS = evac::Wave<1, 256>;
            const int wpe = h->cu_wide4 ? 4 : 1, per_wg = (h->cu_wide4 || h->chain_small) ? 4 : 16;
            const int E = h->p.n_envs, c = h->chain_gen;
            hipStream_t S = h->part_stream[c & 1], O = h->part_stream[(c + 1) & 1];
            // THE INVARIANT OF THE CHAIN: launch g + 1 must not start being dispatched before every workgroup of launch g has a CU.
            // A workgroup of g + 1 holds its CU while it waits for envs of launch g; were workgroups of g still waiting for CUs then,
            // the dispatcher -- which deals a grid's workgroups to the XCDs in order -- could find an XCD's CUs all held by waiting
            // workgroups of g + 1 and launch g would never be placed (seen: both streams released by ONE event started launches g and
            // g + 1 together and the second launch of a sweep timed out once in ~2000 sweeps; another kernel holding CUs while the
            // chain runs does the same; short of a deadlock the interleaved start left the pipeline in a 15-40 % slower rhythm for
            // the whole sweep).  So every workgroup of a chained launch counts itself in `started` when it gets its CU, and the QUEUE
            // of launch g + 1 waits -- hipStreamWaitValue64: the runtime's one-wave wait kernel (__amd_rocclr_streamOpsWait in a kernel
            // trace), one wave slot held, no workgroup of ours -- until the counter says that all workgroups of
            // launches <= g have started (+0.5 us per launch: tools/microbench/waitvalue.hip).  With it every wait inside a kernel is
            // for a workgroup that is resident or done, by induction down to the oldest launch in flight, which waits for nothing.
            const bool fork = h->chain_restart || !h->forked || actions != nullptr;      // (once per join, and with every new input: see the parts' fork above)
            if (fork) {
                if (hipEventRecord(h->fork_ev, s_) != hipSuccess || hipStreamWaitEvent(S, h->fork_ev, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    return fail(h, EVAC_ERR_HIP, "evac_rollout: fork of the chain failed");
                }
                h->forked = true;
            }
            int32_t* moving = h->chain_sched;
            int32_t* perm = h->chain_sched + 4 * (size_t)E;
            if (h->chain_restart) {
                // the state in memory is whatever the caller's stream left: every env at generation c, one deal in all four
                // permutation buffers, the other stream behind both
                hipLaunchKernelGGL(evac::k_chain_import, dim3((unsigned)((E * wpe + 3) / 4)), dim3(256), 0, S, h->p, h->chain_xchg, c, h->chain_abort, h->chain_wgs, wpe);
                if (!h->chain_small) {
                    hipLaunchKernelGGL(evac::k_schedule, dim3(1), dim3(1024), 0, S, E, (const int*)(moving + ((c + 3) & 3) * (size_t)E),
                                       perm + (c & 3) * (size_t)E, (int32_t*)nullptr, per_wg, wpe == 4 ? 4 : 1, option_value("EVAC_CHAIN_DEAL", 0));
                    hipLaunchKernelGGL(evac::k_copy_perm3, dim3(64), dim3(256), 0, S, E, (const int*)(perm + (c & 3) * (size_t)E),
                                       perm + ((c + 1) & 3) * (size_t)E, perm + ((c + 2) & 3) * (size_t)E, perm + ((c + 3) & 3) * (size_t)E);
                }
                // ... and the OTHER queue behind all of this.  Its gate alone does not order it: until the import has set the counter the
                // word holds whatever the workspace's memory held -- the caller's zero fill may not have run yet on this queue's
                // timeline, a recycled allocation carries the count of the handle that used it before -- and a gate that passes on such a
                // value starts launch c + 1 before the deal above exists: it then reads a permutation of ANOTHER batch (seen: a 64-env
                // handle in memory a 512-env handle had used took env indices up to 511 -- tests/test_gpu_parity.py, whole file only).
                if (hipEventRecord(h->chain_ev, S) != hipSuccess || hipStreamWaitEvent(O, h->chain_ev, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    return fail(h, EVAC_ERR_HIP, "evac_rollout: restart of the chain failed");
                }
                h->chain_start = c;
                h->chain_restart = false;
            }
            const bool deals = !h->chain_small && c - h->chain_start >= 2;          // (the loads of launch c - 2, the last launch of this stream)
            const int32_t* deal_loads = deals ? moving + ((c + 2) & 3) * (size_t)E : nullptr;
            int32_t* deal_perm = deals ? perm + ((c + 2) & 3) * (size_t)E : nullptr;
            unsigned long long* started = (unsigned long long*)(h->chain_abort + 8);      // (the same line as the abort word: bytes 32..39)
            // (the counter is never reset while the workspace is bound: a restart's first launch is gated too -- on the launches before
            // the restart, long done -- and the launch after it on the restart's own workgroups, hence behind its import and deal)
            if (h->chain_wgs > 0 && hipStreamWaitValue64(S, started, h->chain_wgs, hipStreamWaitValueGte, ~0ull) != hipSuccess) {
                (void)hipGetLastError();
                return fail(h, EVAC_ERR_HIP, "evac_rollout: hipStreamWaitValue64 (the chain's dispatch gate) failed");
            }
            static const int deal_mode = option_value("EVAC_CHAIN_DEAL", 0);                    // (diagnostic: A/B runs of one binary)
            evac::ChainArgs ca{h->chain_xchg, c, h->chain_abort, h->team_flag_dev, deal_mode, started};
            evac::Params pp = h->p;
            if (h->team_fault && c == h->chain_start + 1) pp.n_envs = E - per_wg;    // fault injection: the last workgroup of ONE launch is never run
#define EVAC_CHAIN_ARGS pp, (int)n_steps, (const float2*)actions, slab_out, final_stats, (const int*)(h->chain_small ? nullptr : perm + (c & 3) * (size_t)E), (int*)(h->chain_small ? nullptr : moving + (c & 3) * (size_t)E), (const int*)deal_loads, (int*)deal_perm, ca
            const dim3 grid((unsigned)(E / per_wg));
            if (h->chain_small) {
                // four one-wave envs per 256-thread workgroup, env = slot: the dispatcher places a workgroup of the next launch wherever
                // four waves have retired, so the chain needs no deal and no pace keeping to keep the CUs full
                if (h->default_cfg && h->p.obs_pos == EVAC_POS_GRAV)
                    hipLaunchKernelGGL((evac::k_rollout_chain_default_config<FS, true>), grid, dim3(FS::kBlock), 0, S, EVAC_CHAIN_ARGS);
                else if (h->default_cfg)
                    hipLaunchKernelGGL((evac::k_rollout_chain_default_config<FS, false>), grid, dim3(FS::kBlock), 0, S, EVAC_CHAIN_ARGS);
                else if (h->p.obs_pos == EVAC_POS_GRAV)
                    hipLaunchKernelGGL((evac::k_rollout_chain<FS, true>), grid, dim3(FS::kBlock), 0, S, EVAC_CHAIN_ARGS);
                else
                    hipLaunchKernelGGL((evac::k_rollout_chain<FS, false>), grid, dim3(FS::kBlock), 0, S, EVAC_CHAIN_ARGS);
            } else if (h->cu_wide4) {
                if (h->default_cfg && h->p.obs_pos == EVAC_POS_GRAV)
                    hipLaunchKernelGGL((evac::k_rollout_chain_default_config<FW4, true>), grid, dim3(FW4::kBlock), 0, S, EVAC_CHAIN_ARGS);
                else if (h->default_cfg)
                    hipLaunchKernelGGL((evac::k_rollout_chain_default_config<FW4, false>), grid, dim3(FW4::kBlock), 0, S, EVAC_CHAIN_ARGS);
                else if (h->p.obs_pos == EVAC_POS_GRAV)
                    hipLaunchKernelGGL((evac::k_rollout_chain<FW4, true>), grid, dim3(FW4::kBlock), 0, S, EVAC_CHAIN_ARGS);
                else
                    hipLaunchKernelGGL((evac::k_rollout_chain<FW4, false>), grid, dim3(FW4::kBlock), 0, S, EVAC_CHAIN_ARGS);
            } else if (h->default_cfg && h->p.obs_pos == EVAC_POS_GRAV)
                hipLaunchKernelGGL((evac::k_rollout_chain_default_config<FW, true>), grid, dim3(FW::kBlock), 0, S, EVAC_CHAIN_ARGS);
            else if (h->default_cfg)
                hipLaunchKernelGGL((evac::k_rollout_chain_default_config<FW, false>), grid, dim3(FW::kBlock), 0, S, EVAC_CHAIN_ARGS);
            else if (h->p.obs_pos == EVAC_POS_GRAV)
                hipLaunchKernelGGL((evac::k_rollout_chain<FW, true>), grid, dim3(FW::kBlock), 0, S, EVAC_CHAIN_ARGS);
            else
                hipLaunchKernelGGL((evac::k_rollout_chain<FW, false>), grid, dim3(FW::kBlock), 0, S, EVAC_CHAIN_ARGS);
#undef EVAC_CHAIN_ARGS
            if (const int lc = check_launch(h, "evac_rollout (chained)"); lc != EVAC_OK) return lc;      // (a launch that never starts must not be waited for)
            h->chain_wgs += (unsigned long long)grid.x;
            h->chain_gen = c + 1;
            h->parts_pending = true;
            h->chain_dirty = true;
            return EVAC_OK;
        }
    }
    h->chain_restart = true;
    if (h->parts_pending)
        if (const int jn = join_parts(h, (hipStream_t)stream); jn != EVAC_OK) return jn;
    if (capture || actions_out || noise)
        EVAC_DISPATCH(h, k_rollout_diag, stream, h->p, (int)n_steps, (const float2*)actions, (float2*)actions_out, slab_out,
                      final_stats, (int)capture_envs, capture, noise);
    else if (h->team_k && h->team_bound && team_grid_fits(h)) {
        // 513..1024 pedestrians, few envs: K workgroups (CUs) per env (evac_team.h).  Workgroup b = j * 8 + xcd carries team (j / K) * 8 + xcd.  All members of a team spin on its
        // counter, so the whole grid must be resident at once: checked by team_grid_fits (occupancy x CUs >= workgroups; a grid
        // that does not fit runs the one-workgroup-per-env kernels below).  A foreign kernel on another stream (the sharded
        // env's all-gather) can delay a member, not starve it -- it ends, the member starts, and the bounded waits (~1 s) outlast
        // it: tests/test_gpu_team.py keeps a second stream busy throughout.  EVAC_TEAM_COOP=1 launches cooperatively instead
        // (the runtime then guarantees co-residency); it costs 3-4 % of the C5 shard's throughput and is not the default.
        hipStream_t s_ = (hipStream_t)stream;
        // every slot of the exchange area starts a launch with tag 31 in every word -- no round's (the previous launch left tagged data)
        if (hipMemsetAsync(h->p.team_rec, 0xff, h->team_xchg_bytes, s_) != hipSuccess) return fail(h, EVAC_ERR_HIP, "evac_rollout: hipMemsetAsync failed");
        const dim3 grid(team_grid(h)), block(1024);
        int n_steps_ = (int)n_steps;
        const float2* actions_ = (const float2*)actions;
        const int* perm_ = nullptr;
        int* moving_ = nullptr;
        void* argv[] = {(void*)&h->p, (void*)&n_steps_, (void*)&actions_, (void*)&slab_out, (void*)&final_stats, (void*)&perm_, (void*)&moving_,
                        (void*)&perm_, (void*)&moving_};
        const void* fn = team_kernel(h);
        hipStreamCaptureStatus tcap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s_, &tcap) != hipSuccess) { (void)hipGetLastError(); tcap = hipStreamCaptureStatusNone; }
        const bool chained = tcap == hipStreamCaptureStatusNone && h->device >= 0 && h->device < kMaxDevices;
        std::unique_lock<std::mutex> chain(g_team_chain_lock, std::defer_lock);
        if (chained) {                               // (see g_team_chain: the previous team grid of this device has drained)
            chain.lock();
            hipEvent_t& ev = g_team_chain[h->device];
            if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev = nullptr; }
            if (ev && hipStreamWaitEvent(s_, ev, 0) != hipSuccess) (void)hipGetLastError();      // (a never-recorded event: no wait)
        }
        hipError_t le = h->team_coop ? hipLaunchCooperativeKernel(fn, grid, block, argv, 0, s_) : hipLaunchKernel(fn, grid, block, argv, 0, s_);
        if (le != hipSuccess && h->team_coop) {      // (e.g. under stream capture): the occupancy check still holds for a plain launch
            (void)hipGetLastError();
            h->team_coop = false;
            le = hipLaunchKernel(fn, grid, block, argv, 0, s_);
        }
        if (chained && g_team_chain[h->device] && le == hipSuccess && hipEventRecord(g_team_chain[h->device], s_) != hipSuccess) (void)hipGetLastError();
        if (chained) chain.unlock();
        if (le != hipSuccess) return fail(h, EVAC_ERR_HIP, std::string("evac_rollout (team launch): ") + hipGetErrorString(le));
    } else if (h->cu_wide || h->cu_wide4) {
        // one-wave envs, batch >= 16 envs per CU (or four-wave envs, >= 4 per CU): CU-wide workgroups, envs dealt to the SIMDs by
        // load when a schedule scratch is bound.  Launch g reads perm[g & 1], leaves its loads in moving[g & 1] and -- workgroup 0,
        // which carries the lightest envs, before it starts stepping (rollout_body) -- deals perm[(g + 1) & 1] for the next launch
        // from the loads launch g - 1 left in moving[(g - 1) & 1]: no launch is spent on sorting, and no buffer is read and
        // written by the same launch.
        using FW = evac::Wave<1, 1024>;
        using FW4 = evac::Wave<4, 1024>;
        hipStream_t s_ = (hipStream_t)stream;
        const int E = h->p.n_envs;
        // (while the stream is being captured into a hipGraph the host-side generation must not decide what the graph contains:
        // a captured launch runs under the deal at hand, deals nothing and does not advance the generation -- any permutation
        // gives the same results; call evac_reschedule outside the graph to refresh the deal)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s_, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        const bool capturing = cap != hipStreamCaptureStatusNone;
        // Long launches (>= 50 steps): the deal as a launch of its own in front of every rollout launch -- 6.5 us next to >= 100,
        // by the loads the previous launch has just left.  Short launches (the driver's 20 steps): the deal of the NEXT launch is
        // made inside this one (rollout_body: workgroup 0, before it starts stepping), by the loads of the launch before.
        const bool in_kernel = n_steps < 50 && E >= (h->cu_wide4 ? 4 : 16);     // (workgroup 0 sorts: it must be a full one)
        if (h->sched && !capturing && (h->sched_gen < 0 || !in_kernel)) deal_now(h, s_, h->sched_gen < 0);
        const int g_ = h->sched_gen;
        const bool dealt = h->sched && g_ >= 0;                                // (never dealt yet, e.g. a first launch under capture: identity)
        const bool deals = dealt && !capturing && in_kernel;
        const int32_t* perm = dealt ? h->sched + (2 + (g_ & 1)) * E : nullptr;
        int32_t* moving = h->sched ? h->sched + (dealt ? (g_ & 1) : 0) * E : nullptr;
        const int32_t* deal_loads = deals ? h->sched + ((g_ + 1) & 1) * E : nullptr;
        int32_t* deal_perm = deals ? h->sched + (2 + ((g_ + 1) & 1)) * E : nullptr;
        if (dealt && !capturing) h->sched_gen = g_ + 1;
#define EVAC_CUWIDE_ARGS h->p, (int)n_steps, (const float2*)actions, slab_out, final_stats, (const int*)perm, (int*)moving, (const int*)deal_loads, (int*)deal_perm
        if (h->cu_wide4) {
            const dim3 grid4((unsigned)((E + FW4::kEnvsPerBlock - 1) / FW4::kEnvsPerBlock));
            if (h->default_cfg && h->p.obs_pos == EVAC_POS_GRAV)
                hipLaunchKernelGGL((evac::k_rollout_default_config<FW4, true>), grid4, dim3(FW4::kBlock), 0, s_, EVAC_CUWIDE_ARGS);
            else if (h->default_cfg)
                hipLaunchKernelGGL((evac::k_rollout_default_config<FW4, false>), grid4, dim3(FW4::kBlock), 0, s_, EVAC_CUWIDE_ARGS);
            else if (h->p.obs_pos == EVAC_POS_GRAV)
                hipLaunchKernelGGL((evac::k_rollout<FW4, true>), grid4, dim3(FW4::kBlock), 0, s_, EVAC_CUWIDE_ARGS);
            else
                hipLaunchKernelGGL((evac::k_rollout<FW4, false>), grid4, dim3(FW4::kBlock), 0, s_, EVAC_CUWIDE_ARGS);
            return check_launch(h, "evac_rollout");
        }
        const dim3 grid((unsigned)((E + FW::kEnvsPerBlock - 1) / FW::kEnvsPerBlock));
        if (h->default_cfg && h->p.obs_pos == EVAC_POS_GRAV)
            hipLaunchKernelGGL((evac::k_rollout_default_config<FW, true>), grid, dim3(FW::kBlock), 0, s_, EVAC_CUWIDE_ARGS);
        else if (h->default_cfg)
            hipLaunchKernelGGL((evac::k_rollout_default_config<FW, false>), grid, dim3(FW::kBlock), 0, s_, EVAC_CUWIDE_ARGS);
        else if (h->p.obs_pos == EVAC_POS_GRAV)
            hipLaunchKernelGGL((evac::k_rollout<FW, true>), grid, dim3(FW::kBlock), 0, s_, EVAC_CUWIDE_ARGS);
        else
            hipLaunchKernelGGL((evac::k_rollout<FW, false>), grid, dim3(FW::kBlock), 0, s_, EVAC_CUWIDE_ARGS);
#undef EVAC_CUWIDE_ARGS
    } else if (h->default_cfg)
        EVAC_DISPATCH(h, k_rollout_default_config, stream, h->p, (int)n_steps, (const float2*)actions, slab_out, final_stats,
                      (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr);
    else
        EVAC_DISPATCH(h, k_rollout, stream, h->p, (int)n_steps, (const float2*)actions, slab_out, final_stats, (const int*)nullptr,
                      (int*)nullptr, (const int*)nullptr, (int*)nullptr);
    return check_launch(h, "evac_rollout");
}

int evac_observe(evac_handle_t h, float* obs_out, void* stream) {
    EVAC_REQUIRE_BOUND(h, "evac_observe");
    EVAC_JOIN_FIRST(h, stream);
    if (!obs_out) return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_observe: obs_out is NULL");
    DeviceGuard g(h->device);
    EVAC_DISPATCH(h, k_observe, stream, h->p, obs_out);
    return check_launch(h, "evac_observe");
}

static unsigned state_grid(const evac::Params& p) {
    const size_t n = (size_t)p.n_envs * p.n_ped;
    const size_t b = (n + 255) / 256;
    return (unsigned)(b < 2048 ? (b ? b : 1) : 2048);
}

int evac_get_state(evac_handle_t h, float* pos, float* dir, uint8_t* status, float* agent_pos, float* agent_dir,
                   int32_t* now, void* stream) {
    EVAC_REQUIRE_BOUND(h, "evac_get_state");
    EVAC_JOIN_FIRST(h, stream);
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(evac::k_get_state, dim3(state_grid(h->p)), dim3(256), 0, (hipStream_t)stream, h->p, (float2*)pos,
                       (float2*)dir, status, (float2*)agent_pos, (float2*)agent_dir, now);
    return check_launch(h, "evac_get_state");
}

int evac_set_state(evac_handle_t h, const float* pos, const float* dir, const uint8_t* status, const float* agent_pos,
                   const float* agent_dir, const int32_t* now, void* stream) {
    EVAC_REQUIRE_BOUND(h, "evac_set_state");
    EVAC_JOIN_FIRST(h, stream);
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(evac::k_set_state, dim3(state_grid(h->p)), dim3(256), 0, (hipStream_t)stream, h->p,
                       (const float2*)pos, (const float2*)dir, status, (const float2*)agent_pos,
                       (const float2*)agent_dir, now);
    return check_launch(h, "evac_set_state");
}

int64_t evac_norm_state_doubles(evac_handle_t h) { return h ? 3ll * h->p.obs_dim + 4 : -1; }

int evac_norm_init(evac_handle_t h, double* norm_state, void* stream) {
    if (!h) return EVAC_ERR_INVALID_ARGUMENT;
    if (!norm_state) return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_norm_init: norm_state is NULL");
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(evac::k_norm_init, dim3(1024), dim3(256), 0, (hipStream_t)stream, h->p.n_envs, h->p.obs_dim, norm_state);
    return check_launch(h, "evac_norm_init");
}

static unsigned norm_grid(const evac::Params& p) {
    const size_t n = (size_t)p.n_envs * (p.obs_dim + 1);
    const size_t b = (n + 255) / 256;
    return (unsigned)(b < 4096 ? (b ? b : 1) : 4096);
}

int evac_norm_reset(evac_handle_t h, const uint8_t* mask, float* obs, double* norm_state, float obs_clip, float epsilon,
                    void* stream) {
    if (!h) return EVAC_ERR_INVALID_ARGUMENT;
    if (!obs || !norm_state) return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_norm_reset: obs / norm_state is NULL");
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(evac::k_norm_step, dim3(norm_grid(h->p)), dim3(256), 0, (hipStream_t)stream, h->p.n_envs, h->p.obs_dim,
                       obs, (float*)nullptr, (float*)nullptr, (const uint8_t*)nullptr, (const uint8_t*)nullptr, mask, norm_state,
                       0.0f, obs_clip, 0.0f, epsilon, 1);
    return check_launch(h, "evac_norm_reset");
}

int evac_norm_step(evac_handle_t h, float* obs, float* final_obs, float* reward, const uint8_t* terminated,
                   const uint8_t* truncated, double* norm_state, float gamma, float obs_clip, float reward_clip,
                   float epsilon, void* stream) {
    if (!h) return EVAC_ERR_INVALID_ARGUMENT;
    if (!obs || !reward || !terminated || !truncated || !norm_state)
        return fail(h, EVAC_ERR_INVALID_ARGUMENT, "evac_norm_step: obs/reward/terminated/truncated/norm_state must be non-NULL");
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(evac::k_norm_step, dim3(norm_grid(h->p)), dim3(256), 0, (hipStream_t)stream, h->p.n_envs, h->p.obs_dim,
                       obs, final_obs, reward, terminated, truncated, (const uint8_t*)nullptr, norm_state, gamma, obs_clip,
                       reward_clip, epsilon, 0);
    return check_launch(h, "evac_norm_step");
}

#ifdef EVAC_STEP_TIMES
// diagnostic build only: the shader clock right now -- a one-wave kernel on `stream` that reads s_memtime (shader clock) and
// s_memrealtime (100 MHz) at both ends of ~20 us; out2[0] = shader cycles, out2[1] = 100 MHz ticks (device memory, 16 bytes)
__global__ void k_debug_clock(unsigned long long* out2) {
    unsigned long long c0, c1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    unsigned n = 0;
    do { asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory"); } while (r1 - r0 < 2000ull && ++n < (1u << 20));
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    out2[0] = c1 - c0;
    out2[1] = r1 - r0;
}
int evac_debug_clock(unsigned long long* out2_dev, void* stream) {
    hipLaunchKernelGGL(k_debug_clock, dim3(1), dim3(64), 0, (hipStream_t)stream, out2_dev);
    return hipGetLastError() == hipSuccess ? EVAC_OK : EVAC_ERR_HIP;
}
int evac_debug_step_times(unsigned long long* out2048) {
    if (hipMemcpyFromSymbol(out2048, HIP_SYMBOL(g_step_times), 16 * 128 * sizeof(unsigned long long)) != hipSuccess) return EVAC_ERR_HIP;
    return EVAC_OK;
}
#endif
#ifdef EVAC_STEP_TIMES
int evac_debug_launch_marks(unsigned long long* out8192) {
    if (hipMemcpyFromSymbol(out8192, HIP_SYMBOL(g_launch_marks), 64 * 2 * 16 * 8 * sizeof(unsigned long long)) != hipSuccess) return EVAC_ERR_HIP;
    return EVAC_OK;
}
#endif
#ifdef EVAC_STEP_TIMES
int evac_debug_launch_span(unsigned long long* out32768) {
    if (hipMemcpyFromSymbol(out32768, HIP_SYMBOL(g_launch_span), 64 * 256 * 2 * sizeof(unsigned long long)) != hipSuccess) return EVAC_ERR_HIP;
    return EVAC_OK;
}
#endif
#ifdef EVAC_STAMP_WAVES
int evac_debug_stamp_block(int block, unsigned long long* slowest) {      // set the reporting workgroup; read and clear the slowest-workgroup word
    unsigned long long z = 0;
    if (slowest && hipMemcpyFromSymbol(slowest, HIP_SYMBOL(g_slowest), 8) != hipSuccess) return EVAC_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_slowest), &z, 8) != hipSuccess) return EVAC_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_block), &block, 4) != hipSuccess) return EVAC_ERR_HIP;
    return EVAC_OK;
}
int evac_debug_wave_stamps(unsigned long long* out256) {
    if (hipMemcpyFromSymbol(out256, HIP_SYMBOL(g_wave_stamps), 256 * sizeof(unsigned long long)) != hipSuccess) return EVAC_ERR_HIP;
    return EVAC_OK;
}
#endif
#ifdef EVAC_STAMP
// diagnostic build only: read and clear the per-phase cycle sums
int evac_debug_stamps(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return EVAC_ERR_HIP;
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return EVAC_ERR_HIP;
    return EVAC_OK;
}
#endif

}  // extern "C"
