/*
 * libevac -- MI355X (gfx950) implementation of the cinemere/evacuation env step path.
 *
 * C ABI: plain pointers and sizes, no C++/torch types.  This is the drop-in boundary: the
 * reference is pure Python and has no FFI of its own, so each entry point below names the
 * reference *Python* interface it replaces (file:line under /root/reference).  INTEGRATION.md
 * shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every function returns 0 on success or a negative evac_status_t; nothing throws;
 *  - the CALLER owns every device buffer (PyTorch tensors in our host code); the library owns
 *    only a host-side copy of the config, the Philox key and the bound pointers;
 *  - all work is enqueued on the caller's stream (a hipStream_t passed as void*); no call
 *    synchronises or allocates device memory, so calls can be captured into a hipGraph;
 *  - one handle per (device, set of state buffers); a handle is not thread-safe, distinct
 *    handles are independent.
 *
 * Kernel-family selection is a CREATE-TIME OPTION (evac_options_t, evac_create_ex); every field defaults to -1 = automatic.  The
 * environment switches below are kept as DIAGNOSTIC OVERRIDES only (A/B runs of an unmodified caller): a variable that is set wins
 * over the option.  EVAC_SUBWAVE=0 selects the one-wave-per-env
 * kernels also for N <= 32 (default: 4 envs per wave for N <= 16, 2 for N <= 32; same results, see
 * tests/test_gpu_parity.py::test_subwave_kernels_match_one_wave_per_env); EVAC_CELLS=1 / 0 forces the cell-list
 * kernels on (for every N > 64) / off (default: N > 512; the all-pairs kernels give the same neighbour sets); EVAC_CU_WIDE=1 / 0
 * forces / forbids the CU-wide rollout workgroups of one-wave envs (default: batches of >= 16 envs per CU); EVAC_TEAM=0 / 2 / 4 / 8 / 16
 * forbids / forces the team rollout kernels of rooms of more than 512 pedestrians (default: as many CUs per env as the batch leaves
 * free; not under EVAC_CELLS); all of them give bit-identical results.
 * EVAC_SPECIALIZE=0 keeps handles of the reference's default configuration (enslaving_degree 1, |noise_coef| <= 0.4, alpha 3 gravity
 * observation or rel + ohe Box, no wall termination) on the generic kernels instead of the k_*_default_config instantiations in
 * which those uniform parameters are compile-time constants (bit-identical, tests/test_gpu_schedule.py).
 * The Python host honours EVAC_WORKSPACE=0 (no workspace) and
 * EVAC_LIB=<path> to load a profiling build of this library instead of evacuation_amd/libevac.so.
 *
 * Device layouts (row-major, E = num_envs, N = n_ped)
 *    ped    float [E][N][4]   (x, y, dir_x, dir_y)      pedestrians.py:17-19
 *    status uint8 [E][N]      1 VISCEK 2 FOLLOWER 3 EXITING 4 ESCAPED   statuses.py:16-27
 *    agent  float [E][4]      (x, y, dir_x, dir_y)      area.py:12-30
 *    clock  int32 [E][4]      (now, n_resets, total_steps, 0)           area.py:42-59
 *    acc    float [E][4]      (episode_reward, episode_intrinsic_reward, episode_status_reward, 0)
 *                                                                       env.py:65-67,168-170
 *    obs    float [E][D]      D = evac_obs_dim(); layout per observation mode below
 *
 * Observation layouts (all f32; D floats per env)
 *    positions=grav                  [agent(2), grad_potential_exit(2), grad_potential_pedestrians(2)]
 *                                    (gymnasium Dict key order, i.e. what FlattenObservation yields)
 *    type=Box                        [(N+2)][C] rows agent, exit, pedestrians; C = 2 / 3 (cat) / 6 (ohe)
 *    type=Dict, positions=abs|rel    [agent(2), exit(2), pedestrians_positions(2N), pedestrians_statuses(4N | N | 0)]
 */
#ifndef EVAC_H
#define EVAC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVAC_VERSION 150          /* 0.1.5: evac_options_t / evac_create_ex, rollouts of one handle as two concurrent kernels (parts), evac_join */
#define EVAC_MAX_PEDESTRIANS 1024 /* one workgroup (<=16 waves) per env */

typedef enum evac_status {
    EVAC_OK = 0,
    EVAC_ERR_INVALID_ARGUMENT = -1,
    EVAC_ERR_NOT_BOUND = -2,
    EVAC_ERR_UNSUPPORTED = -3, /* e.g. positions=grav with type=Box: wrappers/config.py:79-80 raises NotImplementedError */
    EVAC_ERR_HIP = -4,
    EVAC_ERR_NO_DEVICE = -5,
    EVAC_ERR_TEAM_ABORTED = -6 /* an earlier team rollout lost a member, or a chained rollout (evac_options_t.chain) waited in vain for an
                                  env's state: see evac_team_error / evac_team_clear_error */
} evac_status_t;

enum { EVAC_POS_ABS = 0, EVAC_POS_REL = 1, EVAC_POS_GRAV = 2 };   /* wrappers/config.py:19-24 */
enum { EVAC_STAT_NO = 0, EVAC_STAT_OHE = 1, EVAC_STAT_CAT = 2 };  /* wrappers/config.py:26-29 */
enum { EVAC_TYPE_DICT = 0, EVAC_TYPE_BOX = 1 };                   /* wrappers/config.py:31-33 */

/* Parameter block = the fields of EnvConfig (src/env/env/config.py:11-59) and EnvWrappersConfig
 * (src/env/wrappers/config.py:12-38) that enter the arithmetic.  Same names, same meaning. */
typedef struct evac_config {
    int32_t number_of_pedestrians;
    float width;
    float height;
    float step_size;
    float noise_coef;
    float eps;
    float enslaving_degree;
    int32_t is_new_exiting_reward;
    int32_t is_new_followers_reward;
    float intrinsic_reward_coef;
    int32_t is_termination_agent_wall_collision;
    float init_reward_each_step;
    int32_t max_timesteps;
    /* observation wrappers */
    int32_t positions; /* EVAC_POS_*  */
    int32_t statuses;  /* EVAC_STAT_* */
    int32_t type;      /* EVAC_TYPE_* */
    float alpha;       /* GravityEncoding alpha (gravity_encoding.py:41-57) */
    /* 0 (default): bit-faithful to the reference, where a zero heading (0/0, area.py:101) poisons every
     * FOLLOWER/VISCEK pedestrian with NaN.  1: a zero heading contributes nothing (non-reference). */
    int32_t nan_guard;
    /* gym.wrappers.ClipAction of the trainer's wrapper chain (rpo_agent.py:27): clip actions to [-1,1] first */
    int32_t clip_action;
} evac_config_t;

/* Written for envs whose episode ended this step: the nine keys of the reference's per-episode logging dict
 * (env.py:115-125) plus the episode counter Time.n_episodes (area.py:49-51).  Ten 4-byte words; the two
 * counters are integers (a float would lose steps beyond 2^24). */
typedef struct evac_episode_stats {
    float episode_reward;
    float episode_length;
    float episode_intrinsic_reward;
    float episode_status_reward;
    float escaped_pedestrians;
    float exiting_pedestrians;
    float following_pedestrians;
    float viscek_pedestrians;
    int32_t overall_timesteps; /* Time.overall_timesteps (area.py:47,55): steps of this env since creation */
    int32_t n_episodes;        /* Time.n_episodes when the episode ended (resets so far) */
} evac_episode_stats_t;
#define EVAC_EPISODE_STATS_WORDS 10

typedef struct evac_handle* evac_handle_t;

int evac_version(void);
const char* evac_status_string(int status);
/* Last error message of a handle (or of the last failed evac_create when h == NULL). */
const char* evac_last_error(evac_handle_t h);

/* Device-free helpers: validate a config (same errors as evac_create) / floats per env of its
 * observation.  Mirror the checks of wrappers/config.py:76-82. */
int evac_config_validate(const evac_config_t* cfg);
int64_t evac_config_obs_dim(const evac_config_t* cfg);

/* Replaces EvacuationEnv.__init__ + EnvWrappersConfig.wrap_env for a batch of `num_envs` independent
 * envs (env.py:41-84, wrappers/config.py:46-93, src/env/__init__.py:18-21).  `seed` keys the Philox
 * streams; env e uses stream id `env_id_offset + e`, so a sharded run reproduces the single-GPU run. */
int evac_create(const evac_config_t* cfg, int32_t num_envs, int32_t device, uint64_t seed,
                uint64_t env_id_offset, evac_handle_t* out);

/* Create-time options: WHICH kernels a handle launches, never WHAT they compute -- every combination gives bit-identical results
 * (tests/test_gpu_variants_sweep.py, tools/soak_variants.py).  No reference analogue (the reference steps its envs one after another
 * in Python, rpo_agent.py:123-126).  Every field: -1 = automatic (what evac_create does).
 *   subwave     0: one wave per env also for N <= 32; 1: 4 / 2 envs per wave for N <= 16 / 32 (automatic)
 *   cells       1 / 0: the 16 x 16 cell-list kernels for every N > 64 / never (automatic: N > 512)
 *   cu_wide     1 / 0: CU-wide rollout workgroups (16 one-wave or 4 four-wave envs per 1024-thread workgroup) always / never
 *               (automatic: batches of 16..64 one-wave envs, 4..16 four-wave envs per CU)
 *   team        0 / 2 / 4 / 8 / 16: workgroups (CUs) per env of the team rollout kernels, N > 512 (automatic: what the batch leaves free)
 *   specialize  0: never the k_*_default_config instantiations (automatic: when the configuration matches)
 *   parts       1 / 2: evac_rollout issues the batch as ONE kernel on the caller's stream / as TWO half-batch kernels on two streams
 *               the handle owns (see evac_join).  -1: 2 where it pays (CU-wide handles whose halves still fill their CUs), else 1.
 *               evac_create() -- the entry point existing callers use -- always takes 1: its stream contract is unchanged.
 *   team_coop   1: team grids are launched with hipLaunchCooperativeKernel (3-4 % slower; automatic: plain launches)
 *   team_fault  1: fault injection for tests (the team grid is launched one workgroup short; chained launches: the last env's
 *               generation word is never published)
 *   chain       1: CHAINED rollout launches (CU-wide handles with a bound workspace; see below); 0: never; -1: where it
 *               pays (those handles).  Wins over `parts`.  evac_create() always takes 0, like parts = 1.
 *               2: ONE PERSISTENT KERNEL PER JOIN (see below; on request only -- the kernel holds the device until evac_join);
 *               (the CU-wide kernels of one- and four-wave envs, and the team kernels of rooms of more than 512 pedestrians);
 *               where it cannot be had (no such family, a batch of more workgroups than the device has CUs, no large BAR) the
 *               handle falls back to 1, then 0: evac_get_options says what it became. */
typedef struct evac_options {
    int32_t subwave, cells, cu_wide, team, specialize, parts, team_coop, team_fault, chain;
} evac_options_t;
#define EVAC_OPTIONS_AUTO {-1, -1, -1, -1, -1, -1, -1, -1, -1}
/* evac_create with options (NULL: all automatic, parts = 1: exactly evac_create). */
int evac_create_ex(const evac_config_t* cfg, int32_t num_envs, int32_t device, uint64_t seed, uint64_t env_id_offset,
                   const evac_options_t* options_or_null, evac_handle_t* out);
/* The options a handle ended up with (automatic choices resolved, diagnostic environment overrides applied). */
int evac_get_options(evac_handle_t h, evac_options_t* out);
int evac_destroy(evac_handle_t h);

/* Handles created with parts = 2 (no reference analogue).  A rollout launch lasts as long as the heaviest env it carries, and
 * between two launches of one stream lie the queue's kernel boundary and the kernel's prologue; the envs of a batch do not depend
 * on each other, only consecutive launches of the SAME env do.  So such a handle keeps two streams of its own (on two hardware
 * queues) and evac_rollout enqueues envs [0, E/2) and [E/2, E) as two kernels, one per stream: each half waits only for ITS
 * heaviest workgroup and its boundary and prologue run under the other half's steps (N = 60 x 4096 envs, 20 steps per launch:
 * +3.4-5 %).  One call, one slab [T][E][D+3], the same bits as parts = 1 (the Philox streams are keyed by the
 * global env id).  Stream contract of such a handle:
 *   - evac_rollout(h, ..., stream): the kernels go to the handle's own streams, and `stream` does NOT wait for them --
 *     consecutive evac_rollout calls must not meet at a common point, or the halves would run in lock-step again.  The own
 *     streams are put behind what `stream` holds (an event recorded on it) at the FIRST evac_rollout after an evac_join or any
 *     other call on the handle, and at every call that passes `actions` (new inputs); a run of RandomAgent rollout calls
 *     without a join between them is ordered after the work `stream` held at its first call only (a wait per launch costs a
 *     barrier packet, more than the form gains) -- so: join before `stream` reuses or refills anything those launches touch;
 *   - evac_join(h, stream): `stream` waits for everything the handle's own streams have been given so far.  Call it before
 *     anything on `stream` (or the host, after synchronising `stream`) consumes a slab, and before ending a stream capture;
 *   - every other call on the handle (evac_reset, evac_step*, evac_observe, evac_get_state, evac_set_state, evac_reschedule,
 *     rollouts with capture / recorded actions / injected noise) joins first by itself and runs as one kernel on `stream`.
 * evac_num_parts: 1 or 2.  evac_part_stream: the hipStream_t of part k (for timing events and profilers; NULL if k is out of
 * range or the handle owns no streams).  evac_join on a handle that owns no streams is a no-op.
 *
 * CHAINED LAUNCHES (chain = 1; no reference analogue).  With the driver's 20 steps per call a launch lasts as long as its
 * heaviest env, and in the steady state of a batch -- episode phases spread out by early terminations -- every launch carries
 * freshly reset, dense envs: most CUs idle behind them for a quarter of every launch (tools/steady_probe.py: 49.6 us per
 * 4096-env round where the same kernel sustains 36 with every CU kept busy).  Only consecutive launches of the SAME env depend on
 * each other.  A chained handle therefore sends its rollout launches to its two streams ALTERNATELY (launch g to stream g & 1)
 * and orders them per env on the device: a generation word per env in the workspace -- launch g waits for ready[env] == g before
 * it loads the env's state and publishes g + 1 behind its state stores (device-scope accesses; bounded waits) -- so launch
 * g + 1's workgroups take the CUs launch g's light workgroups leave and each wave starts the moment ITS env is ready.  At most
 * two launches overlap (launch g + 2 follows launch g in its stream).  Same bits as every other form.  Stream contract: exactly
 * that of parts = 2 (evac_join; everything that is not a plain rollout joins by itself and restarts the chain behind it).  A wait
 * that times out (it cannot, unless a launch is lost: every launch a wave waits for was dispatched in full before its own) voids
 * the run like a lost team member: the handle's error word is raised, evac_* calls return EVAC_ERR_TEAM_ABORTED until
 * evac_team_clear_error(), and the handle issues plain launches from then on.
 *
 * ONE PERSISTENT KERNEL PER JOIN (chain = 2; no reference analogue).  What a launch adds to the heaviest env's sequence of steps --
 * the queue's boundary, the dispatch of a 1024-thread workgroup (3.4 us on a CU that has just become free), the prologue, the
 * state's way through memory -- is a fifth of a 20-step call even when launches are chained.  With chain = 2 the first evac_rollout
 * after a join starts the rollout kernel on the handle's stream and every call (that one included) becomes a 64-byte COMMAND
 * {steps, slab, episode records} that the host writes, through the PCIe BAR, into a ring in uncached device memory; the resident
 * kernel runs the call's steps into the call's slab and takes the next command with the state still in registers.  evac_join
 * posts STOP: the waves store their state and the kernel ends.  Every call still computes exactly its n_steps into its own
 * buffers; same bits as every other form.  Stream contract: that of parts = 2.  What is particular to this form:
 *   - while it has commands the kernel HOLDS the CUs it runs on (all of them for a batch that fills the device): other kernels of
 *     the process run when it has left;
 *   - a kernel that finds no command for ~150 us LEAVES by itself: every wave stores its env's state and the index of the command
 *     it was waiting for, and the next evac_rollout -- or the join -- starts a kernel that takes every env up where it stopped.
 *     So a caller may pause, wait for the device without having joined, or turn to another handle (also another one with
 *     chain = 2): nothing hangs and nothing starves, there is no bound to exceed and no error to raise; what a gap of more than
 *     ~150 us between two calls costs is a kernel start;
 *   - evac_join posts STOP and enqueues, behind the resident kernel, a FINISHER kernel that runs whatever an env has not run up to
 *     the STOP (nothing, normally: it ends at once) -- the join itself only enqueues, like every call;
 *   - calls with `actions` (and the diagnostic faces) join and run as one kernel on `stream`; more than ~1000 calls without a
 *     join make the library stop the kernel, wait for it on the host and start the next one.
 * evac_own_streams: 0, or 2 for handles with parts = 2, chain = 1 or chain = 2. */
int evac_join(evac_handle_t h, void* stream);
/* The NEXT evac_rollout call puts the handle's own streams behind what its `stream` holds at that moment, as the first call after a
 * join does: for a caller who, between two rollout calls and without a join, gave `stream` work the coming launches must follow -- a
 * freshly allocated or refilled output buffer (a stream-ordered allocator hands out memory that kernels still queued on `stream`
 * may be using), new workspace contents.  The Python host's rollout(), which allocates its outputs, calls it every time. */
int evac_order_next_rollout(evac_handle_t h);
int32_t evac_num_parts(evac_handle_t h);
int32_t evac_own_streams(evac_handle_t h);
void* evac_part_stream(evac_handle_t h, int32_t part);

/* Floats per env in the observation buffer for this handle's observation mode. */
int64_t evac_obs_dim(evac_handle_t h);
int32_t evac_num_envs(evac_handle_t h);
/* Name of the kernel instantiation this handle's evac_step (rollout == 0) / evac_rollout (rollout != 0) launches,
 * e.g. "k_rollout<1 wave/env, grav>": the label bench.py puts next to its roofline numbers (no reference analogue). */
const char* evac_kernel_variant(evac_handle_t h, int32_t rollout);

/* Bind the caller-owned state buffers (device pointers, layouts above). */
int evac_bind_state(evac_handle_t h, float* ped, uint8_t* status, float* agent, int32_t* clock, float* acc);

/* EvacuationEnv.reset (env.py:106-139) for every env, or for those with mask[e] != 0.
 * draws_or_null: float [E][N][4] of U(-1,1) values (pos.x, pos.y, dir.x, dir.y) replacing the Philox
 * draws (pedestrians.py:17-18) -- the injection mode used by the parity tests.
 * obs_out_or_null: reset observation, float [E][D] (only rows of reset envs are written). */
int evac_reset(evac_handle_t h, const uint8_t* mask_or_null, const float* draws_or_null,
               float* obs_out_or_null, void* stream);

/* EvacuationEnv.step + observation wrappers (env.py:141-171, area.py:76-210, statuses.py:29-48,
 * reward.py:19-47, gravity_encoding.py:8-81, wrappers.py:8-96) for the whole batch, one launch.
 *   actions            float [E][2]  (device)
 *   noise_or_null      float [E][N]  per-pedestrian angular noise (injection mode); NULL = Philox
 *   obs_out            float [E][D]
 *   reward_out         float [E];  terminated_out / truncated_out  uint8 [E]
 *   autoreset != 0     finished envs are reset in the same launch (gymnasium-0.29 SyncVectorEnv
 *                      semantics, rpo_agent.py:193-203): obs_out then holds the reset observation,
 *                      final_obs_or_null [E][D] the terminal one, final_stats_or_null [E] the episode record
 */
int evac_step(evac_handle_t h, const float* actions, const float* noise_or_null, float* obs_out,
              float* reward_out, uint8_t* terminated_out, uint8_t* truncated_out, int32_t autoreset,
              float* final_obs_or_null, evac_episode_stats_t* final_stats_or_null, void* stream);

/* T consecutive steps in ONE launch with the env state held in registers/LDS (the trainer's rollout
 * loop rpo_agent.py:180-203 with the policy replaced by caller-provided or RandomAgent actions,
 * random_agent.py:8-9).  Always autoresets.  Buffers are time-major:
 *   actions_or_null     float [T][E][2]   NULL = draw U(-1,1)^2 on device (Philox)
 *   actions_out_or_null float [T][E][2]   records the actions actually used
 *   slab_out            float [T][E][D+3] = [obs(D) | reward | terminated (0/1) | truncated (0/1)] per env-step:
 *                       one packed record (what the trainer's rollout buffers and the all-gather consume)
 *   final_stats_or_null [T][E] (rows of envs that finished at step t)
 *   noise_or_null       float [T][E][N]   per-pedestrian angular noise replacing the Philox draw (injection mode, as
 *                       evac_step's; served by the diagnostic kernel face, like capture / actions_out)
 *   capture_or_null     float [T][capture_envs][N+1][3]: trajectory capture for rendering (the memory that
 *                       Pedestrians.save / Agent.save keep, pedestrians.py:33-35, area.py:32-33, consumed by
 *                       save_animation env.py:241-324): rows 0..N-1 = (x, y, status) of every pedestrian of the
 *                       first `capture_envs` envs after the step (before an autoreset), row N = (leader x, y, 0) */
int evac_rollout(evac_handle_t h, int32_t n_steps, const float* actions_or_null, float* actions_out_or_null,
                 float* slab_out, evac_episode_stats_t* final_stats_or_null, int32_t capture_envs,
                 float* capture_or_null, const float* noise_or_null, void* stream);

/* Optional workspace of evac_rollout (no reference analogue: the reference steps its envs one after another,
 * rpo_agent.py:123-126).  `workspace`: evac_workspace_bytes(h) bytes on the device, 256-byte aligned, ZERO-INITIALISED by the
 * caller, alive and used on one stream at a time like the state buffers.  It holds
 *   - the rollout schedule of large batches of one-wave envs (>= 16 envs per CU; four-wave envs: >= 4 per CU):
 *     moving[2][E] | perm[2][E] int32 -- rollout launch g of the handle runs its envs in the order perm[g & 1], leaves the
 *     pedestrians still moving of each env in moving[g & 1] and, inside the same launch (its first workgroup, which carries the
 *     lightest envs, before it starts stepping), deals the envs to the SIMDs for launch g + 1 by the loads launch g - 1 left:
 *     perm[(g + 1) & 1].  evac_schedule_generation returns g (the number of such launches so far; -1: no schedule, or no deal
 *     yet).  A launch captured into a hipGraph runs under the deal at hand and deals nothing;
 *   - the exchange areas of the team kernels (513..1024 pedestrians, few envs: 2 / 4 / 8 / 16 workgroups per env).
 * Performance devices only: results are bit-identical with and without the workspace.  NULL unbinds.
 *
 * Team rollouts need every workgroup of their grid resident at once (the members of a team wait for each other).  evac_rollout
 * checks that with hipOccupancyMaxActiveBlocksPerMultiprocessor (a grid that does not fit runs one workgroup per env instead);
 * EVAC_TEAM_COOP=1 additionally launches with hipLaunchCooperativeKernel (3-4 % slower).  Should a team lose a member all the
 * same, its waits are bounded: the launch ends, that env's state is NOT written back, and the handle's error word -- 64 bytes of
 * host-mapped memory, the one thing besides the config the library allocates -- is raised.  Every later call on the handle
 * (evac_reset / evac_step* / evac_rollout / evac_observe / evac_get_state / evac_set_state) then returns EVAC_ERR_TEAM_ABORTED
 * without touching the device, until evac_team_clear_error(); the handle uses one workgroup per env from then on.
 * evac_team_error: synchronises the device, then reports the error word (non-zero: the outputs of an earlier launch are void).
 * evac_team_error_nosync: the same word as it stands, without synchronising -- for a caller that has just waited for its
 * stream itself (a pipelined consumer of rollout slabs: wait for the launch, read the word, THEN trust the slab; a launch
 * that aborts returns EVAC_OK when it is enqueued, and so do the launches queued behind it, and a hipGraph replay goes past
 * every host-side check).
 * Team grids of one device run ONE AT A TIME, whatever handles and streams they come from (a device-side wait on the previous
 * team launch's event): two grids in flight could each be resident in part and wait for CUs the other holds.  Launches under
 * stream capture cannot join that chain -- do not replay a graph with team launches beside another team launch; and another
 * PROCESS sharing the GPU is out of reach: run it with EVAC_TEAM=0. */
int64_t evac_workspace_bytes(evac_handle_t h);
int evac_bind_workspace(evac_handle_t h, void* workspace_or_null, int64_t bytes);
/* Deal the envs to the SIMDs NOW (a launch of its own) by the most recent loads the workspace holds, moving[(g - 1) & 1], into
 * the permutation the next rollout launch reads: for callers that restored a state + workspace snapshot and want the next
 * launch to run under that deal.  No-op without a schedule. */
int evac_reschedule(evac_handle_t h, void* stream);
int32_t evac_schedule_generation(evac_handle_t h);
int evac_team_error(evac_handle_t h, int32_t* out);
int evac_team_error_nosync(evac_handle_t h, int32_t* out);
int evac_team_clear_error(evac_handle_t h);

/* The all-gather of the returned observation batch across the ranks of an env-sharded run (BASELINE.json north_star; the
 * reference steps all envs in one process, src/agents/rpo_agent.py:123-126, and has no analogue) as PEER STORES over xGMI,
 * without a library collective: columns [0, take_words) of this rank's record slab src[rows][row_words] -- the rollout's
 * packed [obs | reward | terminated | truncated] records; take_words = obs_dim picks the observation -- are written to
 * slice my_rank of every rank's buffer peer_dst[r] = [world][rows][take_words] (r = my_rank: the local buffer; the others:
 * the peers' buffers mapped into this process, e.g. through hipIpcOpenMemHandle).  One launch on `stream`, all peers at once,
 * wgs_per_peer workgroups of 256 threads each (0: 8); compiled to fit beside a running rollout kernel (evac_gather.h).  A
 * consumer of a buffer needs every rank's launch to have completed (a rendezvous of the ranks).  No handle: any device
 * pointers.  EVAC_ERR_INVALID_ARGUMENT: NULL pointers, world > 16, rows * take_words >= 2^31. */
int evac_peer_gather(const float* src, int64_t rows, int32_t row_words, int32_t take_words, float* const* peer_dst, int32_t world,
                     int32_t my_rank, int32_t wgs_per_peer, void* stream);

/* State exchange in the reference's own shapes (needed for parity tests, checkpoints):
 * pos/dir float [E][N][2], status uint8 [E][N], agent_pos/agent_dir float [E][2], now int32 [E]. */
int evac_get_state(evac_handle_t h, float* pos, float* dir, uint8_t* status, float* agent_pos,
                   float* agent_dir, int32_t* now, void* stream);
int evac_set_state(evac_handle_t h, const float* pos, const float* dir, const uint8_t* status,
                   const float* agent_pos, const float* agent_dir, const int32_t* now, void* stream);

/* Observation of the current state without stepping (EvacuationEnv._get_observation through the
 * wrapper chain, env.py:98-104). */
int evac_observe(evac_handle_t h, float* obs_out, void* stream);

/* Algorithmic HBM bytes of one env-step for this handle (32N+62 for grav, 56N+86
 * for Box+ohe, ...).  Used by bench.py for roofline accounting. */
int64_t evac_algorithmic_bytes_per_env_step(evac_handle_t h);

/* ---- The trainer's per-env wrapper chain on device (rpo_agent.py:24-33) ----
 * NormalizeObservation -> clip(obs, +-obs_clip) -> NormalizeReward(gamma) -> clip(reward, +-reward_clip), one set
 * of running statistics per env (gymnasium wrappers/normalize.py RunningMeanStd, float64), applied IN PLACE to
 * the outputs of evac_step / evac_reset.  ClipAction is evac_config_t.clip_action; FlattenObservation is the
 * flat observation layout; RecordEpisodeStatistics is evac_episode_stats_t.
 *   norm_state  double [E][evac_norm_state_doubles(h)] = obs_mean[D] | obs_var[D] | obs_count[D] | ret_mean |
 *               ret_var | ret_count | returns   (caller-owned, initialise with evac_norm_init)
 * evac_norm_step: for envs that finished (terminated|truncated, same-step autoreset) final_obs (the terminal
 * observation, may be NULL) is normalised and counted first, then obs (the reset observation) -- the order in
 * which SyncVectorEnv calls the wrapped step() and reset(). */
int64_t evac_norm_state_doubles(evac_handle_t h);
/* evac_step with the chain FUSED into the step kernel (one launch; the form the trainer's loop uses): the outputs come
 * out normalised and clipped, norm_state is updated, final_obs (terminal observation of finished envs) is counted
 * before obs.  Same results as evac_step followed by evac_norm_step, bit for bit. */
int evac_step_normalized(evac_handle_t h, const float* actions, const float* noise_or_null, float* obs_out,
                         float* reward_out, uint8_t* terminated_out, uint8_t* truncated_out, int32_t autoreset,
                         float* final_obs_or_null, evac_episode_stats_t* final_stats_or_null, double* norm_state,
                         float gamma, float obs_clip, float reward_clip, float epsilon, void* stream);
int evac_norm_init(evac_handle_t h, double* norm_state, void* stream);
int evac_norm_reset(evac_handle_t h, const uint8_t* mask_or_null, float* obs, double* norm_state, float obs_clip,
                    float epsilon, void* stream);
int evac_norm_step(evac_handle_t h, float* obs, float* final_obs_or_null, float* reward, const uint8_t* terminated,
                   const uint8_t* truncated, double* norm_state, float gamma, float obs_clip, float reward_clip,
                   float epsilon, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVAC_H */
