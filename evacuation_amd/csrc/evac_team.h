// Device code of libevac, part 4: the TEAM family -- one env of 513..1024 pedestrians spread over K workgroups on K
// compute units (K = 2, 4, 8 or 16), for batches that would otherwise leave most of the chip idle (BASELINE config 5 runs 32
// envs of 1024 pedestrians per GPU: with one workgroup per env 224 of 256 CUs have nothing to do, and mid-episode, when the
// crowd has flocked into one corner of the room, the neighbour sum of ONE env is ~350 k true pairs -- tools/row_lengths.py).
//
// Member k of a team owns pedestrians [k P, (k+1) P), P = 1024 / K, in the lanes of its first P / 64 waves ("ped waves");
// all 16 waves of the workgroup share the member's part of the pair work.  The members meet ONCE per step through global
// memory (device-scope write-through stores and device-scope loads, no cache flush or invalidation, and NO counter: every
// published 8-byte half carries the tag of its round and validates itself -- exchange() below; round 2's
// store / acknowledgement / counter / spin / load protocol is tools/experiments/r04_team_counter_exchange.patch;
// tools/microbench/team_sentinel.hip and team_barrier.hip time both in isolation).  What travels in that round:
//   * the step's reduction: every ped wave publishes the same 32-byte record as a wave of Cells<16> leaves in LDS; after the
//     barrier three helper waves of every member fold the 16 records with the same DPP tree (float sums, packed counts and
//     the prefix of the segment counts are independent chains), so rewards / observations / flags are bit-identical to the
//     one-workgroup kernels' and every member takes the same decisions (autoreset, termination) without further talk;
//   * the NEXT step's tile: the head of the next step (pre_pair: escaped pin, exiting heading, unit heading) depends on the
//     pedestrian's own post-step state only, so every ped wave already publishes its moving pedestrians (position x 2^40,
//     integer heading) in the wave's 64-entry segment; every member gathers all 16 segments, compacted, into its LDS
//     tile -- the columns of the next step's distance matrix.  (The first step of a launch and the step after an
//     autoreset run the same exchange on their own.)
// The rows are the member's own pedestrians that need one (step_env: needs_row), compacted per ped wave.  MANY rows (early in
// an episode): two ped waves per pass (two rows per lane -- or one, when the rows of both fit the 64 lanes), each of the 16 waves
// takes 1/16 of the columns (wave-uniform ds_read_b128 broadcasts).  FEW rows (<= kFewRows: most of an episode under enslaving_degree 1, when only the VISCEK
// pedestrians need one): the sweep is transposed -- the rows are dealt to the 16 waves, the lanes hold the columns.  Either
// way the partial sums of a row meet in its LDS accumulator by integer atomics.  Heading sums are INTEGERS
// (taken eight at a time in packed f32 -- exact -- since round 4), so the result does not depend on how the pairs were split -- it is bit-identical to the cell-list
// kernel's (Cells<16>).
// Everything else is the common step body (step_env) and rollout scaffolding (rollout_body); waves without pedestrians
// ("helper" waves) skip the per-pedestrian arithmetic.
//
// The members of a team must be resident together (they poll each other's slots).  The host (evac_rollout, evac_api.hip)
// checks with hipOccupancyMaxActiveBlocksPerMultiprocessor that the whole grid fits the device at once (one 1024-thread
// workgroup per CU) -- otherwise the handle runs the one-workgroup-per-env kernels.  A kernel of another stream (the all-gather
// of the sharded env) can delay a member until it ends, never starve it, and the waits outlast that; EVAC_TEAM_COOP=1 launches
// with hipLaunchCooperativeKernel instead, which lets the runtime guarantee co-residency at 3-4 % of the throughput.  A team sits on ONE XCD (workgroup ids congruent
// mod 8 share an XCD -- round-robin dispatch; nothing depends on it but the latency).  Spins are bounded all the same: a
// team that lost a member sets a sticky abort flag, runs to the end without waiting, does NOT write its env's state back
// (the state of that env stays what it was before the launch) and raises the handle's error word -- host-mapped memory that
// the next evac_* call of the handle reads without synchronising and turns into EVAC_ERR_TEAM_ABORTED.
#pragma once

#include "evac_device.h"

namespace evac {

// (measured at C5, profiles/r03_g_c5_ab_few_rows_threshold.txt and r03_g_c5_ab_sentinel_exchange.txt)
constexpr int kTeamFewRows = 32;     // needed rows of a member up to which its sweep is transposed (with the packed-f32 sweeps of round 4: 16-32 flat, 48 -3 %, 64 -8 %, 96 -20 %: profiles/r04_j_c5_few_rows_threshold.txt)
constexpr int kTeamFirstPoll = 12;   // s_sleep units (64 cycles) between the member's own publish and its first poll (round 4, after the sweeps got faster: 12 +1 % over 8, 4 -3 %, 16 -1 %, 24 -5 %)
constexpr int kTeamPollGap = 2;      // ... between two polls
constexpr int kTeamMinPoll = 4;      // round 5: the first poll adapts to where the last round's data arrived (exchange()); never earlier than this

// (store_dev / store_dev_i32 / load_dev / wait_vmem -- the device-scope accesses of the exchange -- are in evac_common.h: the chained
// rollout launches use the same idiom)
__device__ __forceinline__ void lds_add(int* ptr, int v) { (void)__hip_atomic_fetch_add(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// The self-validating exchange, round 5: every 8-byte half of a published entry / record carries the TAG of its round
// (round mod 31, five spare bits of a word that has them: the flag byte of the integer headings, the top bits of the packed counts);
// the host fills the area with 0xff before a launch, which reads as tag 31 -- never a round's.  A slot is fresh when all its tags
// are the round's; what it held before is a few rounds old (a slot is rewritten every second round, or after at most four when
// rounds without a tile or without records intervene), never a multiple of 31.  Round 3's form kept a sentinel in every word of a
// slot between uses and RESET each slot one round after it was read: a second store per entry and step -- 16.5 KB of the 77.9 KB
// of HBM traffic per env-step (profiles/traffic.json of round 4) -- and a third slot set to keep the reset off the next poll; both
// are gone.  (An 8-byte half is the unit the tags protect: positions and float sums have no spare bits, so they travel next to a
// tagged word; global accesses of a lane are not torn below that on this hardware -- and tools/soak_variants.py compares every
// launch of 10^6 steps with the one-workgroup kernels bit for bit.)
// THE INVARIANT this rests on (nothing enforces it at run time): a slot of set (round & 1) that a reader polls in round r
// must not still hold a tag equal to r mod 31 from an EARLIER round, i.e. every slot is rewritten at least once in any 62 consecutive
// rounds of its set's parity.  Today: stage_next publishes a tile entry from every ped lane on every step, reduce_publish a record
// from every ped wave on every step; the only rounds without a tile are the first step of a launch and the step after an autoreset
// (the host refills the area with tag 31 before a launch), so a slot is at most 4 rounds old when it is polled.  A future path that
// could skip publishing for 62 rounds (records-only or tile-only rounds in a row) must widen the tag or reset the slot -- a stale
// slot would validate as fresh without any error.  And the 8-byte half must stay the store granule (16-byte sc1 stores are observed
// untorn below 8 bytes on gfx950; not an architectural guarantee): both are what tools/soak_variants.py watches over.
static_assert((1 << 5) - 1 == 31, "tags are taken mod 31: five bits, the all-ones pattern reserved for the host's 0xff fill");
constexpr int kTeamSets = 2;
constexpr int kTagBits = 5, kTagMod = 31;
constexpr int kEntryTagShift = 24;                          // heading words: 24-bit field | tag << 24 | flags (bits 29, 30)
constexpr int kEntryNull = 1 << 29, kEntryNan = 1 << 30;   // flag bits of an entry's heading-x word: no pedestrian that moves here / NaN heading
constexpr int kCountTagShift = 27;                          // packed-count words: two counts of at most 1024 in bits 0-10 and 16-26 | tag << 27
__device__ __forceinline__ bool tagged(float w, int shift, int tag) { return ((__builtin_bit_cast(int, w) >> shift) & ((1 << kTagBits) - 1)) == tag; }
__device__ __forceinline__ void land(f4& a, f4& b, f4& c) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c)::"memory"); }   // the loads into a, b and c have returned

template <int K_>
struct Team {
    static constexpr int K = K_;
    static_assert(K == 2 || K == 4 || K == 8 || K == 16, "a team has 2, 4, 8 or 16 members");
    static constexpr int WPE = 16;                       // waves per workgroup = ped waves per team
    static constexpr bool kEnvUniform = true, kPace = false, kHelpers = true, kExitLane = false, kPipelined = true, kEnvBarrier = false;
    static constexpr int kBlock = 1024, kThreadsPerEnv = 1024, kEnvsPerBlock = 1;
    static constexpr int P = 1024 / K;                   // pedestrians per member
    static constexpr int PW = P / kWave;                 // ped waves per member
    static constexpr int kPad = 8;
    static constexpr int kFewRows = kTeamFewRows;                  // needed rows of a member up to which the transposed sweep is used (neighbour_sum)
    static constexpr const char* kName = K == 16 ? "16 CUs/env, all pairs over the team's tile" : (K == 8 ? "8 CUs/env, all pairs over the team's tile" : (K == 4 ? "4 CUs/env, all pairs over the team's tile" : "2 CUs/env, all pairs over the team's tile"));

    struct Smem {
        f4 tile[1024 + kPad];                 // the team's moving pedestrians: (X, Y, heading x, heading y as integers)
        float2 rowpos[PW][kWave];             // this member's needed rows, compacted per ped wave
        alignas(8) int acc[PW][kWave][2];                  // heading sums of the needed rows [ped wave][row slot]: integer LDS atomics of all 16 waves, cleared by their reader
        int rows[PW];                         // needed rows per ped wave
        int abort;                            // sticky: a barrier timed out
        alignas(16) f4 red_f;                 // the folded records (three helper waves -> everybody)
        alignas(16) i4 red_i;
        i2 seg[WPE];                          // where segment w of the exchange area lands in the tile: (offset, entries)
        int segcnt[WPE];                      // entries | NaN headings << 16 of segment w, counted by the wave that fetched it
        i2 totals;                            // entries of the tile, NaN headings among them
        alignas(16) float stage[1][kStageSteps][12];
        int persist_cmd[1][6];                // persistent kernels: the command this member's first wave read and the team agreed on (rollout_body)
    };

    struct Ctx {
        using Family = Team<K_>;
        Smem& sm;
        int env, slot, wave_in_env, lane, i, member, wave;
        bool owner, helper;
        int round = 0;                        // exchange rounds of this launch so far (the same in every wave of the team): slot set round & 1, tag round % 31
        int first_poll = kTeamFirstPoll;      // s_sleep units between the publish and the first poll: adapted to how long the last round's data took
        float* obs_dst = nullptr;             // rollouts with the generic observation: where this step's observation row goes -- stored by step_env
                                              // between the publish of the reduction record and the poll for the others' (under the round's latency)
        // the tile in LDS: valid for the coming step?  its size, its NaN headings, this lane's row slot in it
        bool tile_valid = false, staged = false;
        int n_cols = 0, n_nan = 0, row_slot = 0;
#ifdef EVAC_STAMP
        StampState stamp;
#endif
        __device__ __forceinline__ explicit Ctx(Smem& s) : sm(s) {
            // workgroup b = j * 8 + xcd: team (j / K) * 8 + xcd, member j % K -- the K members of a team share an XCD
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            env = (j / K) * 8 + xcd;
            member = j % K;
            const int t = threadIdx.x;
            wave = __builtin_amdgcn_readfirstlane(t / kWave);
            lane = t & (kWave - 1);
            helper = wave >= PW;
            slot = 0;
            i = helper ? (1 << 20) : member * P + t;                  // helper lanes own nobody
            wave_in_env = helper ? (1 << 10) + wave : member * PW + wave;   // ped waves: the wave index Cells<16> would have
            owner = i == 0;
        }
    };

    static __device__ __forceinline__ void sync() { __syncthreads(); }
    static __device__ __forceinline__ void invalidate(Ctx& c) { c.tile_valid = false; }   // the state changed outside step_env (autoreset)
    // a barrier of this launch timed out: the env's results are void and its state is not written back (rollout_body)
    static __device__ __forceinline__ bool aborted(Ctx& c) {
        __syncthreads();
        return c.sm.abort != 0;
    }
    static __device__ __forceinline__ void init(Ctx& c) {
        if (threadIdx.x == 0) c.sm.abort = 0;
        if (threadIdx.x < PW * kWave) *(i2*)c.sm.acc[threadIdx.x / kWave][threadIdx.x % kWave] = i2{0, 0};
        __syncthreads();
    }

    static __device__ __forceinline__ f4* xtile(const Params& p, const Ctx& c) { return (f4*)p.team_tile + ((size_t)(c.round & 1) * p.n_envs + c.env) * 1024; }
    static __device__ __forceinline__ f4* xrec(const Params& p, const Ctx& c) { return (f4*)p.team_rec + ((size_t)(c.round & 1) * p.n_envs + c.env) * (2 * WPE); }
    static __device__ __forceinline__ int tag_of(const Ctx& c) { return c.round % kTagMod; }

    // EVERY lane of a ped wave publishes an entry for the step that starts from state `q` -- its slot must carry the round's tag
    // for the readers to go on; a pedestrian that does not move is flagged (kEntryNull) and dropped by the reader, a NaN
    // heading is flagged too (the integer conversion loses it).  The row position goes to LDS, compacted per ped wave.
    static __device__ __forceinline__ void publish_entry(const Params& p, Ctx& c, const Ped& q, bool efv, bool row, float ux, float uy) {
        auto& sm = c.sm;
        const unsigned long long m_row = ballot(row);
        c.row_slot = __builtin_amdgcn_mbcnt_hi((unsigned)(m_row >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_row, 0u));
        const float X = q.x * kTileScale, Y = q.y * kTileScale;
        const float hs = p.head_scale;
        const int hx = (int)__builtin_rintf(ux * hs), hy = (int)__builtin_rintf(uy * hs);
        const bool nanh = ux != ux || uy != uy;
        const int tag = tag_of(c);
        const int fx = (hx & 0xffffff) | (tag << kEntryTagShift) | (efv ? (nanh ? kEntryNan : 0) : kEntryNull), fy = (hy & 0xffffff) | (tag << kEntryTagShift);
        // (X, heading x | tag, Y, heading y | tag): a tagged word in each 8-byte half
        store_dev(xtile(p, c) + c.wave_in_env * kWave + c.lane, f4{X, __builtin_bit_cast(float, fx), Y, __builtin_bit_cast(float, fy)});
        if (c.lane == 0) sm.rows[c.wave] = __popcll(m_row);
        if (row) sm.rowpos[c.wave][c.row_slot] = make_float2(X, Y);
    }

    struct Fetched {       // what a wave holds of its segment after a round
        f4 ev;
        bool valid;
        int rank;
    };

    // ONE ROUND of the exchange, called by all 16 waves after this member's ped waves have issued their stores into the round's slot
    // set.  There is no counter and no barrier across the members: every published 8-byte half carries the round's tag (above), so
    // every reader polls the data itself -- TILE rounds: wave w of every member entry `lane` of segment w; RECORD rounds: two helper
    // waves the 16 records (lane w: record w), which they fold with the DPP tree of Wave<16>::reduce (same tree, same rounding) and
    // leave in LDS.  One fabric trip after the last member's stores have landed everybody has the data (the counter protocol of
    // round 2 took three: store acknowledgement, counter, loads).  The workgroup barrier at the top parks the helper waves without
    // memory traffic while the member's own ped waves still compute.
    // Re-use of the slots (two sets, round & 1): a member publishes round r + 2 -- into the set of round r -- only after all its
    // waves have read round r + 1 (the barrier at the bottom), i.e. after EVERY member has published round r + 1, which each did
    // only after all its waves had read round r.  Nothing is reset: round r's data simply stops being fresh.
    // The first poll comes `first_poll` sleep units after the barrier: what the last round's data took, less one gap (polls are
    // device-scope loads that go out to the fabric, and a wave that polls early slows everybody's round trips down: round 3).
    // Bounded: 2^20 polls (about a second), then the team is lost (see team_round).
    template <bool RECORDS>
    static __device__ __forceinline__ Fetched exchange(const Params& p, Ctx& c, bool tile) {
        auto& sm = c.sm;
        __syncthreads();
        const int tag = tag_of(c);
        const f4* gt = xtile(p, c) + c.wave * kWave + c.lane;
        const int w = c.lane < WPE ? c.lane : WPE - 1;
        const bool folds = RECORDS && (c.wave == PW || c.wave == PW + 1);
        const f4* gr = xrec(p, c) + 2 * w;
        const f4 none = f4{0.0f, __builtin_bit_cast(float, kEntryNull), 0.0f, 0.0f};
        f4 ev = none, ra = f4{0.0f, 0.0f, 0.0f, 0.0f}, rb = ra;
        if ((tile || folds) && !sm.abort) {      // (uniform)
            for (int u = c.first_poll; u > 0; u -= kTeamPollGap) __builtin_amdgcn_s_sleep(kTeamPollGap);   // (s_sleep takes an immediate: in steps of the poll gap)
            int tries = 0;
            for (;;) {
                if (tile) load_dev(ev, gt);
                if (folds) { load_dev(ra, gr); load_dev(rb, gr + 1); }
                land(ev, ra, rb);
                const bool stale = (tile && !(tagged(ev.y, kEntryTagShift, tag) && tagged(ev.w, kEntryTagShift, tag))) ||
                                   (folds && !(tagged(ra.y, kCountTagShift, tag) && tagged(ra.w, kCountTagShift, tag) &&
                                               tagged(rb.y, kCountTagShift, tag) && tagged(rb.w, kCountTagShift, tag)));
                if (ballot(stale) == 0ull) break;
                if (++tries >= (1 << 20)) {
                    if (c.lane == 0) {
                        sm.abort = 1;
                        __hip_atomic_store(p.team_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host-mapped: the host reads it without a sync
                    }
                    ev = none;
                    break;
                }
                __builtin_amdgcn_s_sleep(kTeamPollGap);
            }
            // the next round's first poll: where this round's data arrived, less one gap (never before kTeamMinPoll, never later
            // than 64 units: a round that waited for a straggler must not put the next one to sleep)
            c.first_poll = min(max(c.first_poll + (tries - 1) * kTeamPollGap, kTeamMinPoll), 64);
        }
        if constexpr (RECORDS) {
            static_assert(PW + 1 < WPE, "two helper waves fold the records");
            if (c.wave == PW) {                 // the float sums: (f0, ., f1, .) (f2, ., ., .)
                f4 rf = f4{ra.x, ra.z, rb.x, 0.0f};
#define EVAC_RED_STEP(CTRL) rf.x = dpp_add<CTRL, 0xf>(rf.x); rf.y = dpp_add<CTRL, 0xf>(rf.y); rf.z = dpp_add<CTRL, 0xf>(rf.z);
                EVAC_RED_STEP(0x111)
                EVAC_RED_STEP(0x112)
                EVAC_RED_STEP(0x114)
                EVAC_RED_STEP(0x118)
#undef EVAC_RED_STEP
                if (c.lane == WPE - 1) sm.red_f = rf;
            } else if (c.wave == PW + 1) {      // the packed counts: (., c01, ., c23) (., c45, ., c67), the tags taken off
                const float w0 = ra.y, w1 = ra.w, w2 = rb.y, w3 = rb.w;
                constexpr int kCounts = (1 << kCountTagShift) - 1;
                i4 ri = i4{__builtin_bit_cast(int, w0) & kCounts, __builtin_bit_cast(int, w1) & kCounts, __builtin_bit_cast(int, w2) & kCounts, __builtin_bit_cast(int, w3) & kCounts};
#define EVAC_RED_STEP(CTRL)                                                                                       \
    ri.x = dpp_addi<CTRL, 0xf>(ri.x); ri.y = dpp_addi<CTRL, 0xf>(ri.y); ri.z = dpp_addi<CTRL, 0xf>(ri.z);         \
    ri.w = dpp_addi<CTRL, 0xf>(ri.w);
                EVAC_RED_STEP(0x111)
                EVAC_RED_STEP(0x112)
                EVAC_RED_STEP(0x114)
                EVAC_RED_STEP(0x118)
#undef EVAC_RED_STEP
                if (c.lane == WPE - 1) sm.red_i = ri;
            }
        }
        Fetched f{ev, false, 0};
        if (tile) {                             // which entries of the segment are pedestrians that move, and where they go
            const float ey = ev.y;
            const int fx = __builtin_bit_cast(int, ey);
            f.valid = (fx & kEntryNull) == 0;
            const unsigned long long mv = ballot(f.valid);
            f.rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mv >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mv, 0u));
            const int nn = __popcll(ballot((fx & kEntryNan) != 0) & mv);
            if (c.lane == 0) sm.segcnt[c.wave] = __popcll(mv) | (nn << 16);
        }
        __syncthreads();
        c.round += 1;
        return f;
    }

    // after exchange(tile = true): the 16 segments' moving pedestrians -> the LDS tile, in segment order (every wave computes
    // the prefix of the 16 counts for itself: one DPP row)
    static __device__ __forceinline__ void place_tile(Ctx& c, const Fetched& f) {
        auto& sm = c.sm;
        const int cw = sm.segcnt[c.lane < WPE ? c.lane : WPE - 1];
        const int cnt = cw & 0xffff;
        int incl = cnt, nans = cw >> 16;
        incl = dpp_addi<0x111, 0xf>(incl); nans = dpp_addi<0x111, 0xf>(nans);
        incl = dpp_addi<0x112, 0xf>(incl); nans = dpp_addi<0x112, 0xf>(nans);
        incl = dpp_addi<0x114, 0xf>(incl); nans = dpp_addi<0x114, 0xf>(nans);
        incl = dpp_addi<0x118, 0xf>(incl); nans = dpp_addi<0x118, 0xf>(nans);
        const int off = __builtin_amdgcn_readlane(incl - cnt, c.wave);
        const int n_cols = __builtin_amdgcn_readlane(incl, WPE - 1), n_nan = __builtin_amdgcn_readlane(nans, WPE - 1);
        if (f.valid) {
            const float ey = f.ev.y, ew = f.ev.w;
            const int hx = (__builtin_bit_cast(int, ey) << 8) >> 8, hy = (__builtin_bit_cast(int, ew) << 8) >> 8;   // (flag / tag byte off, sign back)
            sm.tile[off + f.rank] = f4{f.ev.x, f.ev.z, (float)hx, (float)hy};       // integer headings AS FLOATS (|h| < 2^22: exact)
        }
        if (threadIdx.x < kPad) sm.tile[n_cols + threadIdx.x] = f4{__builtin_inff(), 0.0f, 0.0f, 0.0f};
        c.n_cols = n_cols;
        c.n_nan = n_nan;
        c.tile_valid = true;
        __syncthreads();
    }

    // step_env, after the move and the classifier: the next step's entry, from a copy of the post-step state
    static __device__ __forceinline__ void stage_next(const Params& p, Ctx& c, const Ped& q, bool work) {
        Ped n = q;
        PrePair pp{};
        if (work) pp = pre_pair(p, n);
        if (work) publish_entry(p, c, n, pp.efv, pp.row, pp.ux, pp.uy);
        c.staged = true;
    }

    // The step's reduction in two halves, so that step_env can put work that does not depend on it -- the observation stores --
    // between the publish and the poll (kPipelined): the round's latency, one device-scope hand-off, is there anyway.
    template <class C>
    static __device__ __forceinline__ void reduce_publish(const Params& p, C& c, Sums& s, const unsigned long long (&pred)[8]) {
        wave_sum3(s.f0, s.f1, s.f2);
#pragma unroll
        for (int k = 0; k < 8; ++k) s.i[k] = mask_count(pred[k]);
        if (!c.helper && c.lane == 0) {        // the wave's record: the same sums and counts as a wave of Cells<16> leaves in LDS
            f4* rec = xrec(p, c);
            const int tg = tag_of(c) << kCountTagShift;       // (a count is at most 64 per wave, 1024 per env: bits 27-31 are free)
            const int c01 = s.i[0] | (s.i[1] << 16) | tg, c23 = s.i[2] | (s.i[3] << 16) | tg, c45 = s.i[4] | (s.i[5] << 16) | tg, c67 = s.i[6] | (s.i[7] << 16) | tg;
            // (sum, counts | tag) pairs: a tagged word in each 8-byte half
            store_dev(rec + 2 * c.wave_in_env, f4{s.f0, __builtin_bit_cast(float, c01), s.f1, __builtin_bit_cast(float, c23)});
            store_dev(rec + 2 * c.wave_in_env + 1, f4{s.f2, __builtin_bit_cast(float, c45), 0.0f, __builtin_bit_cast(float, c67)});
        }
    }
    template <class C>
    static __device__ __forceinline__ void reduce_collect(const Params& p, C& c, Sums& s) {
        const bool staged = c.staged;          // uniform over the team: this round also carries the next step's tile
        c.staged = false;
        EVAC_T(c, 12);   // (sub-phase: per-pedestrian work of the ped waves, record stores)
        const Fetched f = exchange<true>(p, c, staged);
        EVAC_T(c, 13);   // (sub-phase: the round -- waiting for the member's ped waves, polls, folds)
        const f4 rf = c.sm.red_f;
        const i4 ri = c.sm.red_i;
        EVAC_T(c, 14);
        if (staged) place_tile(c, f);
        EVAC_T(c, 15);   // (sub-phase: LDS tile of the next step)
        s.f0 = rf.x;
        s.f1 = rf.y;
        s.f2 = rf.z;
        const int a = ri.x, b = ri.y, d = ri.z, g = ri.w;
        s.i[0] = a & 0xffff; s.i[1] = a >> 16;
        s.i[2] = b & 0xffff; s.i[3] = b >> 16;
        s.i[4] = d & 0xffff; s.i[5] = d >> 16;
        s.i[6] = g & 0xffff; s.i[7] = g >> 16;
    }
    template <bool GUARD, class C>
    static __device__ __forceinline__ void reduce(const Params& p, C& c, Sums& s, const unsigned long long (&pred)[8]) {
        reduce_publish(p, c, s, pred);
        reduce_collect(p, c, s);
    }
    template <class C>
    static __device__ __forceinline__ void exit_publish(C&, bool, float, float) {}
    template <class C>
    static __device__ __forceinline__ void exit_fetch(C&, float, float, int, float& ex, float& ey) { ex = ey = 0.0f; }

    static __device__ __forceinline__ void neighbour_sum(const Params& p, Ctx& c, const Ped& q, bool efv, bool row,
                                                         float ux, float uy, float& sx, float& sy) {
        auto& sm = c.sm;
        if (!c.tile_valid) {      // first step of a launch, or the step after an autoreset: the exchange on its own (uniform over the team)
            if (!c.helper) publish_entry(p, c, q, efv, row, ux, uy);
            place_tile(c, exchange<false>(p, c, true));
        }
        c.tile_valid = false;     // consumed: the step's reduction brings the next one
        EVAC_T(c, 2);   // exchange (only when the tile was not delivered by the previous step)
        const int n_cols = __builtin_amdgcn_readfirstlane(c.n_cols);
        int n_rows[PW], n_rows_all = 0;
#pragma unroll
        for (int pw = 0; pw < PW; ++pw) {
            n_rows[pw] = __builtin_amdgcn_readfirstlane(sm.rows[pw]);
            n_rows_all += n_rows[pw];
        }
        if constexpr (!(EVAC_ABLATE & 1)) {
          if (n_rows_all <= kFewRows) {
            // ---- FEW rows (most of an episode: only the VISCEK pedestrians of the member need one under enslaving_degree 1):
            // the sweep is transposed.  The rows are dealt to the 16 waves round-robin and the LANES hold the columns, 64 per
            // pass, the row's position uniform: r * ceil(n_cols / 64) passes of 7 instructions in all instead of n_cols * 14 per
            // sixteenth, whatever r is (6 rows against 200 columns: 24 passes instead of 200 column visits).  The lanes' shares
            // are folded with DPP and lane 63 adds the total to the row's accumulator (integers: the order does not matter).
            // Round 4: the rows are taken in PAIRS (pair_accumulate_rows2: the two rows' positions uniform in the halves of a
            // register pair, the column per lane) and the heading sums in packed f32 -- the tile holds the integer headings as
            // floats (|h| < 2^22) and a lane adds at most kTeamExactBatch = 8 of them before the partial sum goes to its integer
            // accumulator: exact, the same bits as the integer multiply-adds gave.  6 packed instructions per pass and PAIR of
            // rows where shift + integer multiply-adds took 7 plain ones per row.
            const f4* __restrict__ tile = sm.tile;
            const f2 r2b2 = f2{kRPed2Big, kRPed2Big};
            // the member's rows numbered across its ped waves: row g of the member = (ped wave, slot); pair k = rows 2k, 2k + 1
            auto row_of = [&](int g, int& pw_out) {      // (compile-time indices into n_rows: it lives in scalar registers)
                int pw = 0;
#pragma unroll
                for (int q2 = 0; q2 + 1 < PW; ++q2) {
                    const bool next = pw == q2 && g >= n_rows[q2];
                    g -= next ? n_rows[q2] : 0;
                    pw += next ? 1 : 0;
                }
                pw_out = pw;
                return g;
            };
            if (n_rows_all <= WPE) {
                // at most one row per wave (late in an episode: 2-3 VISCEK rows per member): nothing to pair -- one row, one wave
                int before = 0;                               // rows of the ped waves before this one: the deal goes on across them
#pragma unroll
                for (int pw = 0; pw < PW; ++pw) {
                    for (int r = (c.wave - PW - before) & (WPE - 1); r < n_rows[pw]; r += WPE) {   // (the helper waves first: the ped waves come late)
                        const float2 rp = sm.rowpos[pw][r];   // (uniform address: a broadcast)
                        int ax = 0, ay = 0;
                        for (int j1 = 0; j1 < n_cols; j1 += kTeamExactBatch * kWave) {
                            float fx = 0.0f, fy = 0.0f;
                            const int jn = min(n_cols, j1 + kTeamExactBatch * kWave);
                            for (int j0 = j1; j0 < jn; j0 += kWave)
                                pair_accumulate(rp.x, rp.y, tile[min(j0 + c.lane, n_cols)], kRPed2Big, fx, fy);   // entry n_cols: padding, weight 0
                            ax += (int)fx; ay += (int)fy;
                        }
                        wave_sum2_int_lane63(ax, ay);         // (64 lanes adding to ONE LDS word would be serialised: fold in registers first)
                        if (c.lane == kWave - 1) {
                            lds_add(&sm.acc[pw][r][0], ax);
                            lds_add(&sm.acc[pw][r][1], ay);
                        }
                    }
                    before += n_rows[pw];
                }
            }
            const int n_pairs = n_rows_all <= WPE ? 0 : (n_rows_all + 1) >> 1;
            for (int k = (c.wave - PW) & (WPE - 1); k < n_pairs; k += WPE) {      // (the helper waves first: the ped waves come late)
                int pwa, pwb;
                const int ra = row_of(2 * k, pwa);
                const bool two = 2 * k + 1 < n_rows_all;
                const int rb = two ? row_of(2 * k + 1, pwb) : (pwb = pwa, ra);
                const float2 pa = sm.rowpos[pwa][ra], pb = sm.rowpos[pwb][rb];     // (uniform addresses: broadcasts)
                const f2 X2 = f2{pa.x, pb.x}, Y2 = f2{pa.y, pb.y};
                int ax = 0, ay = 0, bx = 0, by = 0;
                for (int j1 = 0; j1 < n_cols; j1 += kTeamExactBatch * kWave) {            // (one trip unless the tile has more than 512 columns)
                    f2 ax2 = f2{0.0f, 0.0f}, ay2 = f2{0.0f, 0.0f};
                    const int jn = min(n_cols, j1 + kTeamExactBatch * kWave);
                    for (int j0 = j1; j0 < jn; j0 += kWave)
                        pair_accumulate_rows2(X2, Y2, tile[min(j0 + c.lane, n_cols)], r2b2, ax2, ay2);   // entry n_cols: padding, weight 0
                    ax += (int)ax2.x; ay += (int)ay2.x; bx += (int)ax2.y; by += (int)ay2.y;
                }
                wave_sum2_int_lane63(ax, ay);                 // (64 lanes adding to ONE LDS word would be serialised: fold in registers first)
                wave_sum2_int_lane63(bx, by);
                if (c.lane == kWave - 1) {
                    lds_add(&sm.acc[pwa][ra][0], ax);
                    lds_add(&sm.acc[pwa][ra][1], ay);
                    if (two) {
                        lds_add(&sm.acc[pwb][rb][0], bx);
                        lds_add(&sm.acc[pwb][rb][1], by);
                    }
                }
            }
          } else {
            // ---- the member's rows against the tile: two ped waves (two rows per lane) per pass, 1/16 of the columns per wave ----
            const int groups = (n_cols + 3) >> 2;
            const int per = (groups + WPE - 1) / WPE;
            const int jbeg = __builtin_amdgcn_readfirstlane(c.wave * per * 4);
            const int jend = __builtin_amdgcn_readfirstlane(min((c.wave + 1) * per, groups) * 4);
            const f4* __restrict__ tile = sm.tile;
            if constexpr (PW == 1) {       // teams of 16: one ped wave per member, one row per lane
                if (sm.rows[0] != 0) {
                    const float2 rr = sm.rowpos[0][c.lane];
                    int ax = 0, ay = 0;
                    for (int j = jbeg; j < jend; j += 4) {
                        f4 t[4];
                        float fx = 0.0f, fy = 0.0f;          // (four integer headings: the float sums are exact)
#pragma unroll
                        for (int k = 0; k < 4; ++k) t[k] = tile[j + k];
#pragma unroll
                        for (int k = 0; k < 4; ++k) pair_accumulate(rr.x, rr.y, t[k], kRPed2Big, fx, fy);
                        ax += (int)fx; ay += (int)fy;
                    }
                    lds_add(&sm.acc[0][c.lane][0], ax);
                    lds_add(&sm.acc[0][c.lane][1], ay);
                }
            }
#pragma unroll
            for (int pw = 0; pw + 1 < PW; pw += 2) {
                const int na = sm.rows[pw], nb = sm.rows[pw + 1];
                if (na + nb == 0) continue;                                    // uniform
                if (na != 0 && nb != 0 && na + nb <= kWave) {
                    // the rows of BOTH ped waves fit one lane each (the middle of an episode: 33 .. 64 needed rows per member): one
                    // row per lane -- lanes [0, na) the first wave's, [na, na + nb) the second's -- in plain arithmetic, 6
                    // instructions per column + 4 per eight where the two-rows-per-lane form spends 6 PACKED ones (1.6x the pipe
                    // time each) + 10 on register pairs that are half empty
                    const bool second = c.lane >= na;
                    const float2 rr = second ? sm.rowpos[pw + 1][min(c.lane - na, kWave - 1)] : sm.rowpos[pw][c.lane];
                    int ax = 0, ay = 0;
                    int j = jbeg;
                    for (; j + 8 <= jend; j += 8) {
                        f4 t[8];
                        float fx = 0.0f, fy = 0.0f;          // (eight integer headings: the float sums are exact)
#pragma unroll
                        for (int k = 0; k < 8; ++k) t[k] = tile[j + k];
#pragma unroll
                        for (int k = 0; k < 8; ++k) pair_accumulate(rr.x, rr.y, t[k], kRPed2Big, fx, fy);
                        ax += (int)fx; ay += (int)fy;
                    }
                    if (j < jend) {
                        f4 t[4];
                        float fx = 0.0f, fy = 0.0f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) t[k] = tile[j + k];
#pragma unroll
                        for (int k = 0; k < 4; ++k) pair_accumulate(rr.x, rr.y, t[k], kRPed2Big, fx, fy);
                        ax += (int)fx; ay += (int)fy;
                    }
                    if (c.lane < na + nb) {
                        int* dst = second ? sm.acc[pw + 1][c.lane - na] : sm.acc[pw][c.lane];
                        lds_add(dst, ax);
                        lds_add(dst + 1, ay);
                    }
                    continue;
                }
                const float2 ra = sm.rowpos[pw][c.lane], rb = sm.rowpos[pw + 1][c.lane];   // slots beyond the counts hold stale rows: computed, never read
                int ax0 = 0, ay0 = 0, ax1 = 0, ay1 = 0;
                if (na != 0 && nb != 0) {
                    // (round 4: weights AND sums in packed f32 -- kTeamExactBatch = 8 integer headings sum exactly, then the partial
                    // sum goes to the integer accumulator: the same bits as before --: 6 packed instructions per column + 10 per
                    // batch where the integer multiply-adds made it 10 per column)
                    const f2 X2 = f2{ra.x, rb.x}, Y2 = f2{ra.y, rb.y}, r2b2 = f2{kRPed2Big, kRPed2Big};
                    // (software-pipelined: the four columns after the ones at hand are on their way from LDS while these are
                    // evaluated -- the asm fences keep the compiler from sinking the loads to their uses; reading up to four
                    // entries past the wave's share is harmless, the tile is padded)
                    int j = jbeg;
                    f4 ta[4], tb[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) ta[k] = tile[j + k];
                    for (; j + 8 <= jend; j += 8) {             // (the wave's share is a multiple of 4 columns)
                        f2 ax2, ay2;
#pragma unroll
                        for (int k = 0; k < 4; ++k) tb[k] = tile[j + 4 + k];
                        asm volatile("" ::: "memory");
                        pair_start_rows2(X2, Y2, ta[0], r2b2, ax2, ay2);
#pragma unroll
                        for (int k = 1; k < 4; ++k) pair_accumulate_rows2(X2, Y2, ta[k], r2b2, ax2, ay2);
#pragma unroll
                        for (int k = 0; k < 4; ++k) ta[k] = tile[j + 8 + k];
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int k = 0; k < 4; ++k) pair_accumulate_rows2(X2, Y2, tb[k], r2b2, ax2, ay2);
                        ax0 += (int)ax2.x; ax1 += (int)ax2.y; ay0 += (int)ay2.x; ay1 += (int)ay2.y;
                    }
                    if (j < jend) {
                        f2 ax2, ay2;
                        pair_start_rows2(X2, Y2, ta[0], r2b2, ax2, ay2);
#pragma unroll
                        for (int k = 1; k < 4; ++k) pair_accumulate_rows2(X2, Y2, ta[k], r2b2, ax2, ay2);
                        ax0 += (int)ax2.x; ax1 += (int)ax2.y; ay0 += (int)ay2.x; ay1 += (int)ay2.y;
                    }
                } else {
                    const float2 rr = na != 0 ? ra : rb;
                    int ax = 0, ay = 0;
                    for (int j = jbeg; j < jend; j += 4) {
                        f4 t[4];
                        float fx = 0.0f, fy = 0.0f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) t[k] = tile[j + k];
#pragma unroll
                        for (int k = 0; k < 4; ++k) pair_accumulate(rr.x, rr.y, t[k], kRPed2Big, fx, fy);
                        ax += (int)fx; ay += (int)fy;
                    }
                    ax0 = ax1 = ax;
                    ay0 = ay1 = ay;
                }
                // the 16 partial sums of a row meet in its LDS accumulator (integers: any order gives the same bits)
                if (na != 0) { lds_add(&sm.acc[pw][c.lane][0], ax0); lds_add(&sm.acc[pw][c.lane][1], ay0); }
                if (nb != 0) { lds_add(&sm.acc[pw + 1][c.lane][0], ax1); lds_add(&sm.acc[pw + 1][c.lane][1], ay1); }
            }
          }
        }
        __syncthreads();
        int tx = 0, ty = 0;
        if (!c.helper) {      // the row's sum; then the wave clears its accumulators for the next step (a wave's LDS operations execute
                              // in order, and nobody adds again before the next step's barriers)
            const i2 v = *(const i2*)sm.acc[c.wave][c.row_slot];
            tx = v.x;
            ty = v.y;
            *(i2*)sm.acc[c.wave][c.lane] = i2{0, 0};
        }
        sx = row ? (float)tx : 0.0f;
        sy = row ? (float)ty : 0.0f;
        if (c.n_nan != 0) sx = sy = __builtin_nanf("");
    }
};

}  // namespace evac
