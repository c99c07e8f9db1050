// libmgrid_hip.so - batched MultiGrid step / reset / gen_obs for gfx950 (MI355X).
//
// Execution model
// ---------------
// One 64-lane wavefront (= one workgroup) owns G sub-tiles of NE consecutive environments each, with
// NE * A <= 64 agent lanes per sub-tile, and runs four phases on them, all through LDS:
//   stage    the header blocks and the first sub-tile's grids are copied HBM -> LDS with 16-byte loads;
//   act      one lane per AGENT works out its action against the state at the start of the step; the few
//            agents whose actions could interact are marked and then take their turns in the lanes, in
//            the environment's order (the reference's sequential handle_actions loop, run only for the
//            agents it matters for); the environment hook and the step's bookkeeping are settled by the
//            environment's lanes through ballots.  CompetativeRedBlueDoor steps keep one lane per
//            ENVIRONMENT, whose agents act in the caller's order;
//   observe  one lane per AGENT walks its V x V view over the LDS grid, runs the visibility
//            scan on 7-bit row masks in registers and keeps the image in registers;
//   write    the images pass through an LDS buffer that mirrors the output tensor and leave as aligned
//            16-byte stores (optionally also one-hot encoded), the per-agent scalars as one coalesced
//            element per lane, and the header blocks as 16-byte stores.
// A further kernel draws start states on the device (one thread per layout).
//
// Device state (per handle)
// -------------------------
//   hdr   [E][HS] bytes, HS = roundup16(16 + 8 * roundup2(A) + 2 * (A + 2)):
//           int32 step_count | uint32 episode | uint8 flags | 3 pad | uint8 meta[4] | agent rows |
//           front cells
//           agent row   = x y dir terminated carry_type carry_color carry_state color
//           front cells = uint16[A + 2]: the packed grid cell in front of each agent as of the end
//                         of the last launch, then the RedBlueDoors red and blue door cells
//   grid  [E][GS] bytes, GS = roundup16(2 * W * H): one uint16 per cell, index x * H + y,
//           bits 0-3 type, 4-7 state, 8-14 color, 15 "opaque" (wall, or door that is not open).
//   pool_hdr / pool_grid: the uploaded start states in the same two formats.
// The step kernel reads hdr + grid once and writes hdr back plus only the cells that changed.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include <atomic>
#include <mutex>

#include "mgrid.h"
// Rewards are doubles worked out in the reference's order of operations (multigrid/base.py:655-659 and the hooks'
// `+=`), every operation rounded on its own as Python rounds it: no fused multiply-add anywhere in this file.
#pragma clang fp contract(off)
#include "mgrid_experiment.h"  // MG_EXP / MG_STAMP / MG_TICK: empty unless this is a diagnostic build

// Device memory of mg_gen_obs_of_state(), the one entry point that must not allocate: the packed copy of the
// caller's state lives here instead of in buffers of a handle.  Every GPU the library runs on holds its own
// copy of this array until the process ends; host code reaches it only through the per-device address that
// hipGetSymbolAddress returns (scratch_for_device()), never through the name.
constexpr size_t MG_SCRATCH_BYTES = 8u << 20;
static __device__ __align__(256) uint8_t scratch[MG_SCRATCH_BYTES];

namespace {

// ---- encodings (multigrid/core/constants.py:32-102, multigrid/core/actions.py:4-15) ----------
enum : int { T_UNSEEN, T_EMPTY, T_WALL, T_FLOOR, T_DOOR, T_KEY, T_BALL, T_BOX, T_GOAL, T_LAVA, T_AGENT };
enum : int { S_OPEN, S_CLOSED, S_LOCKED, S_PICKEDUP, S_AVAILABLE };
enum : int { C_RED = 0, C_BLUE = 2 };  // multigrid/core/constants.py:50-59
enum : int { ACT_LEFT, ACT_RIGHT, ACT_FORWARD, ACT_PICKUP, ACT_DROP, ACT_TOGGLE, ACT_DONE };
enum : int { R_X, R_Y, R_DIR, R_TERM, R_CT, R_CC, R_CS, R_COLOR };

constexpr uint32_t FLAG_LH_MASK = 0x3F;  // LockedHallway: doors already rewarded, by colour
constexpr uint32_t FLAG_OVER = 0x40;     // episode already counted as finished
constexpr uint32_t FLAG_STALE = 0x80;    // RedBlueDoors: blue door object closed, cell reads open
// CompetativeRedBlueDoorEnvV3 (multigrid/envs/competative_red_blue_door.py): what the arrays do not show
constexpr uint32_t FLAG_KEY_USED = 0x01;   // << 0 / 1: the red / blue key OBJECT is "picked up" (:434-435) while the
                                           // agent row still holds the copy made when it was picked up
constexpr uint32_t FLAG_DOOR_GONE = 0x04;  // << 0 / 1: the red / blue door's cell was cleared under an eliminated
                                           // agent (:503); the orphaned door object stays open
// meta[0] / meta[1] of such an environment: how often the red / blue ball's discount factor was squared (:454)
enum : int { RS_ELIMINATED, RS_KEY, RS_BALL, RS_INVALID };  // reward scheme columns

constexpr int H_STEP = 0, H_EPISODE = 4, H_FLAGS = 8, H_META = 12, H_ROWS = 16;
constexpr uint32_t LAYOUT_STRIDE = 7919;  // layout of episode k of env g: (g + 7919 k) mod N

constexpr int MODE_STEP = 0, MODE_OBS = 1, MODE_RESET = 2;

#ifndef MG_STAGE_LANES
#define MG_STAGE_LANES 32
#endif
// From this many agents per environment on (up to 8), environments whose agents may not overlap take the cell-table
// variant of the act phase (act_lanes<true>): most of their steps have agents to compare.
#ifndef MG_CROWD_MIN_AGENTS
#define MG_CROWD_MIN_AGENTS 6
#endif
constexpr int CROWD_MIN_AGENTS = MG_CROWD_MIN_AGENTS;
#ifndef MG_COPY_BATCH
#define MG_COPY_BATCH 5
#endif
constexpr int STAGE_LANES = MG_STAGE_LANES;  // agents whose images share the LDS staging buffer; a multiple of 16

// Byte offsets into the workgroup's LDS (at most 64 KiB, so 16 bits each: the whole map travels in nine scalar
// registers instead of eighteen).
#ifndef MG_LDS_OFFSET_TYPE
#define MG_LDS_OFFSET_TYPE uint16_t
#endif
typedef MG_LDS_OFFSET_TYPE lds_off_t;
struct LdsMap {
    lds_off_t hdr;    // [ET][HS]      header blocks of the wave's environments
    lds_off_t log;    // [ET][A]   u32 cells written by this step, in order: cell index | packed cell << 16
    lds_off_t act;    // [ET][A]   u8  actions
    lds_off_t ord;    // [ET][A]   u8  order in which agents act
    lds_off_t rewm;   // [ET][A]   u8  reward of agent k = rewm * R
    lds_off_t tout;   // [ET][A]   u8  terminated as returned to the caller
    lds_off_t oterm;  // [ET][A]   u8  terminated as seen by gen_obs
    lds_off_t flag;   // [ET][4]   u8  truncated, will-reset, number of log entries, unused
    lds_off_t R;      // [ET]      f64 _reward() of this step
    lds_off_t racc;   // [ET][A]   f64 rewards of this step               } CompetativeRedBlueDoor
    lds_off_t pxy;    // [ET][A]   u16 positions after the post-step hook } environments only
    lds_off_t log2;   // [ET][A]   u32 cells cleared by the post-step hook}
    lds_off_t tbl;    // [NE][GS+16]   act_lanes() with up to 8 agents: per grid cell, who stands on it and who faces it.
                // The single-step kernel keeps it where the grids go later (nothing is there during the act phase).
    lds_off_t src;    // [ET]      i32 layout the grid is read from (-1: the environment's own grid)
    lds_off_t grid;   // [NE][GS]      grids of the sub-tile being observed
    lds_off_t img;    // scratch of act_lanes() (8 bytes per agent lane) and, with `grid` before it, the image staging buffer
    lds_off_t stage;  // [STAGE_LANES][3VV]+32 image bytes on their way out, in output order: starts where `grid` starts
                // (the grids are dead once a sub-tile has been observed) and runs on into `img`
    lds_off_t oh;     // [192] uint4   one-hot pieces changing lanes (present only in launches that write one-hot images)
};

struct KParams {
    // configuration, packed so that it occupies few scalar registers
    uint32_t geo;   // W | H << 8 | A << 16 | NE << 24
    uint32_t bits;  // kind (2) | overlap | joint | succ_any | fail_any | autoreset (2) | num_rooms (4) | G (8) | walled
    __host__ __device__ int W() const { return geo & 0xFF; }
    __host__ __device__ int H() const { return (geo >> 8) & 0xFF; }
    __host__ __device__ int A() const { return (geo >> 16) & 0xFF; }
    __host__ __device__ int NE() const { return geo >> 24; }
    __host__ __device__ int kind() const { return bits & 3; }
    __host__ __device__ bool overlap() const { return (bits >> 2) & 1; }
    __host__ __device__ bool joint() const { return (bits >> 3) & 1; }
    __host__ __device__ bool succ_any() const { return (bits >> 4) & 1; }
    __host__ __device__ bool fail_any() const { return (bits >> 5) & 1; }
    __host__ __device__ int autoreset() const { return (bits >> 6) & 3; }
    __host__ __device__ int num_rooms() const { return (bits >> 8) & 15; }
    __host__ __device__ int G() const { return (bits >> 12) & 0xFF; }
    __host__ __device__ bool walled() const { return (bits >> 20) & 1; }  // every grid is ringed by (2,5,0) walls
    __host__ __device__ bool stream_out() const { return (bits >> 21) & 1; }     // non-temporal stores of images
    __host__ __device__ bool stream_onehot() const { return (bits >> 22) & 1; }  // ... of one-hot images
    // CompetativeRedBlueDoorEnvV3: the low kind bits then read MG_KIND_EMPTY (no other hook runs)
    __host__ __device__ bool comp() const { return (bits >> 23) & 1; }
    __host__ __device__ bool death_match() const { return (bits >> 24) & 1; }
    __host__ __device__ bool per_agent() const { return (bits >> 25) & 1; }  // dtde_step rather than ctce_step
    int max_steps;
    int E, HS, HFC, GS, num_layouts;  // HFC: offset of the front cells inside a header block
    int GP;  // distance between two environments' grids in LDS: GS + 16 (a tuning build may widen it)
    int HP;  // distance between two environments' header blocks in LDS: HS, or HS + 16 where HS / 16 is even (an odd
             // number of 16-byte slots spreads one-lane-per-environment accesses over all banks)
    uint32_t magic_a, magic_nc, magic_per, magic_ph;  // multiply-high constants for division by A, A + 2, GS / 16, HS / 16
    uint32_t env_offset;
    uint64_t seed;
    uint32_t serial;  // reset launches so far (0: the first reset starts the episode counters)
    int T;            // steps taken by one launch of the multi-step kernel (mg_step_n); inputs and outputs are [T][E][A]...
    // state
    uint8_t *hdr;
    uint8_t *grid;
    const uint8_t *pool_hdr;
    const uint8_t *pool_grid;
    int64_t *stats;  // [64][MG_NUM_STATS] shards
    const double *schemes;  // [A][4] reward schemes (CompetativeRedBlueDoor only)
    // inputs
    const uint8_t *actions;
    const uint8_t *order;
    const int32_t *layout_idx;
    const uint8_t *mask;
    // outputs
    uint8_t *image;
    uint8_t *onehot;
    uint8_t *direction;
    float *reward;
    uint8_t *terminated;
    uint8_t *truncated;
    uint8_t *rows_out;   // [E][A][8] agent rows as the launch leaves them in the returned observation's state
    double *reward64;    // [E][A] the rewards as the doubles the reference computes
    uint8_t *info;       // [E][A] CompetativeRedBlueDoor: the step's info dict, a byte per agent
    LdsMap o;
    MG_EXP(
    int dbg;
    unsigned long long *dbg_buf;
    )
};

// A kernel argument read where it is needed instead of at kernel entry: the compiler loads a by-value argument
// struct as a whole up front and then keeps (or spills) every field for the length of the kernel.  `off` is the
// field's offset in KParams, the kernel's only argument.
template <typename T>
__device__ __forceinline__ T late_arg(size_t off) {
    typedef const __attribute__((address_space(4))) char *karg_ptr;
    karg_ptr base = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));  // opaque: the load below cannot move up past this point
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(base + off);
}

// this workgroup's shard of the episode statistics (touched rarely: the pointer is fetched where it is needed)
__device__ __forceinline__ int64_t *stats_shard() {
    return late_arg<int64_t *>(offsetof(KParams, stats)) + (blockIdx.x & 63) * MG_NUM_STATS;
}

constexpr uint32_t FC_NONE = 0x7FFF;  // "no cell": the position lies outside the grid

__device__ __forceinline__ uint32_t pack_cell(uint32_t t, uint32_t c, uint32_t s) {
    // "opaque" restates see_behind() == False, multigrid/utils/obs.py:44-61
    const uint32_t opaque = (t == T_WALL) | ((t == T_DOOR) & (s != S_OPEN));
    return t | (s << 4) | (c << 8) | (opaque << 15);
}
constexpr uint32_t CELL_EMPTY = T_EMPTY;                        // (1, 0, 0)
constexpr uint32_t CELL_WALL = T_WALL | (5u << 8) | (1u << 15);  // WALL_ENCODING (2, 5, 0), obs.py:13

// DIR_TO_VEC, multigrid/core/constants.py:20-29: right (1,0), down (0,1), left (-1,0), up (0,-1),
// read out of two-bit fields holding component + 1
__device__ __forceinline__ int dir_dx(int d) { return static_cast<int>((0x46u >> (2 * d)) & 3u) - 1; }
__device__ __forceinline__ int dir_dy(int d) { return static_cast<int>((0x19u >> (2 * d)) & 3u) - 1; }

// Position one step ahead of an agent row word (x | y << 8 | dir << 16 | ...), packed x | y << 8;
// a coordinate that leaves 0..255 wraps to a value no grid contains (agent.py:153-160).
__device__ __forceinline__ uint32_t front_xy(uint32_t r) {
    const uint32_t delta = static_cast<uint32_t>(0xFF0000FF01000001ull >> (16 * ((r >> 16) & 3)));
    const uint32_t x = (r + delta) & 0xFF, y = ((r & 0xFF00) + (delta & 0xFF00)) & 0xFF00;
    return x | y;
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// A workgroup here is exactly one wavefront.  LDS instructions of one wavefront execute in issue
// order, so a lane sees what another lane wrote earlier in program order; all that is needed
// between phases is to stop the compiler from moving LDS accesses across the phase boundary.
// Unlike __syncthreads() this does not wait for outstanding global loads and stores.
// The scheduling barrier also pins global loads that were issued early on purpose (the grid
// prefetch) on their side of the phase boundary.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);
}

// Stores of images.  Nothing on the GPU reads them again soon, so beyond a
// size (chosen per launch on the host, see set_out()) they are marked non-temporal and leave the caches to
// the state arrays, which every launch re-reads.  Measured: 8-9 % on 65536 environments of 2 or 4 agents,
// 9 % on 1.2 GB of one-hot images; small outputs are better left to the caches (one-hot images up to
// 300 MB lose 7-14 % when streamed, a 1 MB image 6 %).
// (Written as inline assembly: the optimiser folds "if (stream) non-temporal store; else store;" into one
// plain store.)
typedef uint32_t mg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_out16(uint4 *dst, const uint4 &v, bool stream) {
    if (stream) {
        const mg_u32x4 t = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(t) : "memory");
    } else {
        *dst = v;
    }
}
// n / d for small n, with magic = floor(2^32 / d) + 1 (0 stands for d == 1)
__device__ __forceinline__ int udiv(int n, uint32_t magic) {
    return magic ? static_cast<int>(__umulhi(static_cast<uint32_t>(n), magic)) : n;
}

// layout of episode `episode` of environment `env`: (env_id_offset + env + 7919 * episode) mod N
__device__ __forceinline__ int pick_layout(const KParams &p, int env, uint32_t episode) {
    // (restarts are rare: both numbers are fetched here rather than kept for the whole step)
    const uint32_t n = static_cast<uint32_t>(late_arg<int>(offsetof(KParams, num_layouts)));
    const uint32_t g = late_arg<uint32_t>(offsetof(KParams, env_offset)) + static_cast<uint32_t>(env);  // both below 2^31
    if (n <= 0xFFFFu)  // the same residue from 32-bit pieces: a 64-bit remainder is a long software loop
        return static_cast<int>((g % n + (episode % n) * (LAYOUT_STRIDE % n)) % n);
    return static_cast<int>((static_cast<uint64_t>(g) + static_cast<uint64_t>(LAYOUT_STRIDE) * episode) % static_cast<uint64_t>(n));
}

// On-device agent order (used when the caller passes no order): a Fisher-Yates shuffle of
// 0..A-1, for k = A-1 down to 1 swapping entries k and j = (field * (k + 1)) >> 16.  The 16-bit
// fields are taken low half first from the 32-bit words fmix32(key + (q + 1) * 0x9E3779B9),
// q = 0, 1, ...; the key mixes (seed, global environment index, episode, step) through the
// murmur3 finalizer fmix32.  It stands in for np_random.random(A).argsort() of
// multigrid/base.py:457-461 and does not depend on how launches are issued (eager, graph
// replay, sharding).
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    return h ^ (h >> 16);
}
__device__ __forceinline__ uint32_t order_key(const KParams &p, int env, uint32_t episode, int step) {
    uint32_t h = fmix32(static_cast<uint32_t>(p.seed) ^ (p.env_offset + static_cast<uint32_t>(env)));
    h = fmix32(h ^ static_cast<uint32_t>(p.seed >> 32) ^ episode);
    return fmix32(h ^ static_cast<uint32_t>(step));
}
__device__ __forceinline__ uint32_t order_field(uint32_t key, int f, uint32_t &word) {
    if ((f & 1) == 0) word = fmix32(key + static_cast<uint32_t>((f >> 1) + 1) * 0x9E3779B9u);
    return (word >> (16 * (f & 1))) & 0xFFFF;
}

// _reward(), multigrid/base.py:655-659, in the reference's own order of operations (the quotient first), so
// that the double is the one Python computes
__device__ __forceinline__ double step_reward(const KParams &p, int step) {
    return 1.0 - 0.9 * (static_cast<double>(step) / static_cast<double>(p.max_steps));
}

// rewards[k] after `m` additions of R to 0 (multigrid/envs/locked_hallway.py:207-211 repeats `+=`)
__device__ __forceinline__ double reward_times(double R, int m) {
    double v = 0.0;
    for (int j = 0; j < m; ++j) v += R;
    return v;
}


// ---- act: one lane per environment -------------------------------------------------------------

// index of the cell in front of an agent row (x y dir term), or FC_NONE outside the grid
__device__ __forceinline__ uint32_t front_index(const KParams &p, uint32_t r) {
    const uint32_t f = front_xy(r);
    const uint32_t fx = f & 0xFF, fy = f >> 8;
    return (fx < static_cast<uint32_t>(p.W()) && fy < static_cast<uint32_t>(p.H())) ? fx * p.H() + fy : FC_NONE;
}

// ---- one agent's action ---------------------------------------------------------------------------
//
// The body of the loop multigrid/base.py:464-528 for one agent, split where it asks whether another
// agent stands on the cell in front (base.py:485,512): eval_action() decides everything that depends
// only on the agent and that cell, finish_action() the rest.  Lane-varying conditions are kept as
// bits of one register rather than as booleans.
enum : uint32_t {
    EV_ALIVE = 1u, EV_ONCELL = 2u, EV_FWD = 4u, EV_DROP = 8u, EV_PK = 16u, EV_UNLOCK = 32u, EV_FLIP = 64u,
    EV_BOX = 128u, EV_BLUE = 256u, EV_GOAL = 512u, EV_TERM = 1024u, EV_WRITE = 2048u,
    EV_MAY_WRITE = EV_DROP | EV_PK | EV_UNLOCK | EV_FLIP | EV_BOX,
    // what the action could do if the cell in front held something else (act_lanes(), second look)
    EV_TRY_FWD = 4096u, EV_TRY_DROP = 8192u, EV_UNSTABLE = 16384u,
};

struct ActEval {
    uint2 r;        // the agent's row after the action
    uint32_t fxy;   // position in front, x | y << 8
    uint32_t ci;    // its grid index (meaningful with EV_ONCELL)
    uint32_t w;     // value of that cell when the agent acts
    uint32_t nv;    // value the cell takes (with EV_WRITE)
    uint32_t bits;  // EV_*
};

template <bool COMP = false>
__device__ __forceinline__ ActEval eval_action(const KParams &p, uint2 r0, uint32_t w, int act, uint32_t flags,
                                               uint32_t blue_xy) {
    ActEval e;
    const bool alive = (r0.x >> 24) == 0;  // base.py:467
    const int dir = (r0.x >> 16) & 3;
    // left / right, base.py:471-476
    const int turn = act == ACT_LEFT ? 3 : (act == ACT_RIGHT ? 1 : 0);
    e.r = r0;
    e.r.x = (r0.x & 0xFF00FFFFu) | (static_cast<uint32_t>((dir + turn) & 3) << 16);
    // the cell in front, agent.py:153-160
    e.fxy = front_xy(r0.x);
    const uint32_t fx = e.fxy & 0xFF, fy = e.fxy >> 8;
    const bool on_cell = alive && act >= ACT_FORWARD && act <= ACT_TOGGLE && fx < static_cast<uint32_t>(p.W()) &&
                         fy < static_cast<uint32_t>(p.H());
    e.ci = fx * p.H() + fy;
    e.w = w;
    const int t = w & 15, s = (w >> 4) & 15, c = (w >> 8) & 127;
    const int ct = r0.y & 0xFF, cc = (r0.y >> 8) & 0xFF;
    int cs = (r0.y >> 16) & 0xFF;
    // CompetativeRedBlueDoor: a dropped object takes the OBJECT's state (grid.py:91-95), and a key that has paid its
    // pickup reward is "picked up" whatever the copy in the agent row says (competative_red_blue_door.py:434-435)
    if (COMP && ct == T_KEY && (cc == C_RED || cc == C_BLUE) && ((flags >> (cc >> 1)) & FLAG_KEY_USED)) cs = S_PICKEDUP;
    const bool blue = p.kind() == MG_KIND_REDBLUEDOORS && e.fxy == blue_xy;
    // state of the world OBJECT: equals the cell except for the blue door re-closed by
    // redbluedoors.py:185 without a grid.update
    const int so = (blue && (flags & FLAG_STALE)) ? S_CLOSED : s;
    const bool can_overlap = t == T_EMPTY || t == T_GOAL || t == T_FLOOR || t == T_LAVA || (t == T_DOOR && so == S_OPEN);
    const bool fwd = on_cell && act == ACT_FORWARD && can_overlap;                  // base.py:479-483
    const bool drop = on_cell && act == ACT_DROP && ct != T_EMPTY && t == T_EMPTY;  // base.py:507-511
    const int my_color = r0.y >> 24;
    // competative_red_blue_door.py:573-577: an agent picks up its own colour's key and other colours' balls only
    const bool pickable = COMP ? ((t == T_KEY && c == my_color) || (t == T_BALL && c != my_color))
                                   : (t == T_KEY || t == T_BALL || t == T_BOX);
    const bool pk = on_cell && act == ACT_PICKUP && pickable && ct == T_EMPTY;  // :497-504
    const bool door = on_cell && act == ACT_TOGGLE && t == T_DOOR;  // Door.toggle, world_object.py:456-472
    const bool unlock = door && so == S_LOCKED && ct == T_KEY && cc == c;
    const bool flip = door && so != S_LOCKED;
    const bool box = on_cell && act == ACT_TOGGLE && t == T_BOX;  // Box.toggle, no contents, world_object.py:653-659
    e.nv = (pk || box) ? CELL_EMPTY
           : drop      ? pack_cell(ct, cc, cs)
                       : pack_cell(T_DOOR, c, (unlock || so != S_OPEN) ? S_OPEN : S_CLOSED);
    e.bits = (alive ? EV_ALIVE : 0u) | (on_cell ? EV_ONCELL : 0u) | (fwd ? EV_FWD : 0u) | (drop ? EV_DROP : 0u) |
             (pk ? EV_PK : 0u) | (unlock ? EV_UNLOCK : 0u) | (flip ? EV_FLIP : 0u) | (box ? EV_BOX : 0u) |
             (blue ? EV_BLUE : 0u) | ((on_cell && act == ACT_FORWARD) ? EV_TRY_FWD : 0u) |
             ((on_cell && act == ACT_DROP) ? EV_TRY_DROP : 0u);
    return e;
}

// does the action have to know whether an agent stands in front?  (forward without overlap, drop)
__device__ __forceinline__ bool needs_occupancy(const KParams &p, const ActEval &e) {
    return ((e.bits & EV_FWD) && !p.overlap()) || (e.bits & EV_DROP);
}

// `occupied`: some agent, terminated ones included, stands on the cell in front (base.py:485,512)
__device__ __forceinline__ void finish_action(const KParams &p, ActEval &e, uint2 r0, bool occupied) {
    const int t = e.w & 15;
    const bool mv = (e.bits & EV_FWD) && !(occupied && !p.overlap());
    const bool dr = (e.bits & EV_DROP) && !occupied;
    if (mv) e.r.x = (e.r.x & 0xFFFF0000u) | e.fxy;  // base.py:489
    // on_success / on_failure with the empty terminations dict of base.py:492,494: mode "any"
    // terminates nobody here, mode "all" terminates this agent
    const bool goal = mv && t == T_GOAL, lava = mv && t == T_LAVA;
    if ((goal && !p.succ_any()) || (lava && !p.fail_any())) {
        e.r.x |= 1u << 24;
        e.bits |= EV_TERM;
    }
    if (goal) e.bits |= EV_GOAL;
    if (e.bits & EV_PK)
        e.r.y = (r0.y & 0xFF000000u) | static_cast<uint32_t>(t | (((e.w >> 8) & 127) << 8) | (((e.w >> 4) & 15) << 16));
    if (dr) e.r.y = (r0.y & 0xFF000000u) | T_EMPTY;
    if (dr || (e.bits & (EV_PK | EV_UNLOCK | EV_FLIP | EV_BOX))) e.bits |= EV_WRITE;
}

// ---- act, one lane per environment (CompetativeRedBlueDoor steps, up to 8 agents): the reference's sequential
// handle_actions loop as it stands, per-agent bytes packed into 64-bit registers -------------------------------

__device__ __forceinline__ uint32_t byte_of(uint64_t w, int k) { return static_cast<uint32_t>(w >> (8 * k)) & 0xFF; }

// A (<= 8) consecutive bytes, as one aligned load when A is a power of two
__device__ __forceinline__ uint64_t load_bytes(const uint8_t *ptr, int A) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    if (A == 8 && (a & 7) == 0) return *reinterpret_cast<const uint64_t *>(ptr);
    if (A == 4 && (a & 3) == 0) return *reinterpret_cast<const uint32_t *>(ptr);
    if (A == 2 && (a & 1) == 0) return *reinterpret_cast<const uint16_t *>(ptr);
    uint64_t w = 0;
    for (int k = 0; k < A; ++k) w |= static_cast<uint64_t>(ptr[k]) << (8 * k);
    return w;
}

__device__ __forceinline__ void store_bytes(uint8_t *ptr, uint64_t w, int A) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    if (A == 8 && (a & 7) == 0) { *reinterpret_cast<uint64_t *>(ptr) = w; return; }
    if (A == 4 && (a & 3) == 0) { *reinterpret_cast<uint32_t *>(ptr) = static_cast<uint32_t>(w); return; }
    if (A == 2 && (a & 1) == 0) { *reinterpret_cast<uint16_t *>(ptr) = static_cast<uint16_t>(w); return; }
    for (int k = 0; k < A; ++k) ptr[k] = static_cast<uint8_t>(w >> (8 * k));
}

// MultiGridEnv.step up to (not including) gen_obs, then the environment's hook, for one environment (`slot` of the
// tile): multigrid/base.py:395-396,440-533 and competative_red_blue_door.py:319-523.  Same contract as act_lanes();
// `actw` / `ordw` hold the A action and order bytes (agents act in index order when the caller gives no order).
template <bool COMP>
__device__ void act_env_small(const KParams &p, uint8_t *lds, int slot, int env, uint64_t actw, uint64_t ordw) {
    const int A = p.A(), H = p.H();
    uint8_t *h = lds + p.o.hdr + slot * p.HP;
    uint8_t *rows = h + H_ROWS;
    const uint16_t *fcv = reinterpret_cast<const uint16_t *>(h + p.HFC);
    uint32_t *log = reinterpret_cast<uint32_t *>(lds + p.o.log) + slot * A;
    const uint64_t ones = 0x0101010101010101ull >> (8 * (8 - A));  // byte k = 1 for k < A

    MG_EXP(
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0;
    MG_TICK(q0);
    )
    const uint4 h0 = *reinterpret_cast<const uint4 *>(h);
    const int step = static_cast<int>(h0.x) + 1;  // base.py:395
    *reinterpret_cast<int *>(h + H_STEP) = step;
    uint32_t flags = h0.z & 0xFF;
    const uint32_t was_over = flags & FLAG_OVER;
    const uint32_t red_xy = h0.w & 0xFFFF, blue_xy = h0.w >> 16;
    const bool rbd = p.kind() == MG_KIND_REDBLUEDOORS;
    // _reward(): CompetativeRedBlueDoor adds to it as it goes; the others only work it out (a double division) where
    // somebody is paid, everybody else's reward being 0 * R
    double R = COMP ? step_reward(p, step) : 0.0;

    int nlog = 0;           // cells written so far in this step
    uint64_t termw, rewmw;  // byte k: agent k terminated / reward multiplier
    {
        if (p.order == nullptr) {  // see order_key()
            const uint32_t key = order_key(p, env, h0.y, step);
            ordw = 0x0706050403020100ull;
            uint32_t word = 0;
            // (CompetativeRedBlueDoor agents act in the order of the caller's actions dict, never a drawn one:
            // without an order they act in index order)
#pragma unroll 1
            for (int k = COMP ? 0 : A - 1, f = 0; k > 0; --k, ++f) {
                const int j = static_cast<int>((order_field(key, f, word) * static_cast<uint32_t>(k + 1)) >> 16);
                const uint64_t d = byte_of(ordw, k) ^ byte_of(ordw, j);
                ordw ^= (d << (8 * k)) ^ (d << (8 * j));
            }
        }
        MG_EXP(MG_TICK(q1);)
        termw = 0;
        rewmw = 0ull;  // base.py:454
#pragma unroll 1
        for (int k = 0; k < A; ++k) termw |= static_cast<uint64_t>(rows[8 * k + R_TERM]) << (8 * k);
#pragma unroll 1
        for (int n = 0; n < A; ++n) {  // base.py:464
            const int i = min(static_cast<int>(byte_of(ordw, n)), A - 1);
            uint8_t *a = rows + 8 * i;
            const uint2 r0 = *reinterpret_cast<const uint2 *>(a);  // x y dir term | ct cc cs color
            uint32_t w = fcv[i];
            const uint32_t ci0 = front_index(p, r0.x);
#pragma unroll 1
            for (int j = 0; j < nlog; ++j) {  // cells already written during this step
                const uint32_t e = log[j];
                if ((e & 0xFFFF) == ci0) w = e >> 16;
            }
            ActEval ev = eval_action<COMP>(p, r0, w, byte_of(actw, i), flags, blue_xy);
            bool occupied = false;  // base.py:485,512: every agent counts, terminated ones included
            if (needs_occupancy(p, ev))
#pragma unroll 1
                for (int k = 0; k < A; ++k) occupied |= *reinterpret_cast<const uint16_t *>(rows + 8 * k) == ev.fxy;
            finish_action(p, ev, r0, occupied);
            if (ev.bits & EV_TERM) termw |= 1ull << (8 * i);
            if (ev.bits & EV_GOAL) rewmw = p.joint() ? ones : ((rewmw & ~(0xFFull << (8 * i))) | (1ull << (8 * i)));  // base.py:558-562
            if ((ev.bits & EV_FLIP) && (ev.bits & EV_BLUE)) flags &= ~FLAG_STALE;  // grid.update re-syncs cell and object
            if ((ev.bits & EV_ALIVE) && (ev.r.x != r0.x || ev.r.y != r0.y)) *reinterpret_cast<uint2 *>(a) = ev.r;
            if (ev.bits & EV_WRITE) log[nlog++] = ev.ci | (ev.nv << 16);
        }
    }
    if (p.joint() && rewmw) rewmw = ones;  // base.py:558-559: a reward paid to one is paid to all
    MG_EXP(MG_TICK(q2);)

    // what gen_obs sees (base.py:399) and what step returns (base.py:400-402)
    store_bytes(lds + p.o.oterm + slot * A, termw, A);
    uint64_t toutw = termw;
    const bool trunc = step >= p.max_steps;

    // subclass hooks run on the returned dicts, agents in index order.  An agent whose action was
    // "toggle" has neither moved nor turned, so its header entry still names the cell in front of it.
    bool term_all = false;  // a hook terminated every agent
    uint64_t term_some = 0; // a hook terminated these agents (mode "all")
    uint64_t info_bits = 0; // CompetativeRedBlueDoor: the info dict's bytes, as they are kept in the `tout` bytes
    double paid_sum = 0.0;  // CompetativeRedBlueDoor: sum of this step's rewards
    int comp_cleared = 0;   // ... and the number of cells its hook cleared (`log2`)
    if constexpr (COMP) {
        // CompetativeRedBlueDoorEnvV3.dtde_step / ctce_step with the default handlers
        // (multigrid/envs/competative_red_blue_door.py:319-423): for every agent, in the order of the actions dict,
        // _handle_steps (:475-523) then _default_extra_action_handle_steps (:425-472).  gen_obs has run by now in the
        // reference, so nothing here may show in this step's observation: positions changed by an elimination go to
        // `pxy`, cleared cells to `log2`; the observation pass applies both after the views are taken.
        double *racc = reinterpret_cast<double *>(lds + p.o.racc) + slot * A;
        uint16_t *pxy = reinterpret_cast<uint16_t *>(lds + p.o.pxy) + slot * A;
        uint32_t *log2 = reinterpret_cast<uint32_t *>(lds + p.o.log2) + slot * A;
        const int W = p.W();
        const uint32_t door_xy[2] = {static_cast<uint32_t>(W / 4) | (static_cast<uint32_t>(H - 2) << 8),
                                     static_cast<uint32_t>(W / 4 + W / 2 - 1) | (1u << 8)};  // _gen_grid, :201-211
        uint64_t xw = 0, yw = 0, moved = 0;  // positions as the hook leaves them; agents it has moved
#pragma unroll 1
        for (int k = 0; k < A; ++k) {
            xw |= static_cast<uint64_t>(rows[8 * k + R_X]) << (8 * k);
            yw |= static_cast<uint64_t>(rows[8 * k + R_Y]) << (8 * k);
            racc[k] = byte_of(rewmw, k) ? R : 0.0;  // a goal reached in handle_actions (base.py:491-492)
        }
        uint64_t iterm = termw;  // agent.terminated
        // the step's info dict (:320-328, :380-388), a byte per agent: bit 0 door_open_done, bit 1 got_eliminated_done,
        // bit 2 eliminated_opponents_done, bits 4-5 eliminated_opponent_num (at most one per step: the elimination that
        // leaves no opponent); bits 6 / 7: the agent's pickup was a rewarded key / ball pickup, after which the reference
        // rewrites the team's mission texts (:438-443, :456-468)
        uint64_t infow = 0;
        if (p.per_agent() && termw != 0 && termw != ones) toutw = 0;  // :375-378: some but not all -> nobody
        uint32_t ball_n = h0.w & 0xFFFF;  // meta[0], meta[1]
        int n2 = 0;
        auto succeed = [&](int k) {  // on_success(agent k, reward, terminated), base.py:535-562
            if (p.succ_any()) { iterm = ones; toutw = ones; }
            else { iterm |= 1ull << (8 * k); toutw |= 1ull << (8 * k); }
            if (p.joint()) { for (int j = 0; j < A; ++j) racc[j] = R; }
            else racc[k] = R;
        };
#pragma unroll 1
        for (int n = 0; n < A; ++n) {
            const int i = min(static_cast<int>(byte_of(ordw, n)), A - 1);
            const int act = byte_of(actw, i);
            if (act != ACT_TOGGLE && act != ACT_PICKUP) continue;
            const uint32_t my_color = rows[8 * i + R_COLOR];
            const uint32_t me = byte_of(xw, i) | (byte_of(yw, i) << 8) | (static_cast<uint32_t>(rows[8 * i + R_DIR]) << 16);
            const uint32_t fxy = front_xy(me);
            if (act == ACT_TOGGLE) {
                // the cell in front as it is now; an agent the hook has just moved faces a cell of a side room
                uint32_t w = fcv[i];
                const uint32_t ci = front_index(p, me);
                const bool was_moved = byte_of(moved, i) != 0;
#pragma unroll 1
                for (int j = 0; j < nlog; ++j) {
                    const uint32_t e = log[j];
                    if ((e & 0xFFFF) == ci) w = e >> 16;
                }
#pragma unroll 1
                for (int j = 0; j < n2; ++j)
                    if ((log2[j] & 0xFFFF) == ci) w = CELL_EMPTY;
                if (was_moved) w = CELL_EMPTY;
                int victim = -1;  // :479-481: the last opponent standing in front
#pragma unroll 1
                for (int k = 0; k < A; ++k)
                    if ((byte_of(xw, k) | (byte_of(yw, k) << 8)) == fxy && rows[8 * k + R_COLOR] != my_color) victim = k;
                // the two door OBJECTS: open once orphaned, else as their cells read
                bool door_open = false;
#pragma unroll 1
                for (int d = 0; d < 2; ++d) {
                    uint32_t wd = fcv[A + d];
                    const uint32_t cd = (door_xy[d] & 0xFF) * H + (door_xy[d] >> 8);
#pragma unroll 1
                    for (int j = 0; j < nlog; ++j) {
                        const uint32_t e = log[j];
                        if ((e & 0xFFFF) == cd) wd = e >> 16;
                    }
                    door_open |= (flags & (FLAG_DOOR_GONE << d)) || ((wd & 15) == T_DOOR && ((wd >> 4) & 15) == S_OPEN);
                }
                const bool at_door = (fxy == door_xy[0] && !(flags & FLAG_DOOR_GONE)) ||
                                     (fxy == door_xy[1] && !(flags & (FLAG_DOOR_GONE << 1)));
                if (victim < 0) {
                    if (at_door && (w & 15) == T_DOOR && door_open && ((w >> 8) & 127) == my_color) {  // :484-492
#pragma unroll 1
                        for (int k = 0; k < A; ++k)
                            if (rows[8 * k + R_COLOR] == my_color && !byte_of(iterm, k)) {
                                succeed(k);
                                infow |= 1ull << (8 * k);  // :490-492
                            }
                    }
                } else if (p.death_match()) {  // :498-523
                    if (!p.fail_any()) iterm |= 1ull << (8 * victim);  // on_failure without a terminations dict
                    infow |= 2ull << (8 * victim);  // :502
                    if (at_door && (w & 15) == T_DOOR) flags |= fxy == door_xy[0] ? FLAG_DOOR_GONE : (FLAG_DOOR_GONE << 1);
                    log2[n2++] = ci | (CELL_EMPTY << 16);  // grid.set(*fwd_obj.pos, None), :503
                    const bool blue = rows[8 * victim + R_COLOR] == C_BLUE;  // :504-506
                    xw = (xw & ~(0xFFull << (8 * victim))) | (static_cast<uint64_t>(blue ? 13 : 2) << (8 * victim));
                    yw = (yw & ~(0xFFull << (8 * victim))) | (2ull << (8 * victim));
                    moved |= 1ull << (8 * victim);
                    racc[i] += p.schemes[4 * i + RS_ELIMINATED];
                    racc[victim] -= 1.0;
                    bool all_gone = true;  // :511-517
#pragma unroll 1
                    for (int k = 0; k < A; ++k) all_gone &= rows[8 * k + R_COLOR] == my_color || byte_of(iterm, k) != 0;
                    if (all_gone) {
#pragma unroll 1
                        for (int k = 0; k < A; ++k)
                            if (rows[8 * k + R_COLOR] == my_color && !byte_of(iterm, k)) {
                                succeed(k);
                                infow = (infow | (4ull << (8 * k))) + (16ull << (8 * k));  // :517-522
                            }
                    }
                }
            } else {  // pickup: _default_extra_action_handle_steps, :425-472
                const uint32_t carry = *reinterpret_cast<const uint32_t *>(rows + 8 * i + 4);  // ct cc cs color
                const uint32_t ct = carry & 0xFF, cc = (carry >> 8) & 0xFF;
                uint32_t cs = (carry >> 16) & 0xFF;
                const bool rb = cc == C_RED || cc == C_BLUE;
                if (ct == T_KEY && rb && ((flags >> (cc >> 1)) & FLAG_KEY_USED)) cs = S_PICKEDUP;
                // Ball(color="blue") starts next to the red door, Ball(color="red") next to the blue door (:230-232)
                const uint32_t init_xy = cc == C_RED ? door_xy[1] - 1 : door_xy[0] + 1;
                if (ct == T_KEY && cs == S_AVAILABLE && cc == my_color) {  // :428-436
                    if (rb) flags |= FLAG_KEY_USED << (cc >> 1);
                    racc[i] += p.schemes[4 * i + RS_KEY];
                    infow |= 0x40ull << (8 * i);  // the key pickup that rewrites the team's missions, :438-443
                } else if (ct == T_BALL && rb && cc != my_color && fxy == init_xy) {  // :445-454
                    const int sh = 8 * (cc >> 1), cnt = (ball_n >> sh) & 0xFF;
                    double d = 0.9;  // Ball.__new__ default, world_object.py:583
#pragma unroll 1
                    for (int j = 0; j < cnt; ++j) d *= d;
                    racc[i] += p.schemes[4 * i + RS_BALL] * d;
                    if (cnt < 255) ball_n += 1u << sh;
                    infow |= 0x80ull << (8 * i);  // the ball pickup that rewrites the team's missions, :456-468
                } else {
                    racc[i] -= p.schemes[4 * i + RS_INVALID];  // :470-472
                }
            }
        }
#pragma unroll 1
        for (int k = 0; k < A; ++k) {
            pxy[k] = static_cast<uint16_t>(byte_of(xw, k) | (byte_of(yw, k) << 8));
            if (byte_of(iterm, k)) rows[8 * k + R_TERM] = 1;
            paid_sum += racc[k];
        }
        h[H_META] = static_cast<uint8_t>(ball_n);
        h[H_META + 1] = static_cast<uint8_t>(ball_n >> 8);
        comp_cleared = n2;
        // the info bytes travel in the upper bits of the `tout` bytes: terminated | flags << 1 | count << 4
        info_bits = ((infow & 0x0707070707070707ull) << 1) | (infow & 0xF0F0F0F0F0F0F0F0ull);
    } else if (rbd) {  // redbluedoors.py:176-185
#pragma unroll 1
        for (int i = 0; i < A; ++i) {
            if (byte_of(actw, i) != ACT_TOGGLE) continue;
            if (front_xy(*reinterpret_cast<const uint32_t *>(rows + 8 * i)) != blue_xy) continue;
            uint32_t wb = fcv[A + 1], wr = fcv[A];
            const uint32_t cb = (blue_xy & 0xFF) * H + (blue_xy >> 8), cr = (red_xy & 0xFF) * H + (red_xy >> 8);
#pragma unroll 1
            for (int j = 0; j < nlog; ++j) {
                const uint32_t e = log[j];
                if ((e & 0xFFFF) == cb) wb = e >> 16;
                if ((e & 0xFFFF) == cr) wr = e >> 16;
            }
            if ((wb & 15) != T_DOOR) continue;
            const int so = (flags & FLAG_STALE) ? S_CLOSED : static_cast<int>((wb >> 4) & 15);
            if (so != S_OPEN) continue;
            if (((wr >> 4) & 15) == S_OPEN) {  // on_success, base.py:535-562
                if (p.succ_any()) term_all = true; else term_some |= 1ull << (8 * i);
                rewmw = p.joint() ? ones : ((rewmw & ~(0xFFull << (8 * i))) | (1ull << (8 * i)));
            } else {  // on_failure, base.py:564-585
                if (p.fail_any()) term_all = true; else term_some |= 1ull << (8 * i);
                flags |= FLAG_STALE;
            }
        }
    } else if (p.kind() == MG_KIND_BLOCKEDUNLOCKPICKUP) {  // blockedunlockpickup.py:165-167
#pragma unroll 1
        for (int i = 0; i < A; ++i) {
            if (rows[8 * i + R_CT] != T_BOX) continue;
            term_all = true;  // this environment always terminates "any" (blockedunlockpickup.py:128)
            rewmw = p.joint() ? ones : ((rewmw & ~(0xFFull << (8 * i))) | (1ull << (8 * i)));
        }
        if (!p.succ_any() && term_all) {  // not reachable through the registry; keeps base.py:553-556 exact
            term_all = false;
#pragma unroll 1
            for (int i = 0; i < A; ++i)
                if (rows[8 * i + R_CT] == T_BOX) term_some |= 1ull << (8 * i);
        }
    } else if (p.kind() == MG_KIND_LOCKEDHALLWAY) {  // locked_hallway.py:200-215
#pragma unroll 1
        for (int i = 0; i < A; ++i) {
            if (byte_of(actw, i) != ACT_TOGGLE) continue;
            const uint32_t ci = front_index(p, *reinterpret_cast<const uint32_t *>(rows + 8 * i));
            if (ci == FC_NONE) continue;
            uint32_t w = fcv[i];
#pragma unroll 1
            for (int j = 0; j < nlog; ++j) {
                const uint32_t e = log[j];
                if ((e & 0xFFFF) == ci) w = e >> 16;
            }
            if ((w & 15) != T_DOOR || ((w >> 4) & 15) == S_LOCKED) continue;
            const uint32_t bit = 1u << ((w >> 8) & 7);
            if (flags & bit & FLAG_LH_MASK) continue;
            flags |= bit & FLAG_LH_MASK;
            rewmw += p.joint() ? ones : (1ull << (8 * i));
        }
    }
    if (term_all) term_some = ones;
    if (term_some) {  // hooks set both the agents' own flags and the returned dict
        toutw |= term_some;
#pragma unroll 1
        for (int k = 0; k < A; ++k)
            if (byte_of(term_some, k)) rows[8 * k + R_TERM] = 1;
    }
    if (p.kind() == MG_KIND_LOCKEDHALLWAY && __popc(flags & FLAG_LH_MASK) == p.num_rooms())
        toutw = ones;  // locked_hallway.py:213-215: only the returned dict

    MG_EXP(MG_TICK(q3);)
    store_bytes(lds + p.o.tout + slot * A, toutw | info_bits, A);
    store_bytes(lds + p.o.rewm + slot * A, rewmw, A);
    const bool all_term = toutw == ones;
    const int paid = static_cast<int>(__builtin_amdgcn_sad_u8(static_cast<uint32_t>(rewmw), 0u,
                                      __builtin_amdgcn_sad_u8(static_cast<uint32_t>(rewmw >> 32), 0u, 0u)));  // sum of the bytes
    const bool over = trunc || all_term;
    if (over) flags |= FLAG_OVER;
    h[H_FLAGS] = static_cast<uint8_t>(flags);
    if (!COMP && paid != 0) R = step_reward(p, step);
    reinterpret_cast<double *>(lds + p.o.R)[slot] = R;

    const bool will_reset = over && p.autoreset() != MG_AUTORESET_NONE;
    *reinterpret_cast<uint32_t *>(lds + p.o.flag + 4 * slot) =
        (trunc ? 1u : 0u) | (will_reset ? 0x100u : 0u) | (static_cast<uint32_t>(nlog) << 16) |
        (static_cast<uint32_t>(comp_cleared) << 24);

    // episode statistics: touched only when something happened (rare)
    if (over && !was_over) {
        atomicAdd(reinterpret_cast<unsigned long long *>(stats_shard() + 0), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long *>(stats_shard() + 1), static_cast<unsigned long long>(step));
        atomicAdd(reinterpret_cast<unsigned long long *>(stats_shard() + (all_term ? 2 : 3)), 1ull);
    }
    if (COMP ? paid_sum != 0.0 : paid != 0)
        atomicAdd(reinterpret_cast<unsigned long long *>(stats_shard() + 4),
                  static_cast<unsigned long long>(static_cast<long long>((COMP ? paid_sum : paid * R) * 16777216.0)));

    // cells changed by this step go back to HBM unless the whole grid is about to be replaced
    if (!will_reset) {
        uint16_t *gg = reinterpret_cast<uint16_t *>(late_arg<uint8_t *>(offsetof(KParams, grid)) + static_cast<size_t>(env) * p.GS);
#pragma unroll 1
        for (int j = 0; j < nlog; ++j) {
            const uint32_t e = log[j];
            gg[e & 0xFFFF] = static_cast<uint16_t>(e >> 16);
        }
#pragma unroll 1
        for (int j = 0; j < comp_cleared; ++j) {
            const uint32_t e = reinterpret_cast<const uint32_t *>(lds + p.o.log2)[slot * A + j];
            gg[e & 0xFFFF] = static_cast<uint16_t>(e >> 16);
        }
    }
    // (diagnostic builds: header read + reward + order draw, the agents' actions, hooks, flags + statistics + cells to HBM)
    MG_EXP(
    MG_TICK(q4);
    if (p.dbg_buf && (p.dbg & 16) && threadIdx.x == 0) {
        unsigned long long *d = p.dbg_buf + static_cast<size_t>(blockIdx.x) * 32 + 25;
        d[0] = q1 - q0;
        d[1] = q2 - q1;
        d[2] = q3 - q2;
        d[3] = q4 - q3;
    }
    )
}

// ---- act, one lane per agent -----------------------------------------------------------------------
//
// MultiGridEnv.step up to (not including) gen_obs, then the subclass hook: multigrid/base.py:395-396,440-533 and
// redbluedoors.py:176-185, blockedunlockpickup.py:165-167, locked_hallway.py:200-215 - for the NE environments
// of one sub-tile, every lane holding one agent.
//
// The grid itself is not needed.  Each agent acts once per step and only its own action moves or turns it, so the
// cell in front of every agent is known before the first action: those A cells (plus the two RedBlueDoors doors)
// travel with the header block - the observation pass of the previous launch left them there.  Cells written
// during the step go to a short per-environment log that later readers of the same cell consult.
//
// Agents of one environment rarely get in each other's way within a single step.  Every lane works out its
// agent's action against the state at the start of the step and publishes what the others need to know; a second
// look finds the pairs of agents for which the order of acting could matter: a cell in front of one that the
// other may write, or - where it matters whether the cell in front is occupied - an agent about to walk into or
// out of that cell.  Both agents of such a pair are marked.  For all others the result is final at once, whatever
// the order, and neither depends on nor affects what the marked agents do.  The marked agents then take their
// turns in the lanes, in the environment's order (multigrid/base.py:457-464): round n belongs to the agent at
// position n of its environment's order, rounds in which no marked agent of the wavefront acts are skipped, and an
// agent at its turn sees the rows and the cell log as the earlier rounds left them - the reference's sequential
// loop, run for the few agents it matters for.  Finally the environment's lanes settle the rest of the step
// through ballots: terminations, truncation, the hook, rewards, episode statistics, changed cells to HBM.
// The scratch array of the pair comparison lives in the image staging area, which is idle during the act phase.
// CROWD: the kernel variant for 6 to 8 agents that may not overlap, where most steps have agents to compare: the
// comparison goes through a cell-keyed table instead of over all pairs (kept out of the other variant, whose
// code it would only lengthen).
// AQ: 2 or 4 when the environments have exactly that many agents - an environment then sits in (half) a quad of
// lanes and its agents read each other's registers through quad permutes; 0: any number, through LDS.
template <bool CROWD, int AQ = 0>
__device__ __forceinline__ void act_lanes(const KParams &p, uint8_t *lds, int env0, int s0, int ns, int lane,
                                          int my_slot, int my_agent) {
    const int A = p.A();
    const bool mine = lane < ns * A;
    uint8_t *h = lds + p.o.hdr + (s0 + my_slot) * p.HP;
    uint8_t *a = h + H_ROWS + 8 * my_agent;
    uint2 *pair = reinterpret_cast<uint2 *>(lds + p.o.img);
    MG_EXP(unsigned long long x0 = __builtin_amdgcn_s_memtime(), x1 = 0, x2 = 0, x3 = 0, x4 = 0;)
    uint2 r0 = make_uint2(0u, 0u);
    ActEval ev;
    ev.r = r0;
    ev.fxy = ev.ci = ev.w = ev.nv = ev.bits = 0u;
    int my_act = ACT_DONE;
    uint32_t blue_xy = 0, red_xy = 0, w0 = 0, flags = 0;
    if (mine) {
        r0 = *reinterpret_cast<const uint2 *>(a);  // x y dir term | ct cc cs color
        w0 = reinterpret_cast<const uint16_t *>(h + p.HFC)[my_agent];
        const uint2 hm = *reinterpret_cast<const uint2 *>(h + H_FLAGS);  // flags, door positions
        my_act = lds[p.o.act + s0 * A + lane];
        flags = hm.x & 0xFF;
        red_xy = hm.y & 0xFFFF;
        blue_xy = hm.y >> 16;
        ev = eval_action(p, r0, w0, my_act, flags, blue_xy);
    }
    uint32_t lh_doors = red_xy, lh_distinct = blue_xy >> 8;  // LockedHallway with more than six rooms: meta[0..1], meta[3]
    const bool need_occ = mine && needs_occupancy(p, ev);
    bool occupied = false, clash = false, unstable = false;
    // when nobody may write a cell and nobody asks where the others stand (the usual case in open
    // rooms that allow overlap) there is nothing to compare
    if (CROWD && __ballot(need_occ || (ev.bits & EV_MAY_WRITE)) != 0) {
        // Up to 8 agents per environment: instead of comparing every pair, every agent marks the cell it stands on
        // and the cell it faces in a table with one 16-bit entry per grid cell (low byte: who stands here, high byte:
        // who faces this cell and acts on it) and looks up its own two cells; what the partners' actions are comes
        // from ballots.  The same conditions as in the pairwise form below, evaluated on 8-bit agent sets.
        uint32_t *tbl = reinterpret_cast<uint32_t *>(lds + p.o.tbl + my_slot * p.GP);
        const uint32_t px = r0.x & 0xFF, py = (r0.x >> 8) & 0xFF;
        const bool placed = mine && px < static_cast<uint32_t>(p.W()) && py < static_cast<uint32_t>(p.H());
        const uint32_t pc = px * p.H() + py, me = 1u << my_agent;
        const bool acts = mine && (ev.bits & EV_ONCELL);
        if (placed) atomicOr(tbl + (pc >> 1), me << (16 * (pc & 1)));
        if (acts) atomicOr(tbl + (ev.ci >> 1), (me << 8) << (16 * (ev.ci & 1)));
        wave_sync();
        const int sh = my_slot * A;
        const uint32_t amask = (1u << A) - 1u;
        const uint32_t occ_bits = EV_DROP | (p.overlap() ? 0u : EV_FWD);  // actions that ask who stands in front
        const uint32_t try_occ = EV_TRY_DROP | (p.overlap() ? 0u : EV_TRY_FWD);
        const uint32_t m_fwd = static_cast<uint32_t>(__ballot((ev.bits & EV_FWD) != 0) >> sh) & amask;
        const uint32_t m_wr = static_cast<uint32_t>(__ballot((ev.bits & EV_MAY_WRITE) != 0) >> sh) & amask;
        const uint32_t m_occ = static_cast<uint32_t>(__ballot((ev.bits & occ_bits) != 0) >> sh) & amask;
        uint32_t at_front = 0, same_front = 0, at_its_front = 0;
        if (acts) {
            const uint32_t ef = (tbl[ev.ci >> 1] >> (16 * (ev.ci & 1))) & 0xFFFF;
            at_front = ef & 0xFF;               // who stands on the cell in front of me
            same_front = (ef >> 8) & ~me;       // who else faces the cell I face
        }
        if (placed) at_its_front = ((tbl[pc >> 1] >> (16 * (pc & 1))) >> 8) & 0xFF;  // who faces the cell I stand on
        occupied = at_front != 0;  // base.py:485,512: every agent counts, terminated ones included
        unstable = acts && (same_front & m_wr) != 0;                                  // my result depends on when they act
        clash = need_occ && (m_fwd & (at_front | same_front)) != 0;
        clash |= (ev.bits & EV_MAY_WRITE) && same_front != 0;                        // ... or theirs on when I act
        clash |= (ev.bits & EV_FWD) && (m_occ & (at_its_front | same_front)) != 0;
        clash |= unstable;
        const uint64_t unst = __ballot(unstable);
        if (unst != 0) {  // see the comment in the pairwise form
            const uint32_t m_un = static_cast<uint32_t>(unst >> sh) & amask;
            const uint32_t m_tryfwd = static_cast<uint32_t>(__ballot((ev.bits & EV_TRY_FWD) != 0) >> sh) & amask;
            const uint32_t m_tryocc = static_cast<uint32_t>(__ballot((ev.bits & try_occ) != 0) >> sh) & amask;
            clash |= need_occ && (m_un & m_tryfwd & (at_front | same_front) & ~me) != 0;
            clash |= (ev.bits & EV_FWD) && (m_un & m_tryocc & (at_its_front | same_front) & ~me) != 0;
        }
        // Environments in which nobody may write a cell: the marked agents are all walking forward into cells that
        // others may occupy or leave (agents may not overlap), and whether each one moves only depends on who stands
        // in front of it when its turn comes.  Those turns are taken here, one agent per environment and round in
        // the environment's agent order, with the table's "stands here" sets as the current positions; nothing is
        // left for the in-order replay.  (Unmarked agents are where they were at the start of the step in the table:
        // by the marking rule none of them stands on, enters or leaves a cell that a marked agent faces.)
        const bool simple = mine && m_wr == 0;
        if (__ballot(clash && simple) != 0) {
            uint8_t *ordb = lds + p.o.ord + (s0 + my_slot) * A;
            if (p.order == nullptr && simple && my_agent == 0) {  // the environment's order of this step, see order_key()
                const uint4 h0 = *reinterpret_cast<const uint4 *>(h);
                const uint32_t key = order_key(p, env0 + s0 + my_slot, h0.y, static_cast<int>(h0.x) + 1);
                uint64_t ordw = 0x0706050403020100ull;
                uint32_t word = 0;
#pragma unroll 1
                for (int k = A - 1, f = 0; k > 0; --k, ++f) {
                    const int j = static_cast<int>((order_field(key, f, word) * static_cast<uint32_t>(k + 1)) >> 16);
                    const uint64_t d = ((ordw >> (8 * k)) ^ (ordw >> (8 * j))) & 0xFF;
                    ordw ^= (d << (8 * k)) ^ (d << (8 * j));
                }
#pragma unroll 1
                for (int k = 0; k < A; ++k) ordb[k] = static_cast<uint8_t>(ordw >> (8 * k));
            }
            wave_sync();
#pragma unroll 1
            for (int n = 0; n < A; ++n) {
                if (clash && simple && ordb[n] == my_agent) {  // base.py:464: my turn
                    occupied = ((tbl[ev.ci >> 1] >> (16 * (ev.ci & 1))) & 0xFF) != 0;  // base.py:485, as of now
                    if (!occupied) {  // base.py:489
                        atomicAnd(tbl + (pc >> 1), ~(me << (16 * (pc & 1))));
                        atomicOr(tbl + (ev.ci >> 1), me << (16 * (ev.ci & 1)));
                    }
                }
                wave_sync();
            }
            if (simple) clash = false;
        }
        wave_sync();
        // the table goes back to all-zero: the next sub-tile, or the next step, starts from a clean one
        if (placed) reinterpret_cast<uint16_t *>(tbl)[pc] = 0;
        if (acts) reinterpret_cast<uint16_t *>(tbl)[ev.ci] = 0;
        wave_sync();
    } else if (__ballot(need_occ || (ev.bits & EV_MAY_WRITE)) != 0) {
      if constexpr (AQ != 0) {
        // the same comparison as in the LDS form below, the other agents' words coming through quad permutes
        // (lane ^ 1, and with four agents lane ^ 2 and lane ^ 3: all of them other agents of my environment)
        const uint32_t my_pf = (r0.x & 0xFFFFu) | (ev.fxy << 16), my_pos = r0.x & 0xFFFFu;
        const uint32_t occ_bits = EV_DROP | (p.overlap() ? 0u : EV_FWD);  // actions that ask who stands in front
        const uint32_t try_occ = EV_TRY_DROP | (p.overlap() ? 0u : EV_TRY_FWD);
        uint32_t o_pf[3], o_bits[3];
#define MG_QUAD(x, ctrl) static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), ctrl, 0xF, 0xF, false))
        o_pf[0] = MG_QUAD(my_pf, 0xB1);  // quad_perm [1,0,3,2]
        o_bits[0] = MG_QUAD(ev.bits, 0xB1);
        if (AQ == 4) {
            o_pf[1] = MG_QUAD(my_pf, 0x4E);  // [2,3,0,1]
            o_bits[1] = MG_QUAD(ev.bits, 0x4E);
            o_pf[2] = MG_QUAD(my_pf, 0x1B);  // [3,2,1,0]
            o_bits[2] = MG_QUAD(ev.bits, 0x1B);
        }
        uint32_t rel[3];  // per partner: 1 it stands on the cell in front of me, 2 it faces the cell I face, 4 I stand in front of it
#pragma unroll
        for (int k = 0; k < AQ - 1; ++k) {
            const bool at_front = (o_pf[k] & 0xFFFFu) == ev.fxy, same_front = (o_pf[k] >> 16) == ev.fxy;
            const bool at_its_front = (o_pf[k] >> 16) == my_pos;
            rel[k] = (at_front ? 1u : 0u) | (same_front ? 2u : 0u) | (at_its_front ? 4u : 0u);
            occupied |= at_front;  // base.py:485,512: every agent counts, terminated ones included
            unstable |= (ev.bits & EV_ONCELL) && same_front && (o_bits[k] & EV_MAY_WRITE);
            clash |= need_occ && (o_bits[k] & EV_FWD) && (rel[k] & 3u);
            clash |= (o_bits[k] & EV_ONCELL) && same_front && (ev.bits & EV_MAY_WRITE);
            clash |= (o_bits[k] & occ_bits) && (ev.bits & EV_FWD) && (rel[k] & 6u);
        }
        clash |= unstable;
        if (__ballot(unstable) != 0) {  // see the comment in the LDS form
            const uint32_t un = unstable ? 1u : 0u;
            uint32_t o_un[3];
            o_un[0] = MG_QUAD(un, 0xB1);
            if (AQ == 4) {
                o_un[1] = MG_QUAD(un, 0x4E);
                o_un[2] = MG_QUAD(un, 0x1B);
            }
#pragma unroll
            for (int k = 0; k < AQ - 1; ++k) {
                if (!o_un[k]) continue;
                clash |= need_occ && (o_bits[k] & EV_TRY_FWD) && (rel[k] & 3u);
                clash |= (o_bits[k] & try_occ) && (ev.bits & EV_FWD) && (rel[k] & 6u);
            }
        }
#undef MG_QUAD
      } else {
        if (mine) pair[lane] = make_uint2((r0.x & 0xFFFFu) | (ev.fxy << 16), ev.bits);
        wave_sync();
        if (mine) {
            const uint2 *q = pair + my_slot * A;
            const uint32_t occ_bits = EV_DROP | (p.overlap() ? 0u : EV_FWD);  // actions that ask who stands in front
#pragma unroll 4
            for (int k = 0; k < A; ++k) {
                const uint2 o = q[k];
                const bool at_front = (o.x & 0xFFFFu) == ev.fxy;               // k stands on the cell in front of me
                const bool same_front = (o.x >> 16) == ev.fxy;                 // k faces the cell I face
                const bool at_its_front = (o.x >> 16) == (r0.x & 0xFFFFu);     // I stand on the cell in front of k
                occupied |= at_front;  // base.py:485,512: every agent counts, terminated ones included
                if (k != my_agent) {
                    // my result depends on when k acts ...
                    unstable |= (ev.bits & EV_ONCELL) && same_front && (o.y & EV_MAY_WRITE);
                    clash |= need_occ && (o.y & EV_FWD) && (at_front || same_front);
                    // ... or k's result on when I act
                    clash |= (o.y & EV_ONCELL) && same_front && (ev.bits & EV_MAY_WRITE);
                    clash |= (o.y & occ_bits) && (ev.bits & EV_FWD) && (at_its_front || same_front);
                }
            }
            clash |= unstable;
        }
        // An agent facing a cell that somebody else may write can end up doing something else than it
        // would at the start of the step (walk through a door just opened, fail to pick up what is gone):
        // towards such agents the others go by what the action could do at all, not by what it would do now.
        if (__ballot(unstable) != 0) {
            wave_sync();
            if (unstable) pair[lane].y = ev.bits | EV_UNSTABLE;
            wave_sync();
            if (mine) {
                const uint2 *q = pair + my_slot * A;
                const uint32_t try_occ = EV_TRY_DROP | (p.overlap() ? 0u : EV_TRY_FWD);
#pragma unroll 4
                for (int k = 0; k < A; ++k) {
                    const uint2 o = q[k];
                    if (k == my_agent || !(o.y & EV_UNSTABLE)) continue;
                    const bool at_front = (o.x & 0xFFFFu) == ev.fxy, same_front = (o.x >> 16) == ev.fxy;
                    const bool at_its_front = (o.x >> 16) == (r0.x & 0xFFFFu);
                    clash |= need_occ && (o.y & EV_TRY_FWD) && (at_front || same_front);
                    clash |= (o.y & try_occ) && (ev.bits & EV_FWD) && (at_its_front || same_front);
                }
            }
        }
      }
    }
    const uint64_t group = (A == 64 ? ~0ull : ((1ull << A) - 1ull)) << (my_slot * A);
    const int kind = p.kind();
    MG_EXP(x1 = __builtin_amdgcn_s_memtime();)
    finish_action(p, ev, r0, occupied);
    const bool final_here = mine && !clash;
    // (what happens in few steps - cells written, goals reached, hook events - is first asked of the whole wavefront:
    // the usual answer is "nowhere", and the per-environment questions are then never put)
    uint32_t *log = reinterpret_cast<uint32_t *>(lds + p.o.log) + (s0 + my_slot) * A;
    int nlog = 0;  // cells my environment has written so far
    if (const uint64_t wr_any = __ballot(final_here && (ev.bits & EV_WRITE))) {
        const uint64_t writers = wr_any & group;
        nlog = __popcll(writers);
        if (final_here && (ev.bits & EV_WRITE)) log[__popcll(writers & ((1ull << lane) - 1ull))] = ev.ci | (ev.nv << 16);
    }
    uint64_t resync = 0;  // blue-door toggles re-sync cell and object
    if (kind == MG_KIND_REDBLUEDOORS) resync = __ballot(final_here && (ev.bits & EV_FLIP) && (ev.bits & EV_BLUE));
    // my agent after its action (marked agents: not yet)
    uint2 row = final_here ? ev.r : r0;
    uint32_t my_term = (r0.x >> 24) | ((final_here && (ev.bits & EV_TERM)) ? 1u : 0u);
    bool my_goal = final_here && (ev.bits & EV_GOAL);
    if (final_here && (ev.bits & EV_ALIVE) && (ev.r.x != r0.x || ev.r.y != r0.y)) *reinterpret_cast<uint2 *>(a) = ev.r;
    MG_EXP(x2 = __builtin_amdgcn_s_memtime();)
    const uint64_t left = __ballot(clash);  // marked agents: they act one by one, in the order of base.py:457-464
    const bool replayed = (left & group) != 0;
    if (left != 0) {
        uint8_t *ordb = lds + p.o.ord + (s0 + my_slot) * A;
        int pos = -1;  // my agent's place in the order
        if (AQ != 0) {
            // two or four agents: every marked lane holds its environment's order in a register (given, or drawn as
            // order_key() says: every lane of the environment draws the same)
            uint32_t ordw;
            if (p.order != nullptr) {
                ordw = AQ == 4 ? *reinterpret_cast<const uint32_t *>(ordb) : *reinterpret_cast<const uint16_t *>(ordb);
            } else {
                const uint2 hs = *reinterpret_cast<const uint2 *>(h);  // step_count, episode
                const uint32_t key = order_key(p, env0 + s0 + my_slot, hs.y, static_cast<int>(hs.x) + 1);
                uint32_t word = 0;
                ordw = 0x03020100u;
#pragma unroll
                for (int k = AQ - 1, f = 0; k > 0; --k, ++f) {
                    const int j = static_cast<int>((order_field(key, f, word) * static_cast<uint32_t>(k + 1)) >> 16);
                    const uint32_t d = ((ordw >> (8 * k)) ^ (ordw >> (8 * j))) & 0xFF;
                    ordw ^= (d << (8 * k)) ^ (d << (8 * j));
                }
            }
#pragma unroll
            for (int n = AQ - 1; n >= 0; --n)
                if (((ordw >> (8 * n)) & 0xFF) == static_cast<uint32_t>(my_agent)) pos = n;
        } else {
            if (p.order == nullptr) {  // the environment's order of this step, drawn once per environment: see order_key()
                if (replayed && my_agent == 0) {
                    const uint2 hs = *reinterpret_cast<const uint2 *>(h);  // step_count, episode
                    const uint32_t key = order_key(p, env0 + s0 + my_slot, hs.y, static_cast<int>(hs.x) + 1);
                    uint32_t word = 0;
                    if (A <= 8) {
                        uint64_t ordw = 0x0706050403020100ull;
#pragma unroll 1
                        for (int k = A - 1, f = 0; k > 0; --k, ++f) {
                            const int j = static_cast<int>((order_field(key, f, word) * static_cast<uint32_t>(k + 1)) >> 16);
                            const uint64_t d = ((ordw >> (8 * k)) ^ (ordw >> (8 * j))) & 0xFF;
                            ordw ^= (d << (8 * k)) ^ (d << (8 * j));
                        }
#pragma unroll 1
                        for (int k = 0; k < A; ++k) ordb[k] = static_cast<uint8_t>(ordw >> (8 * k));
                    } else {
#pragma unroll 1
                        for (int k = 0; k < A; ++k) ordb[k] = static_cast<uint8_t>(k);
#pragma unroll 1
                        for (int k = A - 1, f = 0; k > 0; --k, ++f) {
                            const int j = static_cast<int>((order_field(key, f, word) * static_cast<uint32_t>(k + 1)) >> 16);
                            const uint8_t t = ordb[k];
                            ordb[k] = ordb[j];
                            ordb[j] = t;
                        }
                    }
                }
                wave_sync();
            }
            if (clash) {
#pragma unroll 1
                for (int n = A - 1; n >= 0; --n)
                    if (ordb[n] == my_agent) pos = n;
            }
        }
        if (resync & group) flags &= ~FLAG_STALE;
        MG_EXP(x3 = __builtin_amdgcn_s_memtime();)
        uint64_t acted = 0;  // marked agents that have had their turn
#pragma unroll 1
        for (int n = 0; n < A; ++n) {  // base.py:464
            const bool turn = clash && pos == n;
            const uint64_t now = __ballot(turn);
            if (now == 0) continue;
            MG_EXP(if (p.dbg_buf && threadIdx.x == 0) p.dbg_buf[static_cast<size_t>(blockIdx.x) * 32 + 26] += 1;)
            if (turn) {
                // The first marked agent of an environment acts on the state the step began with (what the unmarked
                // agents did does not concern it): its evaluation from above stands.  The others see what the marked
                // agents before them have done.
                if (acted & group) {
                    uint32_t w = w0;  // the cell in front as the agents before me have left it
#pragma unroll 1
                    for (int j = 0; j < nlog; ++j) {
                        const uint32_t e = log[j];
                        if ((e & 0xFFFF) == ev.ci) w = e >> 16;
                    }
                    ev = eval_action(p, r0, w, my_act, flags, blue_xy);
                    bool occ = false;  // base.py:485,512: every agent counts, terminated ones included
                    if (needs_occupancy(p, ev))
#pragma unroll 1
                        for (int k = 0; k < A; ++k) occ |= *reinterpret_cast<const uint16_t *>(h + H_ROWS + 8 * k) == ev.fxy;
                    finish_action(p, ev, r0, occ);
                }
                if (ev.bits & EV_ALIVE) {
                    if (ev.r.x != r0.x || ev.r.y != r0.y) *reinterpret_cast<uint2 *>(a) = ev.r;
                    row = ev.r;
                }
                if (ev.bits & EV_WRITE) log[nlog] = ev.ci | (ev.nv << 16);
                if (ev.bits & EV_TERM) my_term = 1;
                my_goal = (ev.bits & EV_GOAL) != 0;
            }
            acted |= now;
            if (__ballot(turn && (ev.bits & EV_WRITE)) & group) ++nlog;
            if (kind == MG_KIND_REDBLUEDOORS) {
                const uint64_t rs = __ballot(turn && (ev.bits & EV_FLIP) && (ev.bits & EV_BLUE));
                resync |= rs;
                if (rs & group) flags &= ~FLAG_STALE;  // grid.update re-syncs cell and object
            }
            wave_sync();  // the next round sees this round's rows and cells
        }
    }
    if (resync & group) flags &= ~FLAG_STALE;
    MG_EXP(x4 = __builtin_amdgcn_s_memtime();)

    // ---- the rest of the step, the environment's lanes agreeing through ballots ---------------------------
    // The hooks look at the state the actions left (they run after gen_obs in the reference, on the returned
    // dicts), agents in index order and - a quirk kept - terminated agents included:
    //   Empty                no hook;
    //   BlockedUnlockPickup  anybody carrying the box: success (blockedunlockpickup.py:165-167);
    //   LockedHallway        every door that is not locked, faces a toggling agent and has not paid yet pays once,
    //                        to the lowest-index such agent or to all (locked_hallway.py:200-215);
    //   RedBlueDoors         agents toggling towards the open blue door: success for each of them while the red door
    //                        is open, else failure for the first, which re-closes the door OBJECT and so ends the
    //                        matter for the others (redbluedoors.py:176-185).
    // the cell a toggling agent faces as the step leaves it (a toggle neither moves nor turns the agent)
    auto cell_at_end = [&](uint32_t ci, uint32_t at_start) {
        uint32_t w = at_start;
#pragma unroll 1
        for (int j = 0; j < nlog; ++j) {
            const uint32_t e = log[j];
            if ((e & 0xFFFF) == ci) w = e >> 16;
        }
        return w;
    };
    const bool toggles = mine && my_act == ACT_TOGGLE;
    const uint4 h0 = *reinterpret_cast<const uint4 *>(h);  // step_count | episode | flags | meta
    const int step = static_cast<int>(h0.x) + 1;           // base.py:395
    const uint32_t was_over = h0.z & FLAG_OVER;
    uint32_t my_tout = my_term;                  // base.py:400
    uint32_t my_rewm = my_goal ? 1u : 0u;        // base.py:558-562
    if (p.joint())
        if (const uint64_t goals = __ballot(mine && my_goal)) my_rewm = (goals & group) ? 1u : 0u;
    bool hook_term = false;  // the hook terminated my agent (its own flag too, not only the returned one)
    if (kind == MG_KIND_BLOCKEDUNLOCKPICKUP) {
        const bool carrier = mine && (row.y & 0xFF) == T_BOX;
        if (const uint64_t carriers = __ballot(carrier)) {
            if (carriers & group) {  // on_success, base.py:535-562
                hook_term = p.succ_any() || carrier;
                if (hook_term) my_tout = 1;
                if (p.joint() || carrier) my_rewm = 1;
            }
        }
    } else if (kind == MG_KIND_LOCKEDHALLWAY) {
        // Up to six rooms: one door colour each, a door is its colour (flags bits 0-5).  More rooms: colours repeat
        // (locked_hallway.py:150-151), a door is its room - 2 * row + side - in a 16-bit set kept in meta[0..1], and
        // the all-doors ending is reached at the number of DISTINCT colours, meta[3] (`self.rooms` is keyed by colour,
        // :160-166, :213).
        const bool many = p.num_rooms() > 6;
        uint32_t lh_bit = 0;  // the unlocked door I toggle towards
        if (toggles && (ev.fxy & 0xFF) < static_cast<uint32_t>(p.W()) && (ev.fxy >> 8) < static_cast<uint32_t>(p.H())) {
            const uint32_t wa = cell_at_end(ev.ci, w0);
            if ((wa & 15) == T_DOOR && ((wa >> 4) & 15) != S_LOCKED) {
                if (!many) {
                    lh_bit = (1u << ((wa >> 8) & 7)) & FLAG_LH_MASK;
                } else {
                    const uint32_t pitch = static_cast<uint32_t>(p.W() - 1) / 3u;  // room_size - 1
                    const uint32_t row = min((ev.fxy >> 8) / pitch, static_cast<uint32_t>(p.num_rooms() / 2 - 1));
                    lh_bit = 1u << (2u * row + ((ev.fxy & 0xFF) > static_cast<uint32_t>(p.W() / 2) ? 1u : 0u));
                }
            }
        }
        if (__ballot(lh_bit != 0)) {
            uint32_t newbits = 0, first = 0;
            if (!many) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const uint64_t m = __ballot(lh_bit == (1u << c)) & group;
                    if (m != 0 && !(flags & (1u << c))) {
                        newbits |= 1u << c;
                        if (lh_bit == (1u << c) && __builtin_ctzll(m) == lane) first = 1;  // lowest index: locked_hallway.py:200
                    }
                }
                flags |= newbits;
            } else {
                const uint32_t fresh = lh_bit & ~lh_doors;
                first = fresh != 0 ? 1u : 0u;
#pragma unroll 1
                for (int k = 0; k < A; ++k) {  // the environment's agents in index order
                    const uint32_t other = static_cast<uint32_t>(__shfl(static_cast<int>(fresh), my_slot * A + k));
                    newbits |= other;
                    if (k < my_agent && other == fresh) first = 0;  // a lower index toggles towards the same door
                }
                lh_doors |= newbits;
            }
            my_rewm += p.joint() ? static_cast<uint32_t>(__popc(newbits)) : first;  // `+=`, locked_hallway.py:207-211
        }
        if (many ? __popc(lh_doors) == static_cast<int>(lh_distinct) : __popc(flags & FLAG_LH_MASK) == p.num_rooms())
            my_tout = 1;  // only the returned dict, :213-215
    } else if (kind == MG_KIND_REDBLUEDOORS) {
        const bool cand = toggles && ev.fxy == blue_xy;
        if (const uint64_t cands_any = __ballot(cand)) {
            const uint64_t cands = cands_any & group;
            if (cands) {
                const uint16_t *fcv = reinterpret_cast<const uint16_t *>(h + p.HFC);
                const uint32_t wb = cell_at_end((blue_xy & 0xFF) * p.H() + (blue_xy >> 8), fcv[A + 1]);
                const uint32_t wr = cell_at_end((red_xy & 0xFF) * p.H() + (red_xy >> 8), fcv[A]);
                const int so = (flags & FLAG_STALE) ? S_CLOSED : static_cast<int>((wb >> 4) & 15);
                if ((wb & 15) == T_DOOR && so == S_OPEN) {
                    if (((wr >> 4) & 15) == S_OPEN) {  // on_success for every one of them, base.py:535-562
                        hook_term = p.succ_any() || cand;
                        if (p.joint() || cand) my_rewm = 1;
                    } else {  // on_failure for the first, base.py:564-585
                        hook_term = p.fail_any() || (cand && __builtin_ctzll(cands) == lane);
                        flags |= FLAG_STALE;
                    }
                    if (hook_term) my_tout = 1;
                }
            }
        }
    }
    MG_EXP(
    if (p.dbg_buf) {  // which of the rare paths this wavefront took (diagnostic builds)
        const uint64_t b_un = __ballot(unstable), b_wr = __ballot(mine && nlog > 0), b_pay = __ballot(mine && my_rewm != 0);
        const uint64_t b_cl = __ballot(clash);
        if (threadIdx.x == 0) {
            unsigned long long *d = p.dbg_buf + static_cast<size_t>(blockIdx.x) * 32;
            d[27] += b_wr != 0;
            d[28] += b_pay != 0;
            d[29] += b_un != 0;
            d[24] += __popcll(b_cl);
            d[11] += x1 - x0;                       // rows in, first evaluation, marking
            d[12] += x2 - x1;                       // finish, log, rows out
            d[13] += b_cl ? x3 - x2 : 0;            // order and places (wavefronts with marked agents)
            d[14] += b_cl ? x4 - x3 : x4 - x2;      // the rounds
            d[15] = x4;                             // (the rest of act_lanes runs from here to stamp 2)
        }
    }
    )
    {
        bool all_term = false;
        if (const uint64_t touts = __ballot(mine && my_tout != 0)) all_term = (touts & group) == group;
        const bool trunc = step >= p.max_steps;  // base.py:401-402
        const bool over = trunc || all_term;
        const bool will_reset = over && p.autoreset() != MG_AUTORESET_NONE;
        int paid = 0;
        if (__ballot(mine && my_rewm != 0))
            paid = __popcll(__ballot(mine && (my_rewm & 1)) & group) + 2 * __popcll(__ballot(mine && (my_rewm & 2)) & group) +
                   4 * __popcll(__ballot(mine && (my_rewm & 4)) & group) + 8 * __popcll(__ballot(mine && (my_rewm & 8)) & group);
        if (mine) {
            lds[p.o.oterm + s0 * A + lane] = static_cast<uint8_t>(my_term);  // what gen_obs sees (base.py:399)
            lds[p.o.tout + s0 * A + lane] = static_cast<uint8_t>(my_tout);
            lds[p.o.rewm + s0 * A + lane] = static_cast<uint8_t>(my_rewm);
            if (hook_term) a[R_TERM] = 1;
            // cells changed by this step go back to HBM unless the whole grid is about to be replaced: every agent
            // its own where all were independent (no two of them wrote one cell), else the log in its order
            if (!will_reset && nlog > 0) {
                uint16_t *gg = reinterpret_cast<uint16_t *>(late_arg<uint8_t *>(offsetof(KParams, grid)) +
                                                            static_cast<size_t>(env0 + s0 + my_slot) * p.GS);
                if (!replayed) {
                    if (ev.bits & EV_WRITE) gg[ev.ci] = static_cast<uint16_t>(ev.nv);
                } else if (my_agent == 0) {
#pragma unroll 1
                    for (int j = 0; j < nlog; ++j) {
                        const uint32_t e = log[j];
                        gg[e & 0xFFFF] = static_cast<uint16_t>(e >> 16);
                    }
                }
            }
            if (my_agent == 0) {
                const double R = paid != 0 ? step_reward(p, step) : 0.0;  // nobody paid: every reward is 0 * R
                *reinterpret_cast<int *>(h + H_STEP) = step;
                h[H_FLAGS] = static_cast<uint8_t>(flags | (over ? FLAG_OVER : 0u));
                if (kind == MG_KIND_LOCKEDHALLWAY && p.num_rooms() > 6)
                    *reinterpret_cast<uint16_t *>(h + H_META) = static_cast<uint16_t>(lh_doors);
                reinterpret_cast<double *>(lds + p.o.R)[s0 + my_slot] = R;
                *reinterpret_cast<uint32_t *>(lds + p.o.flag + 4 * (s0 + my_slot)) =
                    (trunc ? 1u : 0u) | (will_reset ? 0x100u : 0u) | (static_cast<uint32_t>(nlog) << 16);
                // episode statistics: touched only when something happened (rare)
                if (over && !was_over) {
                    atomicAdd(reinterpret_cast<unsigned long long *>(stats_shard() + 0), 1ull);
                    atomicAdd(reinterpret_cast<unsigned long long *>(stats_shard() + 1), static_cast<unsigned long long>(step));
                    atomicAdd(reinterpret_cast<unsigned long long *>(stats_shard() + (all_term ? 2 : 3)), 1ull);
                }
                if (paid != 0)
                    atomicAdd(reinterpret_cast<unsigned long long *>(stats_shard() + 4),
                              static_cast<unsigned long long>(static_cast<long long>(paid * R * 16777216.0)));
            }
        }
    }
    wave_sync();
}

// Replace one environment's header in LDS and its grid in HBM by a start state from the pool
// (MultiGridEnv.reset, multigrid/base.py:310-362, with the layout precomputed on the host).
// Executed by the whole wavefront for one environment at a time.  `lds_grid`: where the environment's grid lies in
// LDS when it is there already (it then takes the new grid too), else null.  Header block and grid are asked for
// together and answered by one trip to memory.
__device__ __forceinline__ void reload_env(const KParams &p, uint8_t *lds, int slot, int env, int layout,
                                           uint32_t episode, int lane, uint8_t *lds_grid) {
    // (restarts are rare: the pool's addresses are fetched here rather than kept for the whole step)
    const uint4 *sh = reinterpret_cast<const uint4 *>(late_arg<const uint8_t *>(offsetof(KParams, pool_hdr)) + static_cast<size_t>(layout) * p.HS);
    const uint4 *sg = reinterpret_cast<const uint4 *>(late_arg<const uint8_t *>(offsetof(KParams, pool_grid)) + static_cast<size_t>(layout) * p.GS);
    uint4 *dh = reinterpret_cast<uint4 *>(lds + p.o.hdr + slot * p.HP);
    uint4 *gg = reinterpret_cast<uint4 *>(late_arg<uint8_t *>(offsetof(KParams, grid)) + static_cast<size_t>(env) * p.GS);
    uint4 *lg = reinterpret_cast<uint4 *>(lds_grid);
    const int nh = p.HS / 16, ng = p.GS / 16;  // (a header block has at most 42 pieces: one per lane)
    uint4 hv = make_uint4(0u, 0u, 0u, 0u), gv = hv;
    if (lane < nh) hv = sh[lane];
    if (lane < ng) gv = sg[lane];
    if (lane == 0) hv.y = episode;  // keep the environment's own episode counter
    if (lane < nh) dh[lane] = hv;
    if (lane < ng) {
        gg[lane] = gv;
        if (lg) lg[lane] = gv;
    }
#pragma unroll 1
    for (int i = 64 + lane; i < ng; i += 64) {
        const uint4 v = sg[i];
        gg[i] = v;
        if (lg) lg[i] = v;
    }
}

// ---- observe: one lane per agent ------------------------------------------------------------------

// gen_obs_grid_encoding for one agent: multigrid/utils/obs.py:64-99 (gen_obs_grid :127-203,
// get_view_exts :272-310, get_vis_mask :231-269), written for a grid that already carries
// the agent overlay.  The image comes back in registers, as the words of its [u][v][3] byte stream.
template <int V, bool SEE, bool WALLED>
__device__ __forceinline__ void observe_agent(const KParams &p, uint2 row, const uint8_t *gbytes,
                                              uint32_t (&px)[(3 * V * V + 3) / 4]) {
    constexpr int HALF = V / 2;
    const int W = p.W(), H = p.H();
    const int x = row.x & 0xFF, y = (row.x >> 8) & 0xFF, dir = (row.x >> 16) & 3;  // x y dir term | ct cc cs color
    // forward f and right r unit vectors; view cell (u, v) sits (u - HALF) to the right of and
    // (V - 1 - v) ahead of the agent, which is the rotation table of obs.py:174-190
    const int fx = dir_dx(dir), fy = dir_dy(dir);
    const int rx = dir_dx((dir + 1) & 3), ry = dir_dy((dir + 1) & 3);
    const bool r_on_x = rx != 0;
    const int cr = r_on_x ? x : y, lr = r_on_x ? W : H, sr = rx + ry;  // coordinate moved by u
    const int cf = r_on_x ? y : x, lf = r_on_x ? H : W, sf = fx + fy;  // coordinate moved by v
    // byte offsets: cell index = X * H + Y, two bytes per cell
    const int step_u = 2 * (rx * H + ry);
    const int step_f = 2 * (fx * H + fy);
    const int base = 2 * (x * H + y);

    // Per-u and per-v: is the world coordinate inside the grid, and the byte offset contribution of
    // the coordinate clamped into the grid.  A view cell outside the grid reads as a wall
    // (obs.py:193-196); where every grid is ringed by exactly such walls the clamped read already
    // returns one and no per-cell correction is needed.
    int off_u[V], off_v[V];
    uint32_t in_u = 0, in_v = 0;
    const int stride_r = r_on_x ? 2 * H : 2, stride_f = r_on_x ? 2 : 2 * H;
#pragma unroll
    for (int u = 0; u < V; ++u) {
        const int c = cr + (u - HALF) * sr;
        in_u |= (static_cast<unsigned>(c) < static_cast<unsigned>(lr) ? 1u : 0u) << u;
        off_u[u] = (min(max(c, 0), lr - 1) - cr) * stride_r;
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int c = cf + (V - 1 - v) * sf;
        in_v |= (static_cast<unsigned>(c) < static_cast<unsigned>(lf) ? 1u : 0u) << v;
        off_v[v] = base + (min(max(c, 0), lf - 1) - cf) * stride_f;
    }

    uint32_t cell[V * V];
    uint32_t opq[V];  // per view row v: bit u set when cell (u, v) blocks sight
#pragma unroll
    for (int v = 0; v < V; ++v) opq[v] = 0;
#pragma unroll
    for (int u = 0; u < V; ++u) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            uint32_t w = *reinterpret_cast<const uint16_t *>(gbytes + off_v[v] + off_u[u]);
            // the load zero-extends; from here on the cell is a 32-bit value (keeps the compiler from
            // re-deriving a 16-bit one and widening it again at every use)
            asm("" : "+v"(w));
            cell[u * V + v] = w;
        }
    }
    if (!WALLED) {
#pragma unroll
        for (int u = 0; u < V; ++u) {
            const uint32_t row_in = ((in_u >> u) & 1u) ? in_v : 0u;
#pragma unroll
            for (int v = 0; v < V; ++v) cell[u * V + v] = ((row_in >> v) & 1u) ? cell[u * V + v] : CELL_WALL;
        }
    }
    // the agent's own cell shows what it carries, obs.py:201
    cell[HALF * V + (V - 1)] = pack_cell(row.y & 0xFF, (row.y >> 8) & 0xFF, (row.y >> 16) & 0xFF);

    if (!SEE) {
#pragma unroll
        for (int u = 0; u < V; ++u)
#pragma unroll
            for (int v = 0; v < V; ++v) opq[v] |= ((cell[u * V + v] >> 15) & 1u) << u;

        // get_vis_mask, obs.py:231-269, on V-bit rows.  In one row, a visible see-through cell
        // lights its neighbour in the pass direction; a run of see-through cells therefore
        // carries like a binary addition: (S + x) ^ S marks every cell from an emitter x to
        // one past the end of its run.
        constexpr uint32_t FULL = (1u << V) - 1u;
        uint32_t vis[V];
        uint32_t cur = 1u << HALF;  // obs.py:249
#pragma unroll
        for (int v = V - 1; v >= 0; --v) {
            const uint32_t S = ~opq[v] & FULL;
            // forward pass i = 0 .. V-2 (obs.py:254-259)
            const uint32_t F = (cur | ((S + (cur & S)) ^ S)) & FULL;
            const uint32_t EA = F & S & (FULL >> 1);
            // backward pass i = V-1 .. 1 (obs.py:262-267), same trick on the bit-reversed row
            const uint32_t rS = __brev(S), rF = __brev(F);
            const uint32_t B = (F | __brev((rS + (rF & rS)) ^ rS)) & FULL;
            const uint32_t EB = B & S & (FULL & ~1u);
            vis[v] = B;
            cur = (EA | (EA << 1) | EB | (EB >> 1)) & FULL;  // cells lit in the row above
        }
        // (type, colour, state) byte triples: type and colour already sit in the low nibbles of bytes
        // 0 and 1, the state nibble moves up by 12 bits; cells out of sight become UNSEEN (0,0,0),
        // obs.py:14,97, through the same mask that clears the unused nibbles
#pragma unroll
        for (int u = 0; u < V; ++u)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const uint32_t keep = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(vis[v]), u, 1)) & 0x0F0F0Fu;
                cell[u * V + v] = (cell[u * V + v] | (cell[u * V + v] << 12)) & keep;
            }
    } else {
#pragma unroll
        for (int k = 0; k < V * V; ++k) cell[k] = (cell[k] | (cell[k] << 12)) & 0x0F0F0Fu;
    }
    // four triples make three words of the byte stream
#pragma unroll
    for (int k = 0; k + 3 < V * V; k += 4) {
        px[3 * (k / 4) + 0] = __builtin_amdgcn_perm(cell[k + 1], cell[k + 0], 0x04020100u);
        px[3 * (k / 4) + 1] = __builtin_amdgcn_perm(cell[k + 2], cell[k + 1], 0x05040201u);
        px[3 * (k / 4) + 2] = __builtin_amdgcn_perm(cell[k + 3], cell[k + 2], 0x06050402u);
    }
    static_assert(V * V % 4 == 1, "odd view sizes leave one trailing cell");
    px[3 * (V * V / 4)] = cell[V * V - 1];
}

// ---- one-hot images ----------------------------------------------------------------------------------
//
// OneHotObsWrapper.one_hot, multigrid/wrappers.py:114-146, with dim_sizes = (11 types, 6 colours,
// 7 states) of wrappers.py:95: a cell (t, c, s) becomes 24 bytes with ones at t, 11 + c and 17 + s.
// The 24 bytes are three little-endian 64-bit words; a one at byte k of the cell is 1 << 8k taken
// modulo 64 and routed to the word k / 8.  Values the reference cannot produce set a byte only
// while it stays inside the cell's 24.
constexpr int OH_DIM = 24;

__device__ __forceinline__ void one_hot_cell(uint32_t tcs, uint64_t &lo, uint64_t &mid, uint64_t &hi) {
    const uint32_t t = tcs & 0xFF, c = (tcs >> 8) & 0xFF, s = (tcs >> 16) & 0xFF;
    const uint64_t x = 1ull << ((8 * t) & 63);        // byte t
    const uint64_t y = 1ull << ((8 * (c + 3)) & 63);  // byte 11 + c
    const uint64_t z = 1ull << ((8 * (s + 1)) & 63);  // byte 17 + s
    lo = t < 8 ? x : 0ull;
    mid = ((t >= 8 && t < 16) ? x : 0ull) | (c < 5 ? y : 0ull);
    hi = ((c >= 5 && c < 13) ? y : 0ull) | (s < 7 ? z : 0ull);
}

// `n_cells` consecutive cells of the image stream at `src` (LDS, three bytes per cell) expanded to
// `dst` (8-byte aligned).  Cell k belongs to agent first_agent + k / (V * V) of the sub-tile; cells of
// environments outside `slot_mask` are skipped.  With every environment taking part and a 16-byte
// aligned destination, each lane expands two cells (48 bytes); the pieces then change lanes through
// LDS so that every store instruction of the wavefront covers one contiguous KiB.
constexpr int OH_XCHG_BYTES = 64 * 2 * OH_DIM;

template <int V>
__device__ __forceinline__ void emit_one_hot(const KParams &p, uint8_t *lds, const uint8_t *src, int n_cells,
                                             uint8_t *dst, int lane, bool all, uint64_t slot_mask, int first_agent) {
    if (all && (n_cells & 1) == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) {
        const uint16_t *s16 = reinterpret_cast<const uint16_t *>(src);
        uint4 *xb = reinterpret_cast<uint4 *>(lds + p.o.oh);
        const int pairs = n_cells / 2;
#pragma unroll 1
        for (int j0 = 0; j0 < pairs; j0 += 64) {
            const int j = j0 + lane;
            if (j < pairs) {
                const uint32_t a = s16[3 * j], b = s16[3 * j + 1], c = s16[3 * j + 2];
                uint64_t l0, m0, h0, l1, m1, h1;
                one_hot_cell(a | ((b & 0xFFu) << 16), l0, m0, h0);
                one_hot_cell((b >> 8) | (c << 8), l1, m1, h1);
                uint4 *d = xb + 3 * lane;
                d[0] = make_uint4(static_cast<uint32_t>(l0), static_cast<uint32_t>(l0 >> 32), static_cast<uint32_t>(m0),
                                  static_cast<uint32_t>(m0 >> 32));
                d[1] = make_uint4(static_cast<uint32_t>(h0), static_cast<uint32_t>(h0 >> 32), static_cast<uint32_t>(l1),
                                  static_cast<uint32_t>(l1 >> 32));
                d[2] = make_uint4(static_cast<uint32_t>(m1), static_cast<uint32_t>(m1 >> 32), static_cast<uint32_t>(h1),
                                  static_cast<uint32_t>(h1 >> 32));
            }
            wave_sync();
            const int pieces = 3 * min(64, pairs - j0);
            uint4 *out = reinterpret_cast<uint4 *>(dst + static_cast<size_t>(2 * OH_DIM) * j0);
            uint4 q0, q1, q2;
            q0 = xb[min(lane, pieces - 1)];
            q1 = xb[min(64 + lane, pieces - 1)];
            q2 = xb[min(128 + lane, pieces - 1)];
            if (lane < pieces) store_out16(out + lane, q0, p.stream_onehot());
            if (64 + lane < pieces) store_out16(out + 64 + lane, q1, p.stream_onehot());
            if (128 + lane < pieces) store_out16(out + 128 + lane, q2, p.stream_onehot());
            wave_sync();
        }
        return;
    }
#pragma unroll 1
    for (int k = lane; k < n_cells; k += 64) {
        const int slot = udiv(first_agent + k / (V * V), p.magic_a);
        if (!((slot_mask >> slot) & 1)) continue;
        const uint8_t *b = src + 3 * k;
        uint64_t l0, m0, h0;
        one_hot_cell(b[0] | (static_cast<uint32_t>(b[1]) << 8) | (static_cast<uint32_t>(b[2]) << 16), l0, m0, h0);
        uint64_t *d = reinterpret_cast<uint64_t *>(dst + static_cast<size_t>(OH_DIM) * k);
        d[0] = l0;
        d[1] = m0;
        d[2] = h0;
    }
}

// ---- the kernel ----------------------------------------------------------------------------------------
//
// One wavefront owns ET = G * NE consecutive environments.  Header blocks of all ET live in LDS for
// the whole launch; grids are brought in NE environments ("sub-tile", NE * A <= 64 agent lanes)
// at a time for the observation passes.

// MULTI: the launch takes p.T consecutive steps (mg_step_n).  Header blocks AND the grids of all ET environments
// then stay in LDS from the first step to the last; per step only actions come in and observations, rewards and
// flags go out (plus the few grid cells a step changes, which still go to HBM at once, and restarts from the pool).
// COMP: a step of CompetativeRedBlueDoor environments (their hook runs after gen_obs and moves agents).
// STG: agents whose images share the LDS staging buffer on their way out (16, 32 or 64; the host picks per batch).
template <int V, bool SEE, int MODE, bool WALLED, bool OH, bool MULTI = false, bool COMP = false,
          bool CROWD = false, int STG = MG_STAGE_LANES>
#ifndef MG_KERNEL_ATTR
#define MG_KERNEL_ATTR
#endif
// Wavefronts per SIMD the register allocation must leave room for.  The single-step kernels of see-through
// environments sit right at 128 registers (four wavefronts; measured 7-10 % faster than three): the hint keeps them
// there without a spill when an unrelated change nudges the allocator.  The others need 139-170 registers and lose
// more to spills than a fourth wavefront gives (measured), so they are left alone.  Tuning builds can override.
#ifndef MG_WAVES_SINGLE
#define MG_WAVES_SINGLE ((SEE && !OH && !COMP && MODE == MODE_STEP && V <= 7) ? 4 : 1)
#endif
#ifndef MG_WAVES_MULTI
// the multi-step kernels need 146-168 registers: three wavefronts (168) measured best; the cell-table and competitive
// variants of environments with opaque walls need a few more and would spill at 168, so they get two
#define MG_WAVES_MULTI (((COMP || CROWD) && !SEE) ? 2 : 3)
#endif
__global__ __launch_bounds__(64) MG_KERNEL_ATTR
__attribute__((amdgpu_waves_per_eu(MULTI ? MG_WAVES_MULTI : MG_WAVES_SINGLE, 8))) void mg_tile_kernel(const KParams p0) {
    extern __shared__ __align__(16) uint8_t lds[];
    static_assert(!MULTI || (MODE == MODE_STEP && !OH), "the multi-step kernel steps and writes compact images");
    static_assert(!COMP || (MODE == MODE_STEP && !OH && !WALLED), "CompetativeRedBlueDoor steps: one variant");
    static_assert(!CROWD || (MODE == MODE_STEP && !OH && !COMP), "the cell table serves steps of up to 8 agents");
    static_assert(STG == 16 || STG == 32 || STG == 64, "staging width: whole 16-agent blocks of a wavefront");
    // the multi-step kernel advances its input / output pointers from step to step; otherwise p is never written
    typename std::conditional<MULTI, KParams, const KParams &>::type p = p0;
    constexpr int IMG = 3 * V * V;  // bytes per agent image
    const int lane = threadIdx.x;
    const int A = p.A(), NE = p.NE(), NC = A + 2;
    const int ET = p.G() * NE;
    const int e0 = blockIdx.x * ET;
    const int net = min(ET, p.E - e0);
    const int nag = net * A;
    int *lsrc = reinterpret_cast<int *>(lds + p.o.src);

    MG_STAMP(0);
    // (diagnostic builds: MGRID_DBG bit 6 delays the tiles whose index has bit ((dbg >> 16) & 31) set by
    // ((dbg >> 8) & 255) sleeps of 127 x 64 clocks, so that the wavefronts of a SIMD are not all in the same phase)
    MG_EXP(
    unsigned long long tstart = 0;
    MG_TICK(tstart);
    if ((p.dbg & 64) && ((blockIdx.x >> ((p.dbg >> 16) & 31)) & 1))
        for (int i = 0, n = (p.dbg >> 8) & 255; i < n; ++i) __builtin_amdgcn_s_sleep(127);
    )
    // which environments take part (a masked reset skips some)
    bool active = lane < net;
    if (MODE == MODE_RESET && active && p.mask) active = p.mask[e0 + lane] != 0;
    const uint64_t active_mask = __ballot(active);
    if (active_mask == 0) return;

    const int per = p.GS / 16;
    const bool want_img = p.image != nullptr || (OH && p.onehot != nullptr);
    const bool passes = (want_img || MODE == MODE_STEP) MG_EXP(&& !(p.dbg & 32));
    // Grid blocks travel HBM -> registers -> LDS one sub-tile ahead: the loads for sub-tile g + 1
    // are issued before sub-tile g is computed and land in LDS after its observation has read
    // the previous grid, just before its image is stored; so neither the load latency nor the
    // completion of the image stores sits on the critical path.  A sub-tile's grids are one
    // contiguous block of the state array, so the loads need no per-lane address arithmetic
    // beyond lane * 16; environments restarted in this launch are patched in from the pool
    // afterwards (fixup), which is rare enough to stay a plain loop.
    const int gpq = (p.GP - p.GS) >> 4;  // 16-byte pieces between two grids in LDS
    uint4 gb0, gb1, gb2, gb3, gb4;
#if MG_PREFETCH_REGS != 5
    uint4 gb5, gb6, gb7;
#endif
#define MG_ISSUE1(c, r) if ((c) * 64 < n16_) r = src_[min((c) * 64 + lane, n16_ - 1)];
#define MG_COMMIT1(c, r) \
    if ((c) * 64 < n16_ && (c) * 64 + lane < n16_) dg_[(c) * 64 + lane + gpq * udiv((c) * 64 + lane, p.magic_per)] = r;
// 16-byte registers per lane for the grid prefetch (1 KiB of a sub-tile's grids each; what exceeds them is
// loaded when the sub-tile starts).  Five cover the sub-tiles of every registered configuration up to 16 x 16
// cells x 8 agents and keep the kernel 19 registers smaller than eight do.
#ifndef MG_PREFETCH_REGS
#define MG_PREFETCH_REGS 5
#endif
#if MG_PREFETCH_REGS == 5
#define MG_EACH(OP) OP(0, gb0) OP(1, gb1) OP(2, gb2) OP(3, gb3) OP(4, gb4)
#else
#define MG_EACH(OP) OP(0, gb0) OP(1, gb1) OP(2, gb2) OP(3, gb3) OP(4, gb4) OP(5, gb5) OP(6, gb6) OP(7, gb7)
#endif
    auto issue = [&](int s0_) {
        const int n16_ = min(NE, net - s0_) * per;
        const uint4 *src_ = reinterpret_cast<const uint4 *>(p.grid + static_cast<size_t>(e0 + s0_) * p.GS);
        MG_EACH(MG_ISSUE1)
    };
    auto commit = [&](int s0_) {
        const int ns_ = min(NE, net - s0_);
        const int n16_ = ns_ * per;
        uint4 *dg_ = reinterpret_cast<uint4 *>(lds + p.o.grid);
        MG_EACH(MG_COMMIT1)
        if (n16_ > 64 * MG_PREFETCH_REGS) {  // grids beyond what the prefetch registers hold
            const uint4 *src_ = reinterpret_cast<const uint4 *>(p.grid + static_cast<size_t>(e0 + s0_) * p.GS);
#pragma unroll 1
            for (int i = 64 * MG_PREFETCH_REGS + lane; i < n16_; i += 64) dg_[i + gpq * udiv(i, p.magic_per)] = src_[i];
        }
    };
    // environments of a sub-tile that were restarted in this launch: their grid in LDS comes from the layout pool
    // (`gslot_`: where the sub-tile's grids start in LDS)
    auto patch = [&](int s0_, int gslot_) {
        const int ns_ = min(NE, net - s0_);
        if (MODE != MODE_OBS && __ballot(lane < ns_ && lsrc[s0_ + lane] >= 0)) {
            const uint8_t *pool_grid_ = late_arg<const uint8_t *>(offsetof(KParams, pool_grid));
            const int n16_ = ns_ * per;
            uint4 *dg_ = reinterpret_cast<uint4 *>(lds + p.o.grid + gslot_ * p.GP);
#pragma unroll 1
            for (int i = lane; i < n16_; i += 64) {
                const int sub = udiv(i, p.magic_per);
                const int src = lsrc[s0_ + sub];
                if (src >= 0)
                    dg_[i + gpq * sub] = reinterpret_cast<const uint4 *>(pool_grid_ + static_cast<size_t>(src) * p.GS)[i - sub * per];
            }
        }
    };
    // multi-step kernel: every grid of the tile has its own place in LDS (environment s at s * (GS + 16))
    auto load_all_grids = [&]() {
        const int n16_ = net * per;
        const uint4 *src_ = reinterpret_cast<const uint4 *>(p.grid + static_cast<size_t>(e0) * p.GS);
        uint4 *dg_ = reinterpret_cast<uint4 *>(lds + p.o.grid);
#pragma unroll 1
        for (int i = lane; i < n16_; i += 64) dg_[i + gpq * udiv(i, p.magic_per)] = src_[i];
    };
#undef MG_ISSUE1
#undef MG_COMMIT1
#undef MG_EACH
    // ---- header blocks, actions, order and the first sub-tile's grids: every load goes out before the
    // first of them is waited for; environments that turn out to be restarted get their grid from the
    // pool later ----------------------------------------------------------------------------------------
    {
        const uint4 *sh = reinterpret_cast<const uint4 *>(p.hdr + static_cast<size_t>(e0) * p.HS);
        uint4 *dh = reinterpret_cast<uint4 *>(lds + p.o.hdr);
        const int nh = MODE != MODE_RESET ? net * p.HS / 16 : 0;
        uint4 hb0, hb1, hb2, hb3;
        if (0 < nh) hb0 = sh[min(lane, nh - 1)];
        if (64 < nh) hb1 = sh[min(64 + lane, nh - 1)];
        if (128 < nh) hb2 = sh[min(128 + lane, nh - 1)];
        if (192 < nh) hb3 = sh[min(192 + lane, nh - 1)];
        const uint8_t *ga = nullptr, *go = nullptr;
        bool words = false;
        uint32_t aw = 0, ow = 0;
        if (MODE == MODE_STEP) {
            ga = p.actions + static_cast<size_t>(e0) * A;
            go = p.order ? p.order + static_cast<size_t>(e0) * A : nullptr;
            words = ((reinterpret_cast<uintptr_t>(ga) | reinterpret_cast<uintptr_t>(go)) & 3) == 0;
            if (words && 4 * lane + 4 <= nag) {
                aw = *reinterpret_cast<const uint32_t *>(ga + 4 * lane);
                if (go) ow = *reinterpret_cast<const uint32_t *>(go + 4 * lane);
            }
        }
        if (MULTI) load_all_grids(); else
        if (passes && MODE != MODE_RESET) issue(0);
        wave_sync();
        // (in LDS the blocks lie HP bytes apart: piece i of the contiguous HBM block belongs to environment i / (HS / 16))
        const int hpad = (p.HP - p.HS) >> 4;
        auto hslot = [&](int i) { return hpad ? i + hpad * udiv(i, p.magic_ph) : i; };
        if (lane < nh) dh[hslot(lane)] = hb0;
        if (64 + lane < nh) dh[hslot(64 + lane)] = hb1;
        if (128 + lane < nh) dh[hslot(128 + lane)] = hb2;
        if (192 + lane < nh) dh[hslot(192 + lane)] = hb3;
#pragma unroll 1
        for (int i = 256 + lane; i < nh; i += 64) dh[hslot(i)] = sh[i];
        if (MODE == MODE_STEP) {
            if (words && 4 * lane + 4 <= nag) {
                *reinterpret_cast<uint32_t *>(lds + p.o.act + 4 * lane) = aw;
                if (go) *reinterpret_cast<uint32_t *>(lds + p.o.ord + 4 * lane) = ow;
            }
#pragma unroll 1
            for (int b = words ? min(nag & ~3, 256) + lane : lane; b < nag; b += 64) {  // what the words left over
                lds[p.o.act + b] = ga[b];
                if (go) lds[p.o.ord + b] = go[b];
            }
        }
        if (lane < ET) lsrc[lane] = -1;
        if (CROWD) {
            // act_lanes()' cell table starts all-zero (and is left so after every use)
            uint4 *tz = reinterpret_cast<uint4 *>(lds + p.o.tbl);
            const int nz = NE * p.GP / 16;
#pragma unroll 1
            for (int i = lane; i < nz; i += 64) tz[i] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    wave_sync();

    if (MODE == MODE_RESET) {
        // pull start states from the pool: header -> LDS, grid -> HBM
        uint32_t episode = 0;
        int layout = 0;
        if (active) {
            const int env = e0 + lane;
            const uint32_t old = *reinterpret_cast<const uint32_t *>(p.hdr + static_cast<size_t>(env) * p.HS + H_EPISODE);
            episode = p.serial == 0 ? 0u : old + 1u;
            layout = p.layout_idx ? p.layout_idx[env] : pick_layout(p, env, episode);
            layout = min(max(layout, 0), p.num_layouts - 1);
            lsrc[lane] = layout;
        }
        wave_sync();
        // Every 16-byte piece of the tile's header blocks and grids comes from the pool entry of its environment:
        // the pieces are independent, so four are asked for before the first is stored (a reset of many environments
        // is a gather, not a chain of restarts).
        const uint8_t *pool_hdr_ = late_arg<const uint8_t *>(offsetof(KParams, pool_hdr));
        const uint8_t *pool_grid_ = late_arg<const uint8_t *>(offsetof(KParams, pool_grid));
        uint4 *dh = reinterpret_cast<uint4 *>(lds + p.o.hdr);
        uint4 *dg = reinterpret_cast<uint4 *>(late_arg<uint8_t *>(offsetof(KParams, grid)) + static_cast<size_t>(e0) * p.GS);
        const int per_h = p.HS / 16, hpad = (p.HP - p.HS) >> 4;
        // (loads never depend on a condition: a piece beyond the end, or of an environment that is not reset, is read
        // from a valid place and not stored)
        const int n_h = net * per_h, n_g = net * per;
        auto hdr_piece = [&](int i, bool &ok, int &at) {
            const int ic = min(i, n_h - 1), slot = udiv(ic, p.magic_ph);
            ok = i < n_h && ((active_mask >> slot) & 1);
            at = ic + hpad * slot;
            return reinterpret_cast<const uint4 *>(pool_hdr_ + static_cast<size_t>(max(lsrc[slot], 0)) * p.HS)[ic - slot * per_h];
        };
        auto grid_piece = [&](int i, bool &ok) {
            const int ic = min(i, n_g - 1), slot = udiv(ic, p.magic_per);
            ok = i < n_g && ((active_mask >> slot) & 1);
            return reinterpret_cast<const uint4 *>(pool_grid_ + static_cast<size_t>(max(lsrc[slot], 0)) * p.GS)[ic - slot * per];
        };
#pragma unroll 1
        for (int i0 = lane; i0 < n_h; i0 += 256) {
            bool k0, k1, k2, k3;
            int a0, a1, a2, a3;
            const uint4 v0 = hdr_piece(i0, k0, a0), v1 = hdr_piece(i0 + 64, k1, a1), v2 = hdr_piece(i0 + 128, k2, a2),
                        v3 = hdr_piece(i0 + 192, k3, a3);
            if (k0) dh[a0] = v0;
            if (k1) dh[a1] = v1;
            if (k2) dh[a2] = v2;
            if (k3) dh[a3] = v3;
        }
#pragma unroll 1
        for (int i0 = lane; i0 < n_g; i0 += 256) {
            bool k0, k1, k2, k3;
            const uint4 v0 = grid_piece(i0, k0), v1 = grid_piece(i0 + 64, k1), v2 = grid_piece(i0 + 128, k2),
                        v3 = grid_piece(i0 + 192, k3);
            if (k0) dg[i0] = v0;
            if (k1) dg[i0 + 64] = v1;
            if (k2) dg[i0 + 128] = v2;
            if (k3) dg[i0 + 192] = v3;
        }
        wave_sync();
        if (active) *reinterpret_cast<uint32_t *>(lds + p.o.hdr + lane * p.HP + H_EPISODE) = episode;  // the environment's own counter
        wave_sync();
    }
    if (MODE != MODE_STEP) {
        for (int s = lane; s < nag; s += 64) {
            const int slot = udiv(s, p.magic_a);
            lds[p.o.oterm + s] = lds[p.o.hdr + slot * p.HP + H_ROWS + 8 * (s - slot * A) + R_TERM];
        }
        wave_sync();
    }

    MG_STAMP(1);
    // ---- act ---------------------------------------------------------------------------------------
    const int my_slot = udiv(lane, p.magic_a);  // environment (within a sub-tile) of this lane's agent
    const int my_agent = lane - my_slot * A;
    MG_EXP(
    unsigned long long ta0 = 0, ta1 = 0, ta2 = 0, tc = 0;
    MG_TICK(ta0);
    )
    int t = 0;
    // multi-step kernel: the next step's action and order words, loaded while this step's views are made
    uint32_t aw_next = 0, ow_next = 0;
    bool words_next = false;
    do {  // one pass unless MULTI
    if (MULTI && t > 0) {
        // the next step's actions and order (p.actions / p.order have moved on to its slice)
        wave_sync();
        const uint8_t *ga = p.actions + static_cast<size_t>(e0) * A;
        const uint8_t *go = p.order ? p.order + static_cast<size_t>(e0) * A : nullptr;
        if (words_next && 4 * lane + 4 <= nag) {
            *reinterpret_cast<uint32_t *>(lds + p.o.act + 4 * lane) = aw_next;
            if (go) *reinterpret_cast<uint32_t *>(lds + p.o.ord + 4 * lane) = ow_next;
        }
#pragma unroll 1
        for (int b = words_next ? min(nag & ~3, 256) + lane : lane; b < nag; b += 64) {  // what the words left over
            lds[p.o.act + b] = ga[b];
            if (go) lds[p.o.ord + b] = go[b];
        }
        if (lane < ET) lsrc[lane] = -1;
        if (CROWD) {  // the images of the last step lay over the cell table
            uint4 *tz = reinterpret_cast<uint4 *>(lds + p.o.tbl);
            const int nz = NE * p.GP / 16;
#pragma unroll 1
            for (int i = lane; i < nz; i += 64) tz[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        wave_sync();
    }
    // CompetativeRedBlueDoor steps (in-order act phase, one lane per environment) and the cell-table variant (its
    // table lies where the grids go) act on the whole tile before the first view is taken; all others take the
    // tile one sub-tile at a time - act, view, images out - so that the image stores of one sub-tile drain while the
    // next one acts.
#ifndef MG_INTERLEAVE_ACT
#define MG_INTERLEAVE_ACT 0
#endif
    constexpr bool ACT_FIRST = COMP || CROWD || !MG_INTERLEAVE_ACT;
    // the multi-step kernel's next actions set out once this step's are no longer needed in registers
    auto prefetch_actions = [&]() {
        if (MULTI && t + 1 < p.T) {
            const size_t ea = static_cast<size_t>(p.E) * A;
            const uint8_t *ga = p.actions + ea + static_cast<size_t>(e0) * A;
            const uint8_t *go = p.order ? p.order + ea + static_cast<size_t>(e0) * A : nullptr;
            words_next = ((reinterpret_cast<uintptr_t>(ga) | reinterpret_cast<uintptr_t>(go)) & 3) == 0;
            if (words_next && 4 * lane + 4 <= nag) {
                aw_next = *reinterpret_cast<const uint32_t *>(ga + 4 * lane);
                if (go) ow_next = *reinterpret_cast<const uint32_t *>(go + 4 * lane);
            }
        }
    };
    // MG_AUTORESET_SAME_STEP: environments [first, first + count) of the tile whose episode ended in this step take
    // their next start state now, before the views are made
    // (`resident`: environments below this slot have their grids in LDS already; the others' are marked for patch())
    auto restart_now = [&](int first, int count, int resident) {
        if (p.autoreset() != MG_AUTORESET_SAME_STEP) return;
        uint64_t m = __ballot(lane >= first && lane < first + count && lds[p.o.flag + 4 * lane + 1] != 0);
        if (m) {
            while (m) {
                const int s = __builtin_ctzll(m);
                m &= m - 1;
                const int env = e0 + s;
                const uint32_t episode = *reinterpret_cast<const uint32_t *>(lds + p.o.hdr + s * p.HP + H_EPISODE) + 1u;
                const int layout = pick_layout(p, env, episode);
                wave_sync();
                reload_env(p, lds, s, env, layout, episode, lane, s < resident ? lds + p.o.grid + s * p.GP : nullptr);
                if (lane == 0 && s >= resident) lsrc[s] = layout;
                if (lane == 0) lds[p.o.flag + 4 * s + 2] = 0;  // the fresh grid takes no patches
                if (lane == 0) lds[p.o.flag + 4 * s + 3] = 0;
                wave_sync();
                if (lane < A) lds[p.o.oterm + s * A + lane] = lds[p.o.hdr + s * p.HP + H_ROWS + 8 * lane + R_TERM];
                if (COMP && lane < A)  // nothing of the hook is left to apply: the agents stand where the layout puts them
                    reinterpret_cast<uint16_t *>(lds + p.o.pxy)[s * A + lane] =
                        *reinterpret_cast<const uint16_t *>(lds + p.o.hdr + s * p.HP + H_ROWS + 8 * lane);
            }
            wave_sync();
        }
    };
    if (MODE == MODE_STEP && ACT_FIRST) {
        if constexpr (COMP) {
            if (lane < net) {
                act_env_small<true>(p, lds, lane, e0 + lane, load_bytes(lds + p.o.act + lane * A, A),
                                    p.order ? load_bytes(lds + p.o.ord + lane * A, A) : 0ull);
            }
        } else {
#pragma unroll 1
            for (int s0 = 0; s0 < net; s0 += NE) {
                const int ns = min(NE, net - s0);
                // (the quad-permute forms serve the small rooms of BlockedUnlockPickup / LockedHallway / RedBlueDoors,
                // whose agents meet all the time; see-through kernels - open Empty rooms - keep to the one general form)
                if (!CROWD && !SEE && A == 2) act_lanes<false, 2>(p, lds, e0, s0, ns, lane, my_slot, my_agent);
                else if (!CROWD && !SEE && A == 4) act_lanes<false, 4>(p, lds, e0, s0, ns, lane, my_slot, my_agent);
                else act_lanes<CROWD>(p, lds, e0, s0, ns, lane, my_slot, my_agent);
            }
        }
        wave_sync();
        MG_EXP(MG_TICK(ta1);)
        MG_STAMP(2);
        prefetch_actions();
        // the first sub-tile's grids, on their way since the launch began, land in LDS now: environments that restart
        // put their new grid there themselves
        if (!MULTI && passes) {
            commit(0);
            wave_sync();
        }
        restart_now(0, net, MULTI ? net : min(NE, net));
        // (diagnostic builds: staging, act, same-step restarts)
        MG_EXP(
        MG_TICK(ta2);
        if (p.dbg_buf && (p.dbg & 16) && threadIdx.x == 0) {
            unsigned long long *d = p.dbg_buf + static_cast<size_t>(blockIdx.x) * 32 + 20;
            d[0] = ta0 - tstart;
            d[1] = ta1 - ta0;
            d[2] = ta2 - ta1;
        }
        )
    }

    MG_STAMP(3);
    // ---- act (unless done above) and observe: NE environments at a time ---------------------------------------
    if (passes) {
        uint8_t *out_image = p.image;
        if constexpr (!MULTI) out_image = late_arg<uint8_t *>(offsetof(KParams, image));  // (see the per-agent scalars)
        for (int s0 = 0; s0 < net; s0 += NE) {
            const int ns = min(NE, net - s0);
            const int np = ns * A;
            const uint64_t sub_mask = (active_mask >> s0) & ((ns == 64) ? ~0ull : ((1ull << ns) - 1ull));
            const bool all_active = sub_mask == ((ns == 64) ? ~0ull : ((1ull << ns) - 1ull));
            // (a reset takes every grid it looks at from the layout pool - patch() - and prefetches none from the state)
            const bool has_next = !MULTI && MODE != MODE_RESET && s0 + NE < net;
            const int gslot0 = MULTI ? s0 : 0;  // where the sub-tile's grids start in LDS
            MG_EXP(
            unsigned long long tk0 = 0, tka = 0, tk1 = 0, tk2 = 0, tk3 = 0;
            MG_TICK(tk0);
            )
            if (MODE == MODE_STEP && !ACT_FIRST) {
                act_lanes<false>(p, lds, e0, s0, ns, lane, my_slot, my_agent);
                if (s0 == 0) prefetch_actions();
                restart_now(s0, ns, MULTI ? net : 0);
            }
            MG_EXP(MG_TICK(tka);)
            // the sub-tile's grids: the first ones were set out when the launch began, the others came in at the
            // end of the pass before; what was restarted in this launch is taken from the pool
            if (!MULTI && s0 == 0 && MODE != MODE_RESET && !(MODE == MODE_STEP && ACT_FIRST)) commit(0);
            if (!MULTI && !(MODE == MODE_STEP && ACT_FIRST && s0 == 0)) patch(s0, gslot0);
            wave_sync();
            if (s0 == 0) MG_STAMP(4);
            const bool mine = lane < np && ((sub_mask >> my_slot) & 1);
            const uint8_t *my_hdr = lds + p.o.hdr + (s0 + my_slot) * p.HP;
            uint16_t *my_grid = reinterpret_cast<uint16_t *>(lds + p.o.grid + (gslot0 + my_slot) * p.GP);
            if (MODE == MODE_STEP) {
                // cells changed by this step, replayed in order (one lane per environment)
                if (lane < ns && ((sub_mask >> lane) & 1)) {
                    const uint32_t *log = reinterpret_cast<const uint32_t *>(lds + p.o.log) + (s0 + lane) * A;
                    uint16_t *g = reinterpret_cast<uint16_t *>(lds + p.o.grid + (gslot0 + lane) * p.GP);
                    const int n = lds[p.o.flag + 4 * (s0 + lane) + 2];
                    for (int j = 0; j < n; ++j) {
                        const uint32_t e = log[j];
                        g[e & 0xFFFF] = static_cast<uint16_t>(e >> 16);
                    }
                }
                wave_sync();
            }
            // this lane's agent as the act phase left it, read once for the rest of the pass
            uint2 my_row = make_uint2(0u, 0u);
            bool my_gone = true;  // terminated as far as gen_obs is concerned
            if (mine) {
                my_row = *reinterpret_cast<const uint2 *>(my_hdr + H_ROWS + 8 * my_agent);
                my_gone = lds[p.o.oterm + (s0 + my_slot) * A + my_agent] != 0;
            }
            // Front cells for the NEXT launch, taken while no agent is drawn into the grid.  A CompetativeRedBlueDoor
            // step takes them after the views instead: its hook moves agents and clears cells, and none of that may
            // show in this step's observation.
            constexpr bool comp_step = COMP;
            auto refresh_front = [&](uint32_t row_x) {
                if (mine) {
                    const uint32_t fi = front_index(p, row_x);
                    reinterpret_cast<uint16_t *>(lds + p.o.hdr + (s0 + my_slot) * p.HP + p.HFC)[my_agent] =
                        fi == FC_NONE ? static_cast<uint16_t>(0) : my_grid[fi];
                }
                // (a step of CompetativeRedBlueDoor environments is always a COMP launch)
                const bool comp_env = COMP || (MODE != MODE_STEP && p.comp());
                if ((p.kind() == MG_KIND_REDBLUEDOORS || comp_env) && lane < ns && ((sub_mask >> lane) & 1)) {
                    uint8_t *h = lds + p.o.hdr + (s0 + lane) * p.HP;
                    uint32_t meta = *reinterpret_cast<const uint32_t *>(h + H_META);
                    if (comp_env)  // the doors of competative_red_blue_door.py:201-211 never move
                        meta = static_cast<uint32_t>(p.W() / 4) | (static_cast<uint32_t>(p.H() - 2) << 8) |
                               (static_cast<uint32_t>(p.W() / 4 + p.W() / 2 - 1) << 16) | (1u << 24);
                    const uint16_t *g = reinterpret_cast<const uint16_t *>(lds + p.o.grid + (gslot0 + lane) * p.GP);
                    const uint32_t last = static_cast<uint32_t>(p.W() * p.H() - 1);
                    reinterpret_cast<uint16_t *>(h + p.HFC)[A] = g[min((meta & 0xFF) * p.H() + ((meta >> 8) & 0xFF), last)];
                    reinterpret_cast<uint16_t *>(h + p.HFC)[A + 1] = g[min(((meta >> 16) & 0xFF) * p.H() + (meta >> 24), last)];
                }
            };
            if (MODE != MODE_OBS && !comp_step) {
                refresh_front(my_row.x);
                wave_sync();
            }
            if (s0 == 0) MG_STAMP(5);
            // Draw every live agent into the grid copy; where several share a cell the highest index
            // shows (multigrid/utils/obs.py:157-165: index order, later agents overwrite; single-agent
            // environments draw nothing, obs.py:166-167).
            bool draw = false;  // this lane's agent is drawn at cell `ci`, which held `under`
            uint32_t ci = 0, under = 0;
            MG_EXP(if (!(p.dbg & 8)))
            if (want_img && A > 1) {
                // LDS writes of one wavefront land in program order, so drawing the agents index by index
                // leaves the highest one on a shared cell without looking at the other agents at all
                uint32_t val = 0;
                if (mine && !my_gone) {
                    const uint2 r = my_row;
                    const unsigned x = r.x & 0xFF, y = (r.x >> 8) & 0xFF;
                    draw = x < static_cast<unsigned>(p.W()) && y < static_cast<unsigned>(p.H());
                    ci = x * p.H() + y;
                    val = T_AGENT | (((r.x >> 16) & 0xFF) << 4) | ((r.y >> 24) << 8);
                }
                // the multi-step kernel keeps its grids: every lane notes what lies under its agent before the
                // first agent is drawn, and puts it back once the views have been taken
                if ((MULTI || comp_step) && draw) under = my_grid[ci];
#pragma unroll 1
                for (int k = 0; k < A; ++k) {
                    if (draw && my_agent == k) my_grid[ci] = static_cast<uint16_t>(val);
                    asm volatile("" ::: "memory");  // one store per round, in this order
                }
            }
            wave_sync();

            if (s0 == 0) MG_STAMP(6);
            constexpr int NPX = (IMG + 3) / 4;
            uint32_t px[NPX];
#pragma unroll
            for (int k = 0; k < NPX; ++k) px[k] = 0u;
            MG_EXP(if (!(p.dbg & 1)))
            if (want_img && mine)
                observe_agent<V, SEE, WALLED>(p, my_row, lds + p.o.grid + (gslot0 + my_slot) * p.GP, px);
            wave_sync();
            if (MULTI || comp_step) {
                if (draw) my_grid[ci] = static_cast<uint16_t>(under);
                wave_sync();
            }
            if (comp_step) {
                // what the hook did after gen_obs: cells cleared under eliminated agents, agents moved away
                if (lane < ns && ((sub_mask >> lane) & 1)) {
                    const uint32_t *log2 = reinterpret_cast<const uint32_t *>(lds + p.o.log2) + (s0 + lane) * A;
                    uint16_t *g = reinterpret_cast<uint16_t *>(lds + p.o.grid + (gslot0 + lane) * p.GP);
                    const int n2 = lds[p.o.flag + 4 * (s0 + lane) + 3];
                    for (int j = 0; j < n2; ++j) g[log2[j] & 0xFFFF] = static_cast<uint16_t>(log2[j] >> 16);
                }
                uint32_t row_x = my_row.x;
                if (mine) {
                    const uint32_t xy = reinterpret_cast<const uint16_t *>(lds + p.o.pxy)[(s0 + my_slot) * A + my_agent];
                    row_x = (row_x & 0xFFFF0000u) | xy;
                    *reinterpret_cast<uint16_t *>(lds + p.o.hdr + (s0 + my_slot) * p.HP + H_ROWS + 8 * my_agent) =
                        static_cast<uint16_t>(xy);
                }
                wave_sync();
                refresh_front(row_x);
                wave_sync();
            }
            // The next sub-tile's grids set out now, travel while this sub-tile's images are stored and
            // replace this sub-tile's grids in LDS at the end of the pass; they occupy registers only
            // while the observation's own are free.
            MG_EXP(MG_TICK(tk1);)
            if (has_next) issue(s0 + NE);
            __builtin_amdgcn_sched_barrier(0);
            MG_EXP(MG_TICK(tk2);)

            if (s0 == 0) MG_STAMP(7);
            // Images leave through a small LDS buffer, STG agents at a time: their lanes lay the
            // byte streams down back to back, then the whole wavefront stores the block as 16-byte pieces.
            // The stream sits `shift` bytes into the buffer so that LDS and HBM agree modulo 16.
            const size_t tile_agent0 = static_cast<size_t>(e0 + s0) * A;  // first agent of the sub-tile
            const size_t tile_base = tile_agent0 * IMG;  // byte offset of the sub-tile's images
            MG_EXP(if (!(p.dbg & 2)))
            if (want_img && all_active && np % STG == 0 &&
                (out_image == nullptr || ((reinterpret_cast<uintptr_t>(out_image) + tile_base) & 15) == 0)) {
                // whole blocks, 16-byte aligned in HBM (STG * IMG is a multiple of 16)
                constexpr int CH = STG * IMG / 16;  // 16-byte pieces per block
                constexpr int MAXIT = (CH + 63) / 64;
                // An image is 4 M + 3 bytes, so four consecutive agents fill 4 M + 3 words exactly and agent j
                // of such a quad starts j bytes before a word boundary.  Each lane shifts its stream by j bytes
                // and owns the words (M + 1) j .. (M + 1) j + M - 1; the word after those straddles two agents
                // and is put together with the first bytes of the next lane.  All LDS writes are then
                // word-aligned with an odd word stride between lanes.
                constexpr int M = IMG / 4;
                const uint32_t px_first = px[0];
                static_assert(IMG % 4 == 3 && NPX == M + 1, "3 V^2 with odd V is 3 mod 4");
                const int j = lane & 3;
#pragma unroll
                for (int k = 0; k < M; ++k) px[k] = __builtin_amdgcn_alignbyte(px[k + 1], px[k], j);
                {
                    const uint32_t next0 = static_cast<uint32_t>(
                        __builtin_amdgcn_update_dpp(0, static_cast<int>(px_first), 0xF9, 0xF, 0xF, false));  // quad_perm [1,2,3,3]
                    px[M] = __builtin_amdgcn_alignbyte(next0 >> 8, (px[M] & 0xFFFFFFu) | (next0 << 24), j);
                }
                uint32_t *wdst = reinterpret_cast<uint32_t *>(lds + p.o.stage) + ((lane & (STG - 1)) >> 2) * IMG + (M + 1) * j;
                const uint4 *src16 = reinterpret_cast<const uint4 *>(lds + p.o.stage);
                uint4 *dst16 = reinterpret_cast<uint4 *>(out_image + tile_base) + lane;
#pragma unroll
                for (int r = 0; r < 64 / STG; ++r) {
                    if (r * STG < np) {
                        if (lane / STG == r) {
#pragma unroll
                            for (int k = 0; k < M; ++k) wdst[k] = px[k];
                            if (j < 3) wdst[M] = px[M];
                        }
                        wave_sync();
                        if (out_image) {
                            constexpr int BATCH = MG_COPY_BATCH;  // LDS reads of a batch first, then its stores
#pragma unroll
                            for (int b0 = 0; b0 < MAXIT; b0 += BATCH) {
                                uint4 ob[BATCH];
#pragma unroll
                                for (int it = b0; it < min(b0 + BATCH, MAXIT); ++it) ob[it - b0] = src16[min(it * 64 + lane, CH - 1)];
#pragma unroll
                                for (int it = b0; it < min(b0 + BATCH, MAXIT); ++it)
                                    if (it * 64 + 64 <= CH || lane < CH - it * 64) store_out16(dst16 + r * CH + it * 64, ob[it - b0], p.stream_out());
                            }
                        }
                        if (OH && p.onehot)
                            emit_one_hot<V>(p, lds, lds + p.o.stage, STG * V * V,
                                            p.onehot + (tile_agent0 + r * STG) * (OH_DIM * V * V), lane, true, ~0ull,
                                            r * STG);
                        wave_sync();
                    }
                }
            } else if (want_img) {
#pragma unroll 1
                for (int l0 = 0; l0 < np; l0 += STG) {
                    const int nl = min(STG, np - l0);
                    const size_t out_base = tile_base + static_cast<size_t>(l0) * IMG;  // byte offset of the block
                    const int shift = out_image ? static_cast<int>((reinterpret_cast<uintptr_t>(out_image) + out_base) & 15) : 0;
                    if (mine && lane >= l0 && lane < l0 + nl) {
                        uint8_t *dst = lds + p.o.stage + shift + (lane - l0) * IMG;
#pragma unroll
                        for (int k = 0; k + 4 <= NPX; k += 4) __builtin_memcpy(dst + 4 * k, &px[k], 16);
#pragma unroll
                        for (int k = NPX / 4 * 4; k < NPX; ++k) __builtin_memcpy(dst + 4 * k, &px[k], min(4, IMG - 4 * k));
                    }
                    wave_sync();
                    uint8_t *gdst = out_image + out_base - shift;
                    const int total = shift + nl * IMG;
                    const int nchunk = out_image ? (total + 15) / 16 : 0;
                    if (OH && p.onehot)
                        emit_one_hot<V>(p, lds, lds + p.o.stage + shift, nl * V * V,
                                        p.onehot + (tile_agent0 + l0) * (OH_DIM * V * V), lane, false, sub_mask, l0);
#pragma unroll 1
                    for (int c = lane; c < nchunk; c += 64) {
                        const int b0 = 16 * c;
                        if (all_active && b0 >= shift && b0 + 16 <= total) {
                            *reinterpret_cast<uint4 *>(gdst + b0) = *reinterpret_cast<const uint4 *>(lds + p.o.stage + b0);
                        } else {
                            for (int bb = b0; bb < b0 + 16; ++bb) {
                                if (bb < shift || bb >= total) continue;
                                const int slot = (l0 + (bb - shift) / IMG) / A;
                                if ((sub_mask >> slot) & 1) gdst[bb] = lds[p.o.stage + bb];
                            }
                        }
                    }
                    wave_sync();
                }
            }
            if (has_next) {
                commit(s0 + NE);
                wave_sync();
            }
            if (s0 == 0) MG_STAMP(8);
            // (diagnostic builds: compute, commit of the prefetched grids, copy-out, passes, act of the sub-tile)
            MG_EXP(
            MG_TICK(tk3);
            if (p.dbg_buf && (p.dbg & 16) && threadIdx.x == 0) {
                unsigned long long *d = p.dbg_buf + static_cast<size_t>(blockIdx.x) * 32 + 16;
                d[0] += tk1 - tka;
                d[1] += tk2 - tk1;
                d[2] += tk3 - tk2;
                d[3] += 1;
                d[4] += tka - tk0;
            }
            )
        }
    }

    // ---- per-agent scalars: one coalesced element per lane ---------------------------------------------
    // (every pointer is fetched, and every value read from LDS, before the first store goes out: one wait for each
    // kind of memory instead of one per output)
    uint8_t *out_direction = p.direction, *out_terminated = p.terminated, *out_truncated = p.truncated;
    float *out_reward = p.reward;
    uint2 *out_rows = nullptr;      // optional outputs of single-step launches (callers that mirror one environment
    double *out_reward64 = nullptr;  // on the host)
    uint8_t *out_info = COMP ? p.info : nullptr;
    if constexpr (!MULTI) {  // (the multi-step kernel moves its pointers on from step to step)
        out_direction = late_arg<uint8_t *>(offsetof(KParams, direction));
        out_reward = late_arg<float *>(offsetof(KParams, reward));
        out_terminated = late_arg<uint8_t *>(offsetof(KParams, terminated));
        out_truncated = late_arg<uint8_t *>(offsetof(KParams, truncated));
        out_rows = late_arg<uint2 *>(offsetof(KParams, rows_out));
        out_reward64 = late_arg<double *>(offsetof(KParams, reward64));
        if (COMP) out_info = late_arg<uint8_t *>(offsetof(KParams, info));
    }
    for (int s = lane; s < nag; s += 64) {
        const int slot = udiv(s, p.magic_a);
        if (!((active_mask >> slot) & 1)) continue;
        const size_t o = static_cast<size_t>(e0) * A + s;
        const uint2 row = *reinterpret_cast<const uint2 *>(lds + p.o.hdr + slot * p.HP + H_ROWS + 8 * (s - slot * A));
        uint32_t tout = 0, trunc = 0, m = 0;
        double R = 0.0;
        if (MODE == MODE_STEP) {
            tout = lds[p.o.tout + s];
            trunc = lds[p.o.flag + 4 * slot];
            if (COMP) {
                R = reinterpret_cast<const double *>(lds + p.o.racc)[s];
            } else {
                m = lds[p.o.rewm + s];
                R = reinterpret_cast<const double *>(lds + p.o.R)[slot];
            }
        }
        asm volatile("" : "+v"(tout), "+v"(trunc), "+v"(m));  // all of the above before any of the below
        if (out_direction) out_direction[o] = static_cast<uint8_t>(row.x >> 16);
        if (out_rows) out_rows[o] = row;
        if (MODE == MODE_STEP) {
            const double r = COMP ? R : (m == 0 ? 0.0 : reward_times(R, static_cast<int>(m)));
            if (out_reward) out_reward[o] = static_cast<float>(r);
            if (out_reward64) out_reward64[o] = r;
            if (out_terminated) out_terminated[o] = static_cast<uint8_t>(tout & 1u);
            if (COMP && out_info) out_info[o] = static_cast<uint8_t>(((tout >> 1) & 7u) | (tout & 0xF0u));
            if (out_truncated) out_truncated[o] = static_cast<uint8_t>(trunc);
        }
    }

    MG_STAMP(9);
    MG_EXP(MG_TICK(tc);)
    if (MODE == MODE_STEP && p.autoreset() == MG_AUTORESET_NEXT_STEP) {
        uint64_t m = __ballot(lane < net && lds[p.o.flag + 4 * lane + 1] != 0);
        if (m) {
            wave_sync();
            while (m) {
                const int s = __builtin_ctzll(m);
                m &= m - 1;
                const int env = e0 + s;
                const uint32_t episode = *reinterpret_cast<const uint32_t *>(lds + p.o.hdr + s * p.HP + H_EPISODE) + 1u;
                wave_sync();
                const int layout = pick_layout(p, env, episode);
                reload_env(p, lds, s, env, layout, episode, lane, MULTI ? lds + p.o.grid + s * p.GP : nullptr);  // (the grids kept in LDS follow)
            }
            wave_sync();
        }
    }
    if constexpr (MULTI) {  // on to the next step's slices of the inputs and outputs
        const size_t ea = static_cast<size_t>(p.E) * A;
        p.actions += ea;
        if (p.order) p.order += ea;
        if (p.image) p.image += ea * IMG;
        if (p.direction) p.direction += ea;
        if (p.reward) p.reward += ea;
        if (p.terminated) p.terminated += ea;
        if (p.truncated) p.truncated += ea;
        if (COMP && p.info) p.info += ea;
    }
    } while (MULTI && ++t < p.T);  // steps of this launch
    if (MODE != MODE_OBS) {
        // header blocks back to HBM (reset mode: only the environments that were reset)
        uint4 *dh = reinterpret_cast<uint4 *>(late_arg<uint8_t *>(offsetof(KParams, hdr)) + static_cast<size_t>(e0) * p.HS);
        const uint4 *sh = reinterpret_cast<const uint4 *>(lds + p.o.hdr);
        const int per_h = p.HS / 16, hpad = (p.HP - p.HS) >> 4;
        for (int i = lane; i < net * per_h; i += 64) {
            const int slot = udiv(i, p.magic_ph);
            if ((active_mask >> slot) & 1) dh[i] = sh[i + hpad * slot];
        }
    }
    MG_STAMP(10);
    MG_EXP(
    {
        unsigned long long tend = 0;
        MG_TICK(tend);
        if (p.dbg_buf && (p.dbg & 16) && threadIdx.x == 0) {
            p.dbg_buf[static_cast<size_t>(blockIdx.x) * 32 + 23] = tend - tstart;
            p.dbg_buf[static_cast<size_t>(blockIdx.x) * 32 + 30] = tend - tc;
            p.dbg_buf[static_cast<size_t>(blockIdx.x) * 32 + 31] = tc - tstart;
        }
    }
    )
}

// ---- reference-layout <-> device-layout conversion (debug / parity path) -------------------------------

__global__ void mg_set_state_kernel(KParams p, const uint8_t *grid, const uint8_t *agents, const uint8_t *meta,
                                    const int32_t *steps, const uint8_t *flags) {
    const int cells = p.W() * p.H();
    const size_t n = static_cast<size_t>(p.E) * cells;
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const size_t e = i / cells;
        const int c = static_cast<int>(i - e * cells);
        if (grid) {
            const uint8_t *s = grid + i * 3;
            reinterpret_cast<uint16_t *>(p.grid + e * p.GS)[c] = static_cast<uint16_t>(pack_cell(s[0], s[1], s[2]));
        }
        if (c == 0) {
            uint8_t *h = p.hdr + e * p.HS;
            if (steps) *reinterpret_cast<int32_t *>(h + H_STEP) = steps[e];
            if (flags) h[H_FLAGS] = static_cast<uint8_t>(flags[e] & ~FLAG_OVER);
            if (meta)
                for (int k = 0; k < 4; ++k) h[H_META + k] = meta[e * 4 + k];
            if (agents)
                for (int k = 0; k < p.A() * 8; ++k) h[H_ROWS + k] = agents[e * p.A() * 8 + k];
        }
    }
}

// Fields the packed formats cannot hold (type, colour, state above 15; a heading above 3) and agents that stand
// outside the grid (which mg_upload_layouts refuses as well): counted, so that mg_set_state can refuse the state
// instead of truncating it.  Cells and agent rows are walked in loops of their own: there may be more rows than cells.
__global__ void mg_validate_state_kernel(KParams p, const uint8_t *grid, const uint8_t *agents, int *bad) {
    const size_t cells = static_cast<size_t>(p.E) * p.W() * p.H(), rows = static_cast<size_t>(p.E) * p.A();
    const size_t first = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x, stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    int n = 0;
    if (grid)
        for (size_t i = first; i < cells; i += stride) n += (grid[3 * i] > 15) | (grid[3 * i + 1] > 15) | (grid[3 * i + 2] > 15);
    if (agents)
        for (size_t i = first; i < rows; i += stride) {
            const uint8_t *r = agents + 8 * i;
            n += (r[R_X] >= p.W()) | (r[R_Y] >= p.H()) | (r[R_DIR] > 3) | (r[R_CT] > 15) | (r[R_CC] > 15) | (r[R_CS] > 15) |
                 (r[R_COLOR] > 15);
        }
    if (n) atomicAdd(bad, n);
}

// A state in the reference's layouts -> the packed formats, for mg_gen_obs_of_state(): every byte of the
// environments' header blocks and grids is written (whatever an earlier call left there is gone), fields are
// taken modulo 16.
__global__ void mg_pack_state_kernel(KParams p, const uint8_t *grid, const uint8_t *agents) {
    const int cells = p.W() * p.H(), slots = p.GS / 2;
    const size_t n = static_cast<size_t>(p.E) * slots;
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const size_t e = i / slots;
        const int c = static_cast<int>(i - e * slots);
        uint32_t w = 0;
        if (c < cells) {
            const uint8_t *s = grid + (e * cells + c) * 3;
            w = pack_cell(s[0] & 15u, s[1] & 15u, s[2] & 15u);
        }
        reinterpret_cast<uint16_t *>(p.grid + e * p.GS)[c] = static_cast<uint16_t>(w);
        if (c == 0) {
            uint8_t *h = p.hdr + e * p.HS;
            for (int k = 0; k < p.HS; ++k) h[k] = 0;
            for (int k = 0; k < p.A() * 8; ++k) {
                const int f = k & 7;
                h[H_ROWS + k] = agents[e * p.A() * 8 + k] & (f == R_DIR ? 3u : (f >= R_CT ? 15u : 255u));
            }
        }
    }
}

// Recompute the front cells kept in the header blocks from the grid (after mg_set_state).
__global__ void mg_refresh_front_kernel(KParams p) {
    const int n = p.E * (p.A() + 2);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int e = i / (p.A() + 2), k = i - e * (p.A() + 2);
        uint8_t *h = p.hdr + static_cast<size_t>(e) * p.HS;
        const uint16_t *g = reinterpret_cast<const uint16_t *>(p.grid + static_cast<size_t>(e) * p.GS);
        uint32_t ci = FC_NONE;
        if (k < p.A()) {
            ci = front_index(p, *reinterpret_cast<const uint32_t *>(h + H_ROWS + 8 * k));
        } else if (p.kind() == MG_KIND_REDBLUEDOORS) {
            const int mx = h[H_META + 2 * (k - p.A())], my = h[H_META + 2 * (k - p.A()) + 1];
            ci = min(static_cast<uint32_t>(mx * p.H() + my), static_cast<uint32_t>(p.W() * p.H() - 1));
        } else if (p.comp()) {  // the doors of competative_red_blue_door.py:201-211
            ci = k == p.A() ? (p.W() / 4) * p.H() + p.H() - 2 : (p.W() / 4 + p.W() / 2 - 1) * p.H() + 1;
        }
        reinterpret_cast<uint16_t *>(h + p.HFC)[k] = ci == FC_NONE ? static_cast<uint16_t>(0) : g[ci];
    }
}

__global__ void mg_get_state_kernel(KParams p, uint8_t *grid, uint8_t *agents, uint8_t *meta, int32_t *steps,
                                    uint8_t *flags) {
    const int cells = grid ? p.W() * p.H() : 1;  // without the grids: one thread per environment
    const size_t n = static_cast<size_t>(p.E) * cells;
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const size_t e = i / cells;
        const int c = static_cast<int>(i - e * cells);
        if (grid) {
            const uint32_t w = reinterpret_cast<const uint16_t *>(p.grid + e * p.GS)[c];
            uint8_t *d = grid + i * 3;
            d[0] = static_cast<uint8_t>(w & 15);
            d[1] = static_cast<uint8_t>((w >> 8) & 127);
            d[2] = static_cast<uint8_t>((w >> 4) & 15);
        }
        if (c == 0) {
            const uint8_t *h = p.hdr + e * p.HS;
            if (steps) steps[e] = *reinterpret_cast<const int32_t *>(h + H_STEP);
            if (flags) flags[e] = h[H_FLAGS] & ~FLAG_OVER;
            if (meta)
                for (int k = 0; k < 4; ++k) meta[e * 4 + k] = h[H_META + k];
            if (agents)
                for (int k = 0; k < p.A() * 8; ++k) agents[e * p.A() * 8 + k] = h[H_ROWS + k];
        }
    }
}

// PettingZooWrapper.agents (multigrid/pettingzoo/__init__.py:48-52) and MultiGridEnv.is_done
// (multigrid/base.py:587-596) of every environment, one thread per environment, straight from the header blocks.
__global__ void mg_agents_mask_kernel(KParams p, uint8_t *mask, uint8_t *done) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.E) return;
    const uint8_t *h = p.hdr + static_cast<size_t>(e) * p.HS;
    const int a = p.A();
    bool all_term = true;
    for (int k = 0; k < a; ++k) all_term = all_term && h[H_ROWS + 8 * k + 3] != 0;
    const bool over = *reinterpret_cast<const int32_t *>(h + H_STEP) >= p.max_steps || all_term;
    if (done) done[e] = over ? 1 : 0;
    if (mask)
        for (int k = 0; k < a; ++k) mask[static_cast<size_t>(e) * a + k] = (!over && h[H_ROWS + 8 * k + 3] == 0) ? 1 : 0;
}

__global__ void mg_stats_kernel(const int64_t *shards, int64_t *out, int64_t launches) {
    const int k = threadIdx.x;
    if (k >= MG_NUM_STATS) return;
    int64_t s = 0;
    for (int i = 0; i < 64; ++i) s += shards[i * MG_NUM_STATS + k];
    if (k == 5) s = launches;
    out[k] = s;
}

// Per-environment episode statistics of a vector of environments, one thread per environment: what
// gym.wrappers.RecordEpisodeStatistics keeps around every environment of the reference's SyncVectorEnv
// (multigrid/scripts/train_ppo_cleanrl.py:173,366) and what its training loop reads back as
// info["final_info"][i]["episode"]["r" / "l"] (:418-427, :522-527).  An episode is over when every agent is
// terminated or the environment is truncated; the step's outputs come from a same-step auto-reset launch, so the
// counters start again at once.
__global__ void mg_episode_stats_kernel(int E, int A, const float *reward, const uint8_t *terminated,
                                        const uint8_t *truncated, double *ret, int64_t *len, uint8_t *done,
                                        double *ep_ret, int64_t *ep_len) {
    // A block owns 256 / A consecutive environments.  First one thread per environment reads its A flag bytes
    // and decides whether the episode is over; then one thread per (environment, agent) element updates the
    // per-agent arrays, so that consecutive lanes touch consecutive elements.
    __shared__ uint8_t over_s[256];
    const int epb = 256 / A;
    const int e0 = blockIdx.x * epb;
    const int t = threadIdx.x;
    if (t < epb && e0 + t < E) {
        const size_t at = static_cast<size_t>(e0 + t) * A;
        bool all_term = true;
        for (int i = 0; i < A; ++i) all_term = all_term && terminated[at + i] != 0;
        over_s[t] = (all_term || truncated[at] != 0) ? 1 : 0;
    }
    __syncthreads();
    const int le = t / A, i = t - le * A;
    if (le >= epb || e0 + le >= E) return;
    const int e = e0 + le;
    const bool over = over_s[le] != 0;
    const size_t idx = static_cast<size_t>(e) * A + i;
    const double r = ret[idx] + static_cast<double>(reward[idx]);
    ep_ret[idx] = over ? r : 0.0;
    ret[idx] = over ? 0.0 : r;
    if (i == 0) {
        const int64_t n = len[e] + 1;
        ep_len[e] = over ? n : 0;
        len[e] = over ? 0 : n;
        done[e] = over ? 1 : 0;
    }
}

// ---- start states generated on the device ----------------------------------------------------------------
//
// The `_gen_grid` procedures of the four environment classes (multigrid/envs/empty.py:152-171,
// redbluedoors.py:141-168, blockedunlockpickup.py:136-158, locked_hallway.py:146-184) with their helpers
// (Grid.wall_rect grid.py:168-186, MultiGridEnv.place_obj / place_agent base.py:661-749, RoomGrid
// roomgrid.py:197-391), one thread per layout, written straight into the pool formats.  The draws come
// from a counter generator (splitmix64 keyed by seed and layout index) instead of numpy's PCG64, so the
// layouts follow the reference's distribution without repeating its sequences; a rejection loop that
// does not end within LAYOUT_MAX_TRIES draws (a room too full for its agents) reports failure.

constexpr int LAYOUT_MAX_TRIES = 20000;

struct LayoutGen {
    const KParams &p;
    const MgLayoutSpec &ls;
    uint16_t *g;   // the layout's grid in the pool
    uint8_t *rows; // its agent rows (inside the header block)
    uint64_t state;
    bool failed;

    __device__ uint32_t next() {
        state += 0x9E3779B97F4A7C15ull;
        return static_cast<uint32_t>(mix64(state) >> 32);
    }
    // RandomMixin._rand_int(lo, hi): uniform on lo .. hi - 1, multigrid/utils/random.py:22-31
    __device__ int rand_int(int lo, int hi) { return lo + static_cast<int>(__umulhi(next(), static_cast<uint32_t>(hi - lo))); }
    __device__ uint32_t cell(int x, int y) const { return g[x * p.H() + y]; }
    __device__ void set(int x, int y, uint32_t t, uint32_t c, uint32_t st) { g[x * p.H() + y] = static_cast<uint16_t>(pack_cell(t, c, st)); }
    __device__ void wall_rect(int x, int y, int w, int h) {
        for (int i = 0; i < w; ++i) { set(x + i, y, T_WALL, 5, 0); set(x + i, y + h - 1, T_WALL, 5, 0); }
        for (int j = 0; j < h; ++j) { set(x, y + j, T_WALL, 5, 0); set(x + w - 1, y + j, T_WALL, 5, 0); }
    }
    __device__ bool agent_at(int x, int y) const {
        bool hit = false;
        for (int k = 0; k < p.A(); ++k) hit |= rows[8 * k + R_X] == x && rows[8 * k + R_Y] == y;
        return hit;
    }
    // MultiGridEnv.place_obj, base.py:661-728; `near` is reject_next_to of roomgrid.py:44-49.  An agent
    // that has not been placed yet sits at (255, 255) or, in room grids, at the default position.
    __device__ void place(int tx, int ty, int sw, int sh, bool near, int &ox, int &oy) {
        tx = max(tx, 0);
        ty = max(ty, 0);
        const int hx = min(tx + sw, p.W()), hy = min(ty + sh, p.H());
        for (int tries = 0; tries < LAYOUT_MAX_TRIES; ++tries) {
            const int x = rand_int(tx, hx), y = rand_int(ty, hy);
            if ((cell(x, y) & 15) != T_EMPTY) continue;
            if (agent_at(x, y)) continue;
            if (near) {
                bool close = false;
                for (int k = 0; k < p.A(); ++k) {
                    const int dx = x - rows[8 * k + R_X], dy = y - rows[8 * k + R_Y];
                    close |= dx * dx + dy * dy <= 1;
                }
                if (close) continue;
            }
            ox = x;
            oy = y;
            return;
        }
        failed = true;
        ox = tx;
        oy = ty;
    }
    // MultiGridEnv.place_agent, base.py:738-749
    __device__ void place_agent(int k, int tx, int ty, int sw, int sh) {
        rows[8 * k + R_X] = 255;
        rows[8 * k + R_Y] = 255;
        int x, y;
        place(tx, ty, sw, sh, false, x, y);
        rows[8 * k + R_X] = static_cast<uint8_t>(x);
        rows[8 * k + R_Y] = static_cast<uint8_t>(y);
        rows[8 * k + R_DIR] = static_cast<uint8_t>(rand_int(0, 4));
    }
    // RoomGrid._gen_grid, roomgrid.py:197-230: the rooms' walls, every agent at the default position
    __device__ void rooms(int rs, int nrows, int ncols) {
        for (int r = 0; r < nrows; ++r)
            for (int c = 0; c < ncols; ++c) wall_rect(c * (rs - 1), r * (rs - 1), rs, rs);
        for (int k = 0; k < p.A(); ++k) {
            rows[8 * k + R_X] = static_cast<uint8_t>((ncols / 2) * (rs - 1) + rs / 2);
            rows[8 * k + R_Y] = static_cast<uint8_t>((nrows / 2) * (rs - 1) + rs / 2);
            rows[8 * k + R_DIR] = 0;
        }
    }
    // RandomMixin._rand_perm, random.py:72-80 (Fisher-Yates)
    __device__ void shuffle(uint8_t *v, int n) {
        for (int i = n - 1; i > 0; --i) {
            const int j = rand_int(0, i + 1);
            const uint8_t t = v[i];
            v[i] = v[j];
            v[j] = t;
        }
    }
};

__global__ void mg_generate_layouts_kernel(KParams p, MgLayoutSpec ls, uint8_t *pool_hdr, uint8_t *pool_grid, int n,
                                           uint64_t seed, int *failures) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n) return;
    const int W = p.W(), H = p.H(), A = p.A();
    uint8_t *hd = pool_hdr + static_cast<size_t>(l) * p.HS;
    LayoutGen G{p, ls, reinterpret_cast<uint16_t *>(pool_grid + static_cast<size_t>(l) * p.GS), hd + H_ROWS,
                mix64(seed ^ mix64(static_cast<uint64_t>(l) + 0x632BE59BD9B4E019ull)), false};
    for (int i = 0; i < p.HS; ++i) hd[i] = 0;
    for (int i = 0; i < p.GS / 2; ++i) G.g[i] = 0;
    for (int i = 0; i < W * H; ++i) G.g[i] = CELL_EMPTY;  // Grid.__init__, grid.py:53-54
    for (int k = 0; k < A; ++k) {
        uint8_t *r = G.rows + 8 * k;
        r[R_X] = r[R_Y] = 255;  // not placed yet (Agent.reset, agent.py:172-173)
        r[R_CT] = T_EMPTY;       // carrying nothing
        r[R_COLOR] = ls.agent_color[k];
    }
    uint32_t meta = 0;
    const int kind = p.kind();
    if (kind == MG_KIND_EMPTY) {  // empty.py:152-171
        G.wall_rect(0, 0, W, H);
        G.set(W - 2, H - 2, T_GOAL, 1, 0);
        for (int k = 0; k < A; ++k) {
            if (ls.random_start) {
                G.place_agent(k, 0, 0, W, H);
            } else {
                G.rows[8 * k + R_X] = static_cast<uint8_t>(ls.start_x);
                G.rows[8 * k + R_Y] = static_cast<uint8_t>(ls.start_y);
                G.rows[8 * k + R_DIR] = static_cast<uint8_t>(ls.start_dir);
            }
        }
    } else if (kind == MG_KIND_REDBLUEDOORS) {  // redbluedoors.py:141-168
        const int rx0 = W / 4, rw = W / 2;
        G.wall_rect(0, 0, W, H);
        G.wall_rect(rx0, 0, rw, H);
        for (int k = 0; k < A; ++k) G.place_agent(k, rx0, 0, rw, H);
        const int ry = G.rand_int(1, H - 1);
        G.set(rx0, ry, T_DOOR, 0, S_LOCKED);
        const int bx = rx0 + rw - 1, by = G.rand_int(1, H - 1);
        G.set(bx, by, T_DOOR, 2, S_LOCKED);
        meta = static_cast<uint32_t>(rx0) | (ry << 8) | (bx << 16) | (by << 24);
    } else if (kind == MG_KIND_BLOCKEDUNLOCKPICKUP) {  // blockedunlockpickup.py:136-158
        const int rs = ls.size;
        G.rooms(rs, 1, 2);
        int x, y;
        const int box_color = G.rand_int(0, 6);
        G.place(rs - 1, 0, rs, rs, true, x, y);  // the box, in the right room (roomgrid.py:232-270)
        G.set(x, y, T_BOX, box_color, 0);
        const int door_color = G.rand_int(0, 6);
        const int dx = rs - 1, dy = G.rand_int(1, rs - 1);  // Room.set_door_pos, roomgrid.py:86-127
        G.set(dx, dy, T_DOOR, door_color, S_LOCKED);
        G.set(dx - 1, dy, T_BALL, G.rand_int(0, 6), 0);     // the ball in front of the door (:149)
        G.place(0, 0, rs, rs, true, x, y);                  // the door's key, in the left room
        G.set(x, y, T_KEY, door_color, 0);
        for (int k = 0; k < A; ++k) {  // RoomGrid.place_agent, roomgrid.py:363-391
            for (int tries = 0; tries < LAYOUT_MAX_TRIES; ++tries) {
                G.place_agent(k, 0, 0, rs, rs);
                const uint8_t *r = G.rows + 8 * k;
                const uint32_t t = G.cell(r[R_X] + dir_dx(r[R_DIR]), r[R_Y] + dir_dy(r[R_DIR])) & 15;
                if (t == T_EMPTY || t == T_WALL || G.failed) break;
                if (tries == LAYOUT_MAX_TRIES - 1) G.failed = true;
            }
        }
    } else {  // locked_hallway.py:146-184
        const int rs = ls.size, nrooms = p.num_rooms(), nrows = nrooms / 2;
        G.rooms(rs, nrows, 3);
        uint8_t seq[18], doors[14];
        const int reps = (nrooms + 5) / 6;
        for (int i = 0; i < 6 * reps; ++i) seq[i] = static_cast<uint8_t>(i % 6);
        G.shuffle(seq, 6 * reps);
        for (int r = 0; r < nrows - 1; ++r)  // the hallway is one room: remove_wall(down), roomgrid.py:347-349
            for (int i = 1; i < rs - 1; ++i) G.set((rs - 1) + i, r * (rs - 1) + rs - 1, T_EMPTY, 0, 0);
        for (int i = 0; i < nrooms; ++i) doors[i] = seq[i];
        G.shuffle(doors, nrooms);
        uint8_t room_col[6], room_row[6];
        int left = nrooms;
        for (int r = 0; r < nrows; ++r)
            for (int side = 0; side < 2; ++side) {
                const int color = doors[--left];
                const int col = side == 0 ? 0 : 2;
                room_col[color] = static_cast<uint8_t>(col);
                room_row[color] = static_cast<uint8_t>(r);
                const int top = r * (rs - 1), lft = col * (rs - 1);
                G.set(side == 0 ? lft + rs - 1 : lft, (top + top + rs - 1) / 2, T_DOOR, color, S_LOCKED);
            }
        const int hall_keys = G.rand_int(1, ls.max_hallway_keys + 1);
        int x, y;
        for (int i = 0; i < min(hall_keys, nrooms); ++i) {
            G.place(rs - 1, 0, rs, H, false, x, y);
            G.set(x, y, T_KEY, seq[i], 4);  // Key() is "available", world_object.py:519-521
        }
        int key_index = hall_keys;
        while (key_index < nrooms) {
            const int c = seq[key_index - 1];
            const int room_keys = G.rand_int(1, ls.max_keys_per_room + 1);
            const int stop = min(key_index + room_keys, nrooms);
            for (int i = key_index; i < stop; ++i) {
                G.place(room_col[c] * (rs - 1), room_row[c] * (rs - 1), rs, rs, false, x, y);
                G.set(x, y, T_KEY, seq[i], 4);
            }
            key_index = stop;
        }
        for (int k = 0; k < A; ++k) G.place_agent(k, rs - 1, 0, rs, H);
        if (nrooms > 6) {  // the number of distinct door colours, where the all-doors ending is reached (:160-166,213)
            uint32_t colours = 0;
            for (int i = 0; i < nrooms; ++i) colours |= 1u << seq[i];
            meta = static_cast<uint32_t>(__popc(colours)) << 24;
        }
    }
    // header block: door positions and the cells in front of the agents (kept current by the step kernel)
    *reinterpret_cast<uint32_t *>(hd + H_META) = meta;
    uint16_t *fc = reinterpret_cast<uint16_t *>(hd + p.HFC);
    for (int k = 0; k < A; ++k) {
        const uint32_t ci = G.rows[8 * k + R_X] < W ? front_index(p, *reinterpret_cast<const uint32_t *>(G.rows + 8 * k)) : FC_NONE;
        fc[k] = ci == FC_NONE ? static_cast<uint16_t>(0) : G.g[ci];
        if (G.rows[8 * k + R_X] >= W || G.rows[8 * k + R_Y] >= H) G.failed = true;
    }
    if (kind == MG_KIND_REDBLUEDOORS) {
        fc[A] = G.g[(meta & 0xFF) * H + ((meta >> 8) & 0xFF)];
        fc[A + 1] = G.g[((meta >> 16) & 0xFF) * H + (meta >> 24)];
    }
    if (G.failed) atomicAdd(failures, 1);
}

thread_local std::string g_create_error;

// Tuning and test switches are read from the environment only when MGRID_TUNING=1 says so: by default what the
// library does depends on its arguments alone.
const char *tuning_env(const char *name) {
    const char *on = getenv("MGRID_TUNING");
    return (on && on[0] == '1' && on[1] == 0) ? getenv(name) : nullptr;
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------

struct MgEnv {
    MgConfig cfg{};
    KParams kp{};
    size_t lds_bytes = 0;
    int num_tiles = 0;
    // the multi-step kernel (mg_step_n) keeps every grid of a tile in LDS and stages images beside them
    LdsMap o_step{};       // single-step launches with compact images: their own staging width and LDS map
    size_t lds_step = 0;
    int stg_step = MG_STAGE_LANES;
    LdsMap o_multi{};
    size_t lds_multi = 0;  // 0: the configuration does not fit, mg_step_n falls back to one launch per step
    bool multi_enabled = true;
    bool crowd = false;  // 6 to 8 agents that may not overlap: steps take the CROWD kernel variant
    uint8_t *hdr = nullptr, *grid = nullptr, *pool_hdr = nullptr, *pool_grid = nullptr;
    double *schemes = nullptr;  // [A][4], CompetativeRedBlueDoor only
    int64_t *stats = nullptr;
    int num_layouts = 0;
    bool reset_done = false;
    bool pool_walled = false;
    int force_stream = -1;  // tuning / test override of the streaming-store choice: 0 never, 1 always
    uint32_t step_serial = 0;
    uint32_t reset_serial = 0;
    int64_t launches = 0;
    std::string err;
};

namespace {

int fail(MgEnv *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h)
        h->err = buf;
    else
        g_create_error = buf;
    return code;
}

#define MG_HIP(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) return fail(h, MG_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// One instantiation of the tile kernel per combination the dispatch below can reach:
//   view size x see-through x mode, then
//   steps of CompetativeRedBlueDoor environments (general border code, compact images), one or many per launch;
//   steps of the other kinds: many per launch (mg_step_n), or one per launch with the cell-table act phase, or one per
//     launch with the staging width picked for the batch (16 / 32 / 64) - each for walled and for general grids;
//   launches of any mode that also write one-hot images (bound by those stores: general border code);
//   reset and gen_obs launches, walled or general.
// Which instantiations this process has launched so far (mg_launch_census): every launch_k<> owns one counter,
// linked into a list the first time it is used.
struct LaunchCount {
    const char *what;  // the template arguments, "V,SEE,MODE,WALLED,OH,MULTI,COMP,CROWD,STG"
    std::atomic<long> n{0};
    LaunchCount *next = nullptr;
};
std::atomic<LaunchCount *> g_launch_counts{nullptr};

template <int V, bool SEE, int MODE, bool WALLED, bool OH, bool MULTI, bool COMP, bool CROWD, int STG>
void launch_k(const MgEnv *h, const KParams &kp, size_t lds, hipStream_t s) {
    static char what[48];
    static LaunchCount count;
    if (count.n.fetch_add(1, std::memory_order_relaxed) == 0) {
        snprintf(what, sizeof what, "%d,%d,%d,%d,%d,%d,%d,%d,%d", V, SEE, MODE, WALLED, OH, MULTI, COMP, CROWD, STG);
        count.what = what;
        count.next = g_launch_counts.load(std::memory_order_relaxed);
        while (!g_launch_counts.compare_exchange_weak(count.next, &count, std::memory_order_release, std::memory_order_relaxed)) {}
    }
    hipLaunchKernelGGL((mg_tile_kernel<V, SEE, MODE, WALLED, OH, MULTI, COMP, CROWD, STG>), dim3(h->num_tiles), dim3(64), lds, s, kp);
}

template <int V, bool SEE, int MODE>
void launch_vs(const MgEnv *h, const KParams &kp_in, hipStream_t s) {
    KParams kp = kp_in;
    constexpr int STG0 = MG_STAGE_LANES;
    size_t lds = h->lds_bytes;
    kp.o.oh = static_cast<lds_off_t>(h->lds_bytes);  // the one-hot exchange buffer follows everything else
    if constexpr (MODE == MODE_STEP) {
        const bool multi = kp.T > 1;
        if (multi) {
            kp.o = h->o_multi;
            lds = h->lds_multi;
        }
        if (kp.comp()) {
            if (multi) launch_k<V, SEE, MODE, false, false, true, true, false, STG0>(h, kp, lds, s);
            else launch_k<V, SEE, MODE, false, false, false, true, false, STG0>(h, kp, lds, s);
            return;
        }
        if (multi) {  // mg_step_n: one launch for T steps (never with one-hot images)
            if (h->crowd) {
                if (kp.walled()) launch_k<V, SEE, MODE, true, false, true, false, true, STG0>(h, kp, lds, s);
                else launch_k<V, SEE, MODE, false, false, true, false, true, STG0>(h, kp, lds, s);
            } else {
                if (kp.walled()) launch_k<V, SEE, MODE, true, false, true, false, false, STG0>(h, kp, lds, s);
                else launch_k<V, SEE, MODE, false, false, true, false, false, STG0>(h, kp, lds, s);
            }
            return;
        }
        if (!kp.onehot) {
            if (h->crowd) {
                if (kp.walled()) launch_k<V, SEE, MODE, true, false, false, false, true, STG0>(h, kp, lds, s);
                else launch_k<V, SEE, MODE, false, false, false, false, true, STG0>(h, kp, lds, s);
                return;
            }
            if (h->stg_step != STG0) {  // the staging width picked for this batch, with the LDS map that goes with it
                const lds_off_t oh = kp.o.oh;
                kp.o = h->o_step;
                kp.o.oh = oh;
                lds = h->lds_step;
                if (h->stg_step == 16) {
                    if (kp.walled()) launch_k<V, SEE, MODE, true, false, false, false, false, 16>(h, kp, lds, s);
                    else launch_k<V, SEE, MODE, false, false, false, false, false, 16>(h, kp, lds, s);
                } else {
                    if (kp.walled()) launch_k<V, SEE, MODE, true, false, false, false, false, 64>(h, kp, lds, s);
                    else launch_k<V, SEE, MODE, false, false, false, false, false, 64>(h, kp, lds, s);
                }
                return;
            }
        }
    }
    // launches that also write one-hot images are bound by those stores; they share the general-border code
    if (kp.onehot)
        launch_k<V, SEE, MODE, false, true, false, false, false, STG0>(h, kp, lds + OH_XCHG_BYTES, s);
    else if (kp.walled())
        launch_k<V, SEE, MODE, true, false, false, false, false, STG0>(h, kp, lds, s);
    else
        launch_k<V, SEE, MODE, false, false, false, false, false, STG0>(h, kp, lds, s);
}

template <int V, int MODE>
void launch_vm(const MgEnv *h, const KParams &kp, hipStream_t s) {
    if (h->cfg.see_through_walls) launch_vs<V, true, MODE>(h, kp, s); else launch_vs<V, false, MODE>(h, kp, s);
}

// View sizes 9 and 11 (the reference takes any odd agent_view_size, multigrid/base.py:96; none of its registered
// configurations sets one above 7): the same kernel with 61 / 91 registers of image per lane, in its general forms
// only - general border code, default staging width, lane-per-agent act phase without the cell table, one step per
// launch (mg_step_n queues single steps).
template <int V, int MODE>
void launch_big(const MgEnv *h, const KParams &kp_in, hipStream_t s) {
    KParams kp = kp_in;
    constexpr int STG0 = MG_STAGE_LANES;
    kp.o.oh = static_cast<lds_off_t>(h->lds_bytes);
    const bool see = h->cfg.see_through_walls != 0;
    if (kp.onehot) {
        if (see) launch_k<V, true, MODE, false, true, false, false, false, STG0>(h, kp, h->lds_bytes + OH_XCHG_BYTES, s);
        else launch_k<V, false, MODE, false, true, false, false, false, STG0>(h, kp, h->lds_bytes + OH_XCHG_BYTES, s);
        return;
    }
    if constexpr (MODE == MODE_STEP) {
        if (kp.comp()) {
            if (see) launch_k<V, true, MODE, false, false, false, true, false, STG0>(h, kp, h->lds_bytes, s);
            else launch_k<V, false, MODE, false, false, false, true, false, STG0>(h, kp, h->lds_bytes, s);
            return;
        }
    }
    if (see) launch_k<V, true, MODE, false, false, false, false, false, STG0>(h, kp, h->lds_bytes, s);
    else launch_k<V, false, MODE, false, false, false, false, false, STG0>(h, kp, h->lds_bytes, s);
}

template <int MODE>
int launch_tile(MgEnv *h, const KParams &kp, hipStream_t s) {
    switch (h->cfg.view_size) {
#ifndef MG_ONLY_VIEW7  // tuning builds compile the 7 x 7 kernels alone
        case 3: launch_vm<3, MODE>(h, kp, s); break;
        case 5: launch_vm<5, MODE>(h, kp, s); break;
        case 9: launch_big<9, MODE>(h, kp, s); break;
        case 11: launch_big<11, MODE>(h, kp, s); break;
#endif
        case 7: launch_vm<7, MODE>(h, kp, s); break;
        default: return fail(h, MG_EINVAL, "view_size %d not supported", h->cfg.view_size);
    }
    MG_HIP(h, hipGetLastError());
    return MG_OK;
}

// Address of this device's copy of `scratch` (looked up once per device, with that device current).
constexpr int MG_MAX_DEVICES = 64;
uint8_t *g_scratch_ptr[MG_MAX_DEVICES];
std::mutex g_scratch_mutex;

int scratch_for_device(int device, uint8_t **ptr) {
    if (device < 0 || device >= MG_MAX_DEVICES) return fail(nullptr, MG_EINVAL, "device %d not supported (0..%d)", device, MG_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *addr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&addr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !addr)
            return fail(nullptr, MG_EHIP, "hipGetSymbolAddress(scratch) on device %d failed: %s", device, hipGetErrorString(e));
        g_scratch_ptr[device] = static_cast<uint8_t *>(addr);
    }
    *ptr = g_scratch_ptr[device];
    return MG_OK;
}

bool out_ok(const MgOut *out) {  // the one-hot image, the agent rows and the double rewards are written in 8-byte pieces
    return !out || ((reinterpret_cast<uintptr_t>(out->onehot_dev) | reinterpret_cast<uintptr_t>(out->rows_dev) |
                     reinterpret_cast<uintptr_t>(out->reward64_dev)) & 7) == 0;
}

void set_out(KParams &kp, const MgOut *out, int view, int force_stream) {
    kp.image = out ? out->image_dev : nullptr;
    kp.onehot = out ? out->onehot_dev : nullptr;
    kp.direction = out ? out->direction_dev : nullptr;
    kp.reward = out ? out->reward_dev : nullptr;
    kp.terminated = out ? out->terminated_dev : nullptr;
    kp.truncated = out ? out->truncated_dev : nullptr;
    kp.rows_out = out ? out->rows_dev : nullptr;
    kp.reward64 = out ? out->reward64_dev : nullptr;
    kp.info = out ? out->info_dev : nullptr;
    // outputs large enough to be streamed past the caches (thresholds measured on MI355X)
    const size_t cells = static_cast<size_t>(kp.E) * kp.A() * view * view;
    kp.bits &= ~(3u << 21);
    if (force_stream < 0 ? 3 * cells > (4u << 20) : force_stream != 0) kp.bits |= 1u << 21;
    if (force_stream < 0 ? OH_DIM * cells > (512u << 20) : force_stream != 0) kp.bits |= 1u << 22;
}

}  // namespace

extern "C" {

int mg_abi_version(void) { return MG_ABI_VERSION; }

const char *mg_last_error(MgHandle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

}  // extern "C"

namespace {

// Everything mg_create derives from the configuration without touching a device: validation, packed
// kernel parameters, tile shape, LDS map.  Errors go to the creation-error text.
int configure(MgEnv *h, const MgConfig &c) {
    if (c.kind < 0 || c.kind > MG_KIND_COMPETITIVE) return fail(nullptr, MG_EINVAL, "unknown kind %d", c.kind);
    if (c.kind == MG_KIND_COMPETITIVE) {
        if (c.num_agents > 8) return fail(nullptr, MG_EINVAL, "CompetativeRedBlueDoor supports at most 8 agents, got %d", c.num_agents);
        if (c.width < 8 || c.height < 4 || c.width % 4 != 0)
            return fail(nullptr, MG_EINVAL, "CompetativeRedBlueDoor needs a grid of 2 * size x size cells, got %dx%d", c.width, c.height);
        // eliminated agents are put at (13, 2) / (2, 2) whatever the size (competative_red_blue_door.py:504-506)
        if (c.death_match && (c.width != 16 || c.height != 8))
            return fail(nullptr, MG_EINVAL, "CompetativeRedBlueDoor with death_match needs the 16x8 grid (size 8), got %dx%d", c.width, c.height);
    }
    if (c.width < 3 || c.height < 3 || c.width > 255 || c.height > 255 || c.width * c.height > 4096)
        return fail(nullptr, MG_EINVAL, "grid %dx%d not supported (3..255 per side, at most 4096 cells)", c.width,
                    c.height);
    if (c.num_agents < 1 || c.num_agents > 64)
        return fail(nullptr, MG_EINVAL, "num_agents %d not supported (1..64)", c.num_agents);
    if (c.view_size < 3 || c.view_size > 11 || c.view_size % 2 == 0)
        return fail(nullptr, MG_EINVAL, "view_size %d not supported (3, 5, 7, 9 or 11)", c.view_size);
    if (c.max_steps < 1) return fail(nullptr, MG_EINVAL, "max_steps must be positive");
    if (c.num_envs < 1) return fail(nullptr, MG_EINVAL, "num_envs must be positive");
    if (c.batch_envs < 0) return fail(nullptr, MG_EINVAL, "batch_envs must not be negative");
    if (c.autoreset < 0 || c.autoreset > MG_AUTORESET_NEXT_STEP)
        return fail(nullptr, MG_EINVAL, "unknown autoreset mode %d", c.autoreset);
    if (c.kind == MG_KIND_LOCKEDHALLWAY && (c.num_rooms < 1 || c.num_rooms > 14))
        return fail(nullptr, MG_EINVAL, "LockedHallway needs 1..14 rooms, got %d", c.num_rooms);
    // more than six rooms: door colours repeat and a door is told by the room it belongs to, which takes the RoomGrid
    // geometry of locked_hallway.py:138-145 (three columns of rooms, num_rooms / 2 rows, shared walls)
    if (c.kind == MG_KIND_LOCKEDHALLWAY && c.num_rooms > 6 &&
        (c.num_rooms % 2 != 0 || c.width < 10 || (c.width - 1) % 3 != 0 || c.height != (c.num_rooms / 2) * ((c.width - 1) / 3) + 1))
        return fail(nullptr, MG_EINVAL, "LockedHallway with %d rooms needs an even number of rooms on a %d-row, 3-column room grid, got %dx%d",
                    c.num_rooms, c.num_rooms / 2, c.width, c.height);

    h->cfg = c;
    KParams &kp = h->kp;
    kp.geo = static_cast<uint32_t>(c.width) | (static_cast<uint32_t>(c.height) << 8) |
             (static_cast<uint32_t>(c.num_agents) << 16);
    // (CompetativeRedBlueDoor: low kind bits 0, its own bit 23, then death_match and per_agent_outputs)
    kp.bits = (c.kind == MG_KIND_COMPETITIVE ? ((1u << 23) | (c.death_match ? 1u << 24 : 0u) | (c.per_agent_outputs ? 1u << 25 : 0u))
                                             : static_cast<uint32_t>(c.kind)) |
              (c.allow_agent_overlap ? 4u : 0u) | (c.joint_reward ? 8u : 0u) |
              (c.success_any ? 16u : 0u) | (c.failure_any ? 32u : 0u) | (static_cast<uint32_t>(c.autoreset) << 6) |
              (static_cast<uint32_t>(c.num_rooms & 15) << 8);
    kp.max_steps = c.max_steps;
    kp.E = c.num_envs;
    kp.HFC = 16 + 8 * round_up(c.num_agents, 2);
    kp.HS = round_up(kp.HFC + 2 * (c.num_agents + 2), 16);
    kp.GS = round_up(2 * c.width * c.height, 16);
#ifndef MG_HDR_PITCH_PAD
#define MG_HDR_PITCH_PAD 16
#endif
    kp.HP = kp.HS + (((kp.HS / 16) & 1) ? 0 : MG_HDR_PITCH_PAD);
    // In LDS 16 bytes separate the grids of a sub-tile, which spreads most grid sizes over the banks.  Making the
    // distance an odd number of 16-byte pieces for every size (MG_ODD_GRID_PITCH: 13 x 9 and 11 x 6 grids then get 32
    // bytes) measured no faster on 65536 x 4 agents of LockedHallway and 2 % slower on 2-agent batches (more LDS
    // per wavefront), so it stays a tuning switch.
#ifdef MG_ODD_GRID_PITCH
    kp.GP = kp.GS + (((kp.GS / 16) & 1) ? 32 : 16);
#else
    kp.GP = kp.GS + 16;
#endif
    kp.env_offset = static_cast<uint32_t>(c.env_id_offset);
    kp.seed = c.seed;

    // Tile shape.  NE environments per observation pass give at most 64 agent lanes; one wavefront
    // owns G such sub-tiles so that its act phase (one lane per environment) fills up to 64 lanes.
    // Small batches get small tiles: the launch then has enough wavefronts to cover the chip and
    // each wavefront a short critical path.
    const int img = 3 * c.view_size * c.view_size;
    // (CompetativeRedBlueDoor environments act in order, one lane per environment; all others one lane per agent)
    const bool big_view = c.view_size > 7;  // 9 x 9 and 11 x 11 views: the kernel's general forms only (launch_big)
    h->crowd = c.kind != MG_KIND_COMPETITIVE && !c.allow_agent_overlap && c.num_agents >= CROWD_MIN_AGENTS && c.num_agents <= 8 &&
               !big_view;
    if (const char *ev = tuning_env("MGRID_CROWD")) h->crowd = h->crowd && atoi(ev) != 0;  // tuning / test override
    auto lds_for = [&](int ne, int g, bool multi = false, int stg = STAGE_LANES) {
        const int et = ne * g, np = ne * c.num_agents;
        int o = 0;
        auto take = [&](int bytes) { const int at = o; o += round_up(bytes, 16); return at; };
        kp.o.hdr = take(et * kp.HP);
        kp.o.log = take(4 * et * c.num_agents);
        kp.o.act = take(et * c.num_agents);
        kp.o.ord = take(et * c.num_agents);
        kp.o.rewm = take(et * c.num_agents);
        kp.o.tout = take(et * c.num_agents);
        kp.o.oterm = take(et * c.num_agents);
        kp.o.flag = take(4 * et);
        kp.o.R = take(8 * et);
        kp.o.src = take(4 * et);
        if (c.kind == MG_KIND_COMPETITIVE) {
            kp.o.racc = take(8 * et * c.num_agents);
            kp.o.pxy = take(2 * et * c.num_agents);
            kp.o.log2 = take(4 * et * c.num_agents);
        }
        const bool table = h->crowd;  // act_lanes() cell table
        if (multi) {  // every grid of the tile stays; images are staged beside them
            kp.o.grid = take(et * kp.GP);
            // the staging buffer is idle during the act phase: act_lanes()' scratch or cell table lies there too
            kp.o.stage = take(std::max(std::max(std::min(np, stg) * img + 32, 8 * np), table ? ne * kp.GP : 0));
            kp.o.img = kp.o.stage;
            kp.o.tbl = kp.o.stage;
            return o;
        }
        kp.o.grid = take(ne * kp.GP);  // 16 bytes between grids spread them over the LDS banks
        kp.o.tbl = kp.o.grid;
        const int grids = round_up(ne * kp.GP, 16);
        kp.o.stage = kp.o.grid;  // images are staged over the grids of the sub-tile they were made from
        kp.o.img = take(std::max(std::min(np, stg) * img + 32 - grids, 8 * np));  // act_lanes() scratch + overflow
        return o;
    };
    // Measured on MI355X: with the in-order act phase (one lane per environment) about 32 environments
    // per wavefront balance lane use against LDS-limited occupancy; with the lane-per-agent act phase
    // 16 are enough and leave room for more wavefronts per CU, and one full sub-tile of 8 agents x 8
    // environments is (slightly) better still.
    // A handle that holds one slice of a larger batch (batch_envs, see mg_step_shards) runs beside the other
    // slices' launches: its tiles are shaped for the whole batch, not for the slice.
    const int shape_envs = std::max(c.num_envs, c.batch_envs);
    int ne = 64 / c.num_agents;
    const int want_waves = 2048;
    while (ne > 1 && (shape_envs + ne - 1) / ne < want_waves && ne * c.num_agents > 16) ne = (ne + 1) / 2;
    while (ne > 1 && lds_for(ne, 1) > 40 * 1024) --ne;
    int g = 1;
    const int et_target = c.num_agents >= 6 ? (c.num_agents <= 8 ? 8 : 16) : 32;
    while ((g + 1) * ne <= et_target && (shape_envs + (g + 1) * ne - 1) / ((g + 1) * ne) >= want_waves &&
           lds_for(ne, g + 1) <= 24 * 1024)
        ++g;
    if (const char *ev = tuning_env("MGRID_STREAM")) h->force_stream = atoi(ev) != 0;  // tuning override: "0" / "1"
    if (const char *ev = tuning_env("MGRID_TILE")) {  // tuning override: "NE,G"
        int a = 0, b = 0;
        if (sscanf(ev, "%d,%d", &a, &b) == 2 && a >= 1 && b >= 1 && a * c.num_agents <= 64 && a * b <= 64) {
            ne = a;
            g = b;
        }
    }
    {  // the same tile shape for the multi-step kernel, where it fits
        const int need = lds_for(ne, g, true);
        h->o_multi = kp.o;
        h->o_multi.oh = 0;
        h->lds_multi = need <= 64 * 1024 && !big_view ? static_cast<size_t>(need) : 0;
        if (const char *ev = tuning_env("MGRID_MULTI_STEP")) h->multi_enabled = atoi(ev) != 0;  // tuning / test override
    }
    {
        // Staging width of the plain step kernel, measured on MI355X.  Launches of at most one round of wavefronts
        // (8 per CU) run at the speed of one wavefront's dependent chain: the whole sub-tile's images leave in one go
        // (64; 2 % faster on 65536 x 2 and x 4 agents than 32).  Many-round launches of the lane-per-agent kind are
        // bound by how many wavefronts fit a CU and by their store bursts: small blocks (16; 2 % faster on
        // 131072 x 8 agents than 32 and 5 % on 2^20 x 8, where 64 is 6 % slower).  The cell-table variant stays
        // at 32 (16 measured 2 % slower there).
        const int tiles = (c.num_envs + ne * g - 1) / (ne * g);
        h->stg_step = (h->crowd || c.num_agents > 8 || big_view) ? 32 : tiles <= 2048 ? 64 : (c.num_agents >= 6 ? 16 : 32);
        if (const char *ev = tuning_env("MGRID_STAGE")) {  // tuning / test override
            const int v = atoi(ev);
            if (!h->crowd && c.num_agents <= 8 && !big_view && (v == 16 || v == 32 || v == 64)) h->stg_step = v;
        }
        h->lds_step = static_cast<size_t>(lds_for(ne, g, false, h->stg_step));
        if (h->lds_step > 64 * 1024) {  // does not fit: the default width
            h->stg_step = MG_STAGE_LANES;
            h->lds_step = static_cast<size_t>(lds_for(ne, g, false, h->stg_step));
        }
        h->o_step = kp.o;
    }
    h->lds_bytes = static_cast<size_t>(lds_for(ne, g));
    MG_EXP(
    if (const char *ev = tuning_env("MGRID_LDS_PAD")) h->lds_bytes += static_cast<size_t>(atoi(ev));
    if (const char *ev = tuning_env("MGRID_DBG")) {
        kp.dbg = atoi(ev);
        if (kp.dbg & 32) h->lds_bytes = static_cast<size_t>(kp.o.grid);
    }
    )
    if (h->lds_bytes + OH_XCHG_BYTES > 64 * 1024)
        return fail(nullptr, MG_EINVAL, "configuration needs %zu bytes of LDS per environment tile", h->lds_bytes);
    kp.geo |= static_cast<uint32_t>(ne) << 24;
    kp.bits |= static_cast<uint32_t>(g) << 12;
    kp.magic_a = c.num_agents == 1 ? 0u : 0xFFFFFFFFu / c.num_agents + 1u;
    kp.magic_nc = 0xFFFFFFFFu / (c.num_agents + 2) + 1u;
    kp.magic_per = kp.GS == 16 ? 0u : 0xFFFFFFFFu / (kp.GS / 16) + 1u;
    kp.magic_ph = kp.HS == 16 ? 0u : 0xFFFFFFFFu / (kp.HS / 16) + 1u;
    h->num_tiles = (c.num_envs + ne * g - 1) / (ne * g);
    return MG_OK;
}

}  // namespace

extern "C" {

int mg_create(const MgConfig *cfg, MgHandle *out) {
    if (!cfg || !out) return fail(nullptr, MG_EINVAL, "mg_create: null argument");
    *out = nullptr;
    MgEnv *h = new (std::nothrow) MgEnv();
    if (!h) return fail(nullptr, MG_EINVAL, "out of host memory");
    const int rc = configure(h, *cfg);
    if (rc != MG_OK) {
        delete h;
        return rc;
    }
    const MgConfig &c = h->cfg;
    KParams &kp = h->kp;

    hipError_t e = hipSetDevice(c.device);
    if (e == hipSuccess) e = hipMalloc(&h->hdr, static_cast<size_t>(c.num_envs) * kp.HS);
    if (e == hipSuccess) e = hipMalloc(&h->grid, static_cast<size_t>(c.num_envs) * kp.GS);
    if (e == hipSuccess) e = hipMalloc(&h->stats, (64 * MG_NUM_STATS + 1) * sizeof(int64_t));  // + mg_set_state's counter
    if (e == hipSuccess) e = hipMemset(h->hdr, 0, static_cast<size_t>(c.num_envs) * kp.HS);
    if (e == hipSuccess) e = hipMemset(h->grid, 0, static_cast<size_t>(c.num_envs) * kp.GS);
    if (e == hipSuccess) e = hipMemset(h->stats, 0, (64 * MG_NUM_STATS + 1) * sizeof(int64_t));
    if (e != hipSuccess) {
        fail(nullptr, MG_EHIP, "mg_create: %s", hipGetErrorString(e));
        mg_destroy(h);
        return MG_EHIP;
    }
    if (e == hipSuccess && c.kind == MG_KIND_COMPETITIVE) {  // every reward zero until mg_set_reward_schemes
        e = hipMalloc(&h->schemes, sizeof(double) * 4 * c.num_agents);
        if (e == hipSuccess) e = hipMemset(h->schemes, 0, sizeof(double) * 4 * c.num_agents);
        if (e != hipSuccess) {
            fail(nullptr, MG_EHIP, "mg_create: %s", hipGetErrorString(e));
            mg_destroy(h);
            return MG_EHIP;
        }
    }
    kp.hdr = h->hdr; kp.grid = h->grid; kp.stats = h->stats; kp.schemes = h->schemes;
    *out = h;
    return MG_OK;
}

int mg_destroy(MgHandle h) {
    if (!h) return MG_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->hdr) (void)hipFree(h->hdr);
    if (h->grid) (void)hipFree(h->grid);
    if (h->pool_hdr) (void)hipFree(h->pool_hdr);
    if (h->pool_grid) (void)hipFree(h->pool_grid);
    if (h->stats) (void)hipFree(h->stats);
    if (h->schemes) (void)hipFree(h->schemes);
    delete h;
    return MG_OK;
}

int mg_upload_layouts(MgHandle h, const uint8_t *grids, const uint8_t *agents, const uint8_t *meta, int32_t n) {
    if (!h) return MG_EINVAL;
    if (!grids || !agents || n < 1) return fail(h, MG_EINVAL, "mg_upload_layouts: null pointer or n < 1");
    if (h->kp.kind() == MG_KIND_REDBLUEDOORS && !meta)
        return fail(h, MG_EINVAL, "mg_upload_layouts: RedBlueDoors layouts need `meta` (the red and blue door positions)");
    const KParams &kp = h->kp;
    const int cells = kp.W() * kp.H(), A = kp.A();
    std::vector<uint8_t> ph(static_cast<size_t>(n) * kp.HS, 0), pg(static_cast<size_t>(n) * kp.GS, 0);
    bool walled = true;  // every layout ringed by walls that encode like "outside the grid" (obs.py:13)
    for (int l = 0; l < n; ++l) {
        uint8_t *hd = ph.data() + static_cast<size_t>(l) * kp.HS;
        uint16_t *gd = reinterpret_cast<uint16_t *>(pg.data() + static_cast<size_t>(l) * kp.GS);
        for (int c = 0; c < cells; ++c) {
            const uint8_t *s = grids + (static_cast<size_t>(l) * cells + c) * 3;
            if (s[0] > 15 || s[1] > 15 || s[2] > 15)
                return fail(h, MG_EINVAL, "layout %d cell %d = (%d,%d,%d) exceeds type<16, color<16, state<16", l, c,
                            s[0], s[1], s[2]);
            const uint32_t opaque = (s[0] == T_WALL) | ((s[0] == T_DOOR) & (s[2] != S_OPEN));
            gd[c] = static_cast<uint16_t>(s[0] | (s[2] << 4) | (s[1] << 8) | (opaque << 15));
        }
        for (int x = 0; x < kp.W(); ++x)
            for (int y = 0; y < kp.H(); ++y)
                if ((x == 0 || y == 0 || x == kp.W() - 1 || y == kp.H() - 1) && gd[x * kp.H() + y] != CELL_WALL) walled = false;
        if (meta) memcpy(hd + H_META, meta + static_cast<size_t>(l) * 4, 4);
        if (kp.kind() == MG_KIND_LOCKEDHALLWAY && kp.num_rooms() > 6) {
            // no doors paid yet; the all-doors ending comes at the number of distinct door colours (locked_hallway.py:160-166,213)
            uint32_t colours = 0;
            for (int c = 0; c < cells; ++c)
                if ((gd[c] & 15) == T_DOOR) colours |= 1u << ((gd[c] >> 8) & 15);
            hd[H_META] = hd[H_META + 1] = hd[H_META + 2] = 0;
            hd[H_META + 3] = static_cast<uint8_t>(__builtin_popcount(colours));
        }
        const uint8_t *ar = agents + static_cast<size_t>(l) * A * 8;
        for (int k = 0; k < A; ++k) {
            if (ar[8 * k + R_X] >= kp.W() || ar[8 * k + R_Y] >= kp.H() || ar[8 * k + R_DIR] > 3)
                return fail(h, MG_EINVAL, "layout %d agent %d at (%d,%d) dir %d is outside the grid", l, k,
                            ar[8 * k + R_X], ar[8 * k + R_Y], ar[8 * k + R_DIR]);
            if (ar[8 * k + R_CT] > 15 || ar[8 * k + R_CC] > 15 || ar[8 * k + R_CS] > 15 || ar[8 * k + R_COLOR] > 15)
                return fail(h, MG_EINVAL, "layout %d agent %d carries an out-of-range encoding", l, k);
        }
        memcpy(hd + H_ROWS, ar, static_cast<size_t>(A) * 8);
        // front cells of the start state (the step kernel keeps them current afterwards)
        uint16_t *fc = reinterpret_cast<uint16_t *>(hd + kp.HFC);
        static const int DX[4] = {1, 0, -1, 0}, DY[4] = {0, 1, 0, -1};
        for (int k = 0; k < A; ++k) {
            const int fx = ar[8 * k + R_X] + DX[ar[8 * k + R_DIR]], fy = ar[8 * k + R_Y] + DY[ar[8 * k + R_DIR]];
            fc[k] = (fx >= 0 && fx < kp.W() && fy >= 0 && fy < kp.H()) ? gd[fx * kp.H() + fy] : 0;
        }
        if (kp.kind() == MG_KIND_REDBLUEDOORS && meta) {
            const uint8_t *m = meta + static_cast<size_t>(l) * 4;
            if (m[0] >= kp.W() || m[2] >= kp.W() || m[1] >= kp.H() || m[3] >= kp.H())
                return fail(h, MG_EINVAL, "layout %d door position outside the grid", l);
            fc[A] = gd[m[0] * kp.H() + m[1]];
            fc[A + 1] = gd[m[2] * kp.H() + m[3]];
            // Color.red = 0, Color.blue = 2 (multigrid/core/constants.py:50-59)
            if ((fc[A] & 0x0F0F) != (T_DOOR | (0u << 8)) || (fc[A + 1] & 0x0F0F) != (T_DOOR | (2u << 8)))
                return fail(h, MG_EINVAL, "layout %d: meta does not name a red door at (%d,%d) and a blue door at (%d,%d)", l,
                            m[0], m[1], m[2], m[3]);
        }
        if (kp.comp()) {  // the doors of competative_red_blue_door.py:201-211
            fc[A] = gd[(kp.W() / 4) * kp.H() + kp.H() - 2];
            fc[A + 1] = gd[(kp.W() / 4 + kp.W() / 2 - 1) * kp.H() + 1];
        }
    }
    MG_HIP(h, hipSetDevice(h->cfg.device));
    MG_HIP(h, hipDeviceSynchronize());
    // the new pool is complete on the device before it replaces the old one: a failure leaves the handle as it was
    uint8_t *nh = nullptr, *ng = nullptr;
    hipError_t e = hipMalloc(&nh, ph.size());
    if (e == hipSuccess) e = hipMalloc(&ng, pg.size());
    if (e == hipSuccess) e = hipMemcpy(nh, ph.data(), ph.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ng, pg.data(), pg.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (nh) (void)hipFree(nh);
        if (ng) (void)hipFree(ng);
        return fail(h, MG_EHIP, "mg_upload_layouts: %s", hipGetErrorString(e));
    }
    if (h->pool_hdr) (void)hipFree(h->pool_hdr);
    if (h->pool_grid) (void)hipFree(h->pool_grid);
    h->pool_hdr = nh;
    h->pool_grid = ng;
    h->num_layouts = n;
    h->kp.pool_hdr = h->pool_hdr;
    h->kp.pool_grid = h->pool_grid;
    h->kp.num_layouts = n;
    h->pool_walled = walled;
    if (!walled) h->kp.bits &= ~(1u << 20);
    return MG_OK;
}

int mg_set_reward_schemes(MgHandle h, const double *schemes) {
    if (!h) return MG_EINVAL;
    if (!schemes) return fail(h, MG_EINVAL, "mg_set_reward_schemes: null pointer");
    if (!h->schemes) return fail(h, MG_EINVAL, "mg_set_reward_schemes: only MG_KIND_COMPETITIVE has reward schemes");
    MG_HIP(h, hipSetDevice(h->cfg.device));
    MG_HIP(h, hipDeviceSynchronize());
    MG_HIP(h, hipMemcpy(h->schemes, schemes, sizeof(double) * 4 * h->cfg.num_agents, hipMemcpyHostToDevice));
    return MG_OK;
}

int mg_generate_layouts(MgHandle h, const MgLayoutSpec *spec, int32_t n, uint64_t seed) {
    if (!h) return MG_EINVAL;
    if (!spec || n < 1) return fail(h, MG_EINVAL, "mg_generate_layouts: null spec or n < 1");
    if (h->kp.comp()) return fail(h, MG_EINVAL, "mg_generate_layouts: not available for MG_KIND_COMPETITIVE (upload host layouts)");
    const KParams &kp = h->kp;
    const int W = kp.W(), H = kp.H(), kind = kp.kind();
    // the grid size must be the one the family builds
    bool ok = true;
    if (kind == MG_KIND_EMPTY) {
        ok = spec->size == W && W == H &&
             (spec->random_start || (spec->start_x >= 0 && spec->start_x < W && spec->start_y >= 0 && spec->start_y < H &&
                                     spec->start_dir >= 0 && spec->start_dir < 4));
    } else if (kind == MG_KIND_REDBLUEDOORS) {
        ok = W == 2 * spec->size && H == spec->size && spec->size >= 4;
    } else if (kind == MG_KIND_BLOCKEDUNLOCKPICKUP) {
        ok = spec->size >= 4 && W == 2 * (spec->size - 1) + 1 && H == spec->size;
    } else {
        ok = spec->size >= 4 && kp.num_rooms() >= 2 && kp.num_rooms() % 2 == 0 && W == 3 * (spec->size - 1) + 1 &&
             H == (kp.num_rooms() / 2) * (spec->size - 1) + 1 && spec->max_hallway_keys >= 1 && spec->max_keys_per_room >= 1;
    }
    if (!ok) return fail(h, MG_EINVAL, "mg_generate_layouts: spec does not match the %dx%d grid of this handle", W, H);
    for (int k = 0; k < kp.A(); ++k)
        if (spec->agent_color[k] > 15) return fail(h, MG_EINVAL, "mg_generate_layouts: agent colour out of range");
    MG_HIP(h, hipSetDevice(h->cfg.device));
    MG_HIP(h, hipDeviceSynchronize());
    uint8_t *ph = nullptr, *pg = nullptr;
    int *failures = nullptr;
    const size_t hb = static_cast<size_t>(n) * kp.HS, gb = static_cast<size_t>(n) * kp.GS;
    hipError_t e = hipMalloc(&ph, hb);
    if (e == hipSuccess) e = hipMalloc(&pg, gb);
    if (e == hipSuccess) e = hipMalloc(&failures, sizeof(int));
    if (e == hipSuccess) e = hipMemset(failures, 0, sizeof(int));
    int failed = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mg_generate_layouts_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, kp, *spec, ph, pg, n, seed,
                           failures);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&failed, failures, sizeof(int), hipMemcpyDeviceToHost);
    if (failures) (void)hipFree(failures);
    if (e != hipSuccess || failed) {
        if (ph) (void)hipFree(ph);
        if (pg) (void)hipFree(pg);
        if (e != hipSuccess) return fail(h, MG_EHIP, "mg_generate_layouts: %s", hipGetErrorString(e));
        return fail(h, MG_EINVAL, "mg_generate_layouts: %d of %d layouts could not be completed (rooms too full?)", failed, n);
    }
    if (h->pool_hdr) (void)hipFree(h->pool_hdr);
    if (h->pool_grid) (void)hipFree(h->pool_grid);
    h->pool_hdr = ph;
    h->pool_grid = pg;
    h->num_layouts = n;
    h->kp.pool_hdr = ph;
    h->kp.pool_grid = pg;
    h->kp.num_layouts = n;
    h->pool_walled = true;  // every family rings its grid with (2, 5, 0) walls
    return MG_OK;
}

int mg_reset(MgHandle h, const int32_t *layout_idx_dev, const uint8_t *mask_dev, const MgOut *out, void *stream) {
    if (!h) return MG_EINVAL;
    if (h->num_layouts < 1) return fail(h, MG_ESTATE, "mg_reset: no layouts uploaded");
    if (!out_ok(out)) return fail(h, MG_EINVAL, "mg_reset: onehot_dev, rows_dev and reward64_dev must be 8-byte aligned");
    MG_HIP(h, hipSetDevice(h->cfg.device));
    KParams kp = h->kp;
    kp.layout_idx = layout_idx_dev;
    kp.mask = mask_dev;
    kp.serial = h->reset_serial;  // 0: first reset, episode counters start at 0
    set_out(kp, out, h->cfg.view_size, h->force_stream);
    const int rc = launch_tile<MODE_RESET>(h, kp, static_cast<hipStream_t>(stream));
    if (rc != MG_OK) return rc;
    h->reset_serial++;
    h->reset_done = true;
    // after a reset of every environment all grids come from the pool
    if (!mask_dev && h->pool_walled) h->kp.bits |= 1u << 20;
    return MG_OK;
}

}  // extern "C"

namespace {
// what mg_step checks before it launches anything
int step_ready(MgHandle h, const uint8_t *actions_dev, const MgOut *out) {
    if (!actions_dev) return fail(h, MG_EINVAL, "mg_step: actions_dev is null");
    if (!out_ok(out)) return fail(h, MG_EINVAL, "mg_step: onehot_dev, rows_dev and reward64_dev must be 8-byte aligned");
    if (h->kp.comp() && out && out->onehot_dev)
        return fail(h, MG_EINVAL, "mg_step: one-hot images are not written for MG_KIND_COMPETITIVE (use mg_gen_obs)");
    if (!h->reset_done) return fail(h, MG_ESTATE, "mg_step: call mg_reset (or mg_set_state) first");
    if (h->cfg.autoreset != MG_AUTORESET_NONE && h->num_layouts < 1)
        return fail(h, MG_ESTATE, "mg_step: autoreset needs uploaded layouts");
    return MG_OK;
}
}  // namespace

extern "C" {

int mg_step(MgHandle h, const uint8_t *actions_dev, const uint8_t *order_dev, const MgOut *out, void *stream) {
    if (!h) return MG_EINVAL;
    const int ready = step_ready(h, actions_dev, out);
    if (ready != MG_OK) return ready;
    MG_HIP(h, hipSetDevice(h->cfg.device));
    KParams kp = h->kp;
    kp.actions = actions_dev;
    kp.order = order_dev;
    kp.serial = h->step_serial;
    kp.T = 1;
    set_out(kp, out, h->cfg.view_size, h->force_stream);
    const int rc = launch_tile<MODE_STEP>(h, kp, static_cast<hipStream_t>(stream));
    if (rc != MG_OK) return rc;
    h->step_serial++;
    h->launches++;
    return MG_OK;
}

int mg_step_n(MgHandle h, int32_t num_steps, const uint8_t *actions_dev, const uint8_t *order_dev, const MgOut *out,
              void *stream) {
    if (!h) return MG_EINVAL;
    if (num_steps < 1) return fail(h, MG_EINVAL, "mg_step_n: num_steps must be positive");
    if (num_steps > 1 && h->lds_multi && h->multi_enabled && !(out && (out->onehot_dev || out->rows_dev || out->reward64_dev))) {
        // ONE launch takes all the steps: header blocks and grids stay in LDS in between, per step only the
        // actions come in and the observations, rewards and flags go out
        if (!actions_dev) return fail(h, MG_EINVAL, "mg_step_n: actions_dev is null");
        if (!h->reset_done) return fail(h, MG_ESTATE, "mg_step_n: call mg_reset (or mg_set_state) first");
        if (h->cfg.autoreset != MG_AUTORESET_NONE && h->num_layouts < 1)
            return fail(h, MG_ESTATE, "mg_step_n: autoreset needs uploaded layouts");
        MG_HIP(h, hipSetDevice(h->cfg.device));
        KParams kp = h->kp;
        kp.actions = actions_dev;
        kp.order = order_dev;
        kp.serial = h->step_serial;
        kp.T = num_steps;
        set_out(kp, out, h->cfg.view_size, h->force_stream);
        const int rc = launch_tile<MODE_STEP>(h, kp, static_cast<hipStream_t>(stream));
        if (rc != MG_OK) return rc;
        h->step_serial += static_cast<uint32_t>(num_steps);
        h->launches += num_steps;
        return MG_OK;
    }
    // otherwise one launch per step, all queued on the stream by this one call; step t reads slice t of the
    // inputs and writes slice t of every output
    const size_t per = static_cast<size_t>(h->kp.E) * h->kp.A();
    const size_t img = 3u * h->cfg.view_size * h->cfg.view_size;
    for (int32_t t = 0; t < num_steps; ++t) {
        MgOut o{};
        if (out) {
            o.image_dev = out->image_dev ? out->image_dev + t * per * img : nullptr;
            o.onehot_dev = out->onehot_dev ? out->onehot_dev + t * per * img * (OH_DIM / 3) : nullptr;
            o.direction_dev = out->direction_dev ? out->direction_dev + t * per : nullptr;
            o.reward_dev = out->reward_dev ? out->reward_dev + t * per : nullptr;
            o.terminated_dev = out->terminated_dev ? out->terminated_dev + t * per : nullptr;
            o.truncated_dev = out->truncated_dev ? out->truncated_dev + t * per : nullptr;
            o.rows_dev = out->rows_dev ? out->rows_dev + t * per * 8 : nullptr;
            o.reward64_dev = out->reward64_dev ? out->reward64_dev + t * per : nullptr;
            o.info_dev = out->info_dev ? out->info_dev + t * per : nullptr;
        }
        const int rc = mg_step(h, actions_dev ? actions_dev + t * per : nullptr, order_dev ? order_dev + t * per : nullptr,
                               out ? &o : nullptr, stream);
        if (rc != MG_OK) return rc;
    }
    return MG_OK;
}

int mg_step_shards(const MgHandle *shards, int32_t num_shards, const uint8_t *actions_dev, const uint8_t *order_dev,
                   const MgOut *out, void *const *streams) {
    if (!shards || num_shards < 1 || !shards[0]) return fail(nullptr, MG_EINVAL, "mg_step_shards: no shards");
    if (!actions_dev) return fail(shards[0], MG_EINVAL, "mg_step_shards: actions_dev is null");
    // the shards must be consecutive slices of one batch on one device: the same results as one handle holding it all
    const MgConfig &c0 = shards[0]->cfg;
    int64_t first = 0;
    for (int32_t k = 0; k < num_shards; ++k) {
        const MgHandle h = shards[k];
        if (!h) return fail(shards[0], MG_EINVAL, "mg_step_shards: shard %d is null", k);
        const MgConfig &c = h->cfg;
        if (c.num_agents != c0.num_agents || c.view_size != c0.view_size || c.device != c0.device)
            return fail(shards[0], MG_EINVAL, "mg_step_shards: shard %d differs from shard 0 in agents, view size or device", k);
        if (static_cast<int64_t>(c.env_id_offset) != static_cast<int64_t>(c0.env_id_offset) + first)
            return fail(shards[0], MG_EINVAL, "mg_step_shards: shard %d has env_id_offset %d, its slice starts at %lld", k,
                        c.env_id_offset, static_cast<long long>(c0.env_id_offset + first));
        for (int32_t j = 0; j < k; ++j)
            if (shards[j] == h) return fail(shards[0], MG_EINVAL, "mg_step_shards: shard %d is listed twice", k);
        // no shard steps unless all of them can (the slices' output pointers share the alignment that matters)
        const int ready = step_ready(h, actions_dev, out);
        if (ready != MG_OK) {
            if (k > 0) fail(shards[0], ready, "mg_step_shards: shard %d: %s", k, h->err.c_str());
            return ready;
        }
        first += c.num_envs;
    }
    const size_t a = static_cast<size_t>(c0.num_agents), img = 3u * c0.view_size * c0.view_size;
    size_t at = 0;  // agents before this shard's slice
    for (int32_t k = 0; k < num_shards; ++k) {
        MgOut o{};
        if (out) {
            o.image_dev = out->image_dev ? out->image_dev + at * img : nullptr;
            o.onehot_dev = out->onehot_dev ? out->onehot_dev + at * img * (OH_DIM / 3) : nullptr;
            o.direction_dev = out->direction_dev ? out->direction_dev + at : nullptr;
            o.reward_dev = out->reward_dev ? out->reward_dev + at : nullptr;
            o.terminated_dev = out->terminated_dev ? out->terminated_dev + at : nullptr;
            o.truncated_dev = out->truncated_dev ? out->truncated_dev + at : nullptr;
            o.rows_dev = out->rows_dev ? out->rows_dev + at * 8 : nullptr;
            o.reward64_dev = out->reward64_dev ? out->reward64_dev + at : nullptr;
            o.info_dev = out->info_dev ? out->info_dev + at : nullptr;
        }
        const int rc = mg_step(shards[k], actions_dev + at, order_dev ? order_dev + at : nullptr, out ? &o : nullptr,
                               streams ? streams[k] : nullptr);
        if (rc != MG_OK) {
            if (k > 0) fail(shards[0], rc, "mg_step_shards: shard %d: %s", k, shards[k]->err.c_str());
            return rc;
        }
        at += static_cast<size_t>(shards[k]->cfg.num_envs) * a;
    }
    return MG_OK;
}

int mg_gen_obs(MgHandle h, const MgOut *out, void *stream) {
    if (!h) return MG_EINVAL;
    if (!h->reset_done) return fail(h, MG_ESTATE, "mg_gen_obs: call mg_reset (or mg_set_state) first");
    if (!out_ok(out)) return fail(h, MG_EINVAL, "mg_gen_obs: onehot_dev, rows_dev and reward64_dev must be 8-byte aligned");
    MG_HIP(h, hipSetDevice(h->cfg.device));
    KParams kp = h->kp;
    set_out(kp, out, h->cfg.view_size, h->force_stream);
    return launch_tile<MODE_OBS>(h, kp, static_cast<hipStream_t>(stream));
}

int mg_gen_obs_of_state(const MgConfig *cfg, const uint8_t *grid_dev, const uint8_t *agents_dev, const MgOut *out,
                        void *stream) {
    if (!cfg || !grid_dev || !agents_dev || !out) return fail(nullptr, MG_EINVAL, "mg_gen_obs_of_state: null argument");
    if (!out_ok(out)) return fail(nullptr, MG_EINVAL, "mg_gen_obs_of_state: onehot_dev, rows_dev and reward64_dev must be 8-byte aligned");
    MgConfig c = *cfg;  // only the geometry, the view and see_through_walls matter for an observation
    c.kind = MG_KIND_EMPTY;
    c.autoreset = MG_AUTORESET_NONE;
    c.num_rooms = 0;
    c.max_steps = 1;
    MgEnv h;  // on the stack: nothing of a handle is allocated
    int rc = configure(&h, c);
    if (rc != MG_OK) return rc;
    KParams kp = h.kp;
    const size_t hdr_bytes = (static_cast<size_t>(kp.E) * kp.HS + 255) / 256 * 256;
    const size_t need = hdr_bytes + static_cast<size_t>(kp.E) * kp.GS;
    if (need > MG_SCRATCH_BYTES)
        return fail(nullptr, MG_EINVAL,
                    "mg_gen_obs_of_state: %d environments of %dx%d cells and %d agents need %zu bytes of scratch memory, "
                    "the fixed buffer holds %zu (at most %zu such environments per call)",
                    kp.E, kp.W(), kp.H(), kp.A(), need, MG_SCRATCH_BYTES,
                    (MG_SCRATCH_BYTES - 256) / static_cast<size_t>(kp.HS + kp.GS));
    hipError_t e = hipSetDevice(c.device);
    if (e != hipSuccess) return fail(nullptr, MG_EHIP, "mg_gen_obs_of_state: hipSetDevice(%d): %s", c.device, hipGetErrorString(e));
    uint8_t *base = nullptr;
    rc = scratch_for_device(c.device, &base);
    if (rc != MG_OK) return rc;
    kp.hdr = base;
    kp.grid = base + hdr_bytes;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n = static_cast<size_t>(kp.E) * (kp.GS / 2);
    hipLaunchKernelGGL(mg_pack_state_kernel, dim3(static_cast<unsigned>(std::min<size_t>((n + 255) / 256, 65535))), dim3(256), 0,
                       st, kp, grid_dev, agents_dev);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, MG_EHIP, "mg_gen_obs_of_state: %s", hipGetErrorString(e));
    set_out(kp, out, c.view_size, h.force_stream);
    rc = launch_tile<MODE_OBS>(&h, kp, st);
    if (rc != MG_OK) fail(nullptr, rc, "mg_gen_obs_of_state: %s", h.err.c_str());
    return rc;
}

int mg_get_state(MgHandle h, uint8_t *grid_dev, uint8_t *agents_dev, uint8_t *meta_dev, int32_t *steps_dev,
                 uint8_t *flags_dev, void *stream) {
    if (!h) return MG_EINVAL;
    MG_HIP(h, hipSetDevice(h->cfg.device));
    const size_t n = static_cast<size_t>(h->kp.E) * (grid_dev ? h->kp.W() * h->kp.H() : 1);
    const int blocks = static_cast<int>(std::min<size_t>((n + 255) / 256, 65535));
    hipLaunchKernelGGL(mg_get_state_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), h->kp,
                       grid_dev, agents_dev, meta_dev, steps_dev, flags_dev);
    MG_HIP(h, hipGetLastError());
    return MG_OK;
}

int mg_set_state(MgHandle h, const uint8_t *grid_dev, const uint8_t *agents_dev, const uint8_t *meta_dev,
                 const int32_t *steps_dev, const uint8_t *flags_dev, void *stream) {
    if (!h) return MG_EINVAL;
    MG_HIP(h, hipSetDevice(h->cfg.device));
    const size_t n = static_cast<size_t>(h->kp.E) * h->kp.W() * h->kp.H();
    const int blocks = static_cast<int>(std::min<size_t>((n + 255) / 256, 65535));
    if (grid_dev || agents_dev) {
        // a state the packed formats cannot hold is refused as a whole (this waits for the stream)
        int *bad_dev = reinterpret_cast<int *>(h->stats + 64 * MG_NUM_STATS), bad = 0;
        hipStream_t st = static_cast<hipStream_t>(stream);
        MG_HIP(h, hipMemsetAsync(bad_dev, 0, sizeof(int), st));
        hipLaunchKernelGGL(mg_validate_state_kernel, dim3(blocks), dim3(256), 0, st, h->kp, grid_dev, agents_dev, bad_dev);
        MG_HIP(h, hipGetLastError());
        MG_HIP(h, hipMemcpyAsync(&bad, bad_dev, sizeof(int), hipMemcpyDeviceToHost, st));
        MG_HIP(h, hipStreamSynchronize(st));
        if (bad)
            return fail(h, MG_EINVAL, "mg_set_state: %d cells or agent rows exceed type<16, color<16, state<16, dir<4 or "
                        "name a position outside the grid; nothing was changed", bad);
    }
    hipLaunchKernelGGL(mg_set_state_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), h->kp,
                       grid_dev, agents_dev, meta_dev, steps_dev, flags_dev);
    MG_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(mg_refresh_front_kernel, dim3(std::min(4096, (h->kp.E * (h->kp.A() + 2) + 255) / 256)), dim3(256),
                       0, static_cast<hipStream_t>(stream), h->kp);
    MG_HIP(h, hipGetLastError());
    h->reset_done = true;
    if (grid_dev) h->kp.bits &= ~(1u << 20);  // grids of unknown shape: no assumption about their border
    return MG_OK;
}

MG_EXP(
int mg_dbg_buffer(MgHandle h, void *buf_dev, int *num_tiles) {
    h->kp.dbg_buf = static_cast<unsigned long long *>(buf_dev);
    if (num_tiles) *num_tiles = h->num_tiles;
    return MG_OK;
}
)

int mg_agents_mask(MgHandle h, uint8_t *mask_dev, uint8_t *done_dev, void *stream) {
    if (!h) return MG_EINVAL;
    if (!h->reset_done) return fail(h, MG_ESTATE, "mg_agents_mask: call mg_reset (or mg_set_state) first");
    MG_HIP(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(mg_agents_mask_kernel, dim3((h->kp.E + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       h->kp, mask_dev, done_dev);
    MG_HIP(h, hipGetLastError());
    return MG_OK;
}

int mg_episode_stats(int32_t num_envs, int32_t num_agents, const float *reward_dev, const uint8_t *terminated_dev,
                     const uint8_t *truncated_dev, double *return_acc_dev, int64_t *length_acc_dev, uint8_t *done_dev,
                     double *episode_return_dev, int64_t *episode_length_dev, void *stream) {
    if (num_envs < 1 || num_agents < 1 || num_agents > 256)
        return fail(nullptr, MG_EINVAL, "mg_episode_stats: num_envs must be positive and num_agents 1..256");
    if (!reward_dev || !terminated_dev || !truncated_dev || !return_acc_dev || !length_acc_dev || !done_dev ||
        !episode_return_dev || !episode_length_dev)
        return fail(nullptr, MG_EINVAL, "mg_episode_stats: null argument");
    const int per_block = 256 / num_agents;
    hipLaunchKernelGGL(mg_episode_stats_kernel, dim3((num_envs + per_block - 1) / per_block), dim3(256), 0, static_cast<hipStream_t>(stream),
                       num_envs, num_agents, reward_dev, terminated_dev, truncated_dev, return_acc_dev, length_acc_dev,
                       done_dev, episode_return_dev, episode_length_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, MG_EHIP, "mg_episode_stats: %s", hipGetErrorString(e));
    return MG_OK;
}

int mg_launch_census(char *buf, int32_t size) {
    if (!buf || size < 1) return MG_EINVAL;
    std::string out;
    for (const LaunchCount *c = g_launch_counts.load(std::memory_order_acquire); c; c = c->next) {
        char line[96];
        snprintf(line, sizeof line, "%s:%ld\n", c->what, c->n.load(std::memory_order_relaxed));
        out += line;
    }
    if (static_cast<int>(out.size()) + 1 > size) return MG_EINVAL;
    memcpy(buf, out.c_str(), out.size() + 1);
    return MG_OK;
}

int mg_stats(MgHandle h, int64_t *stats_dev, void *stream) {
    if (!h) return MG_EINVAL;
    if (!stats_dev) return fail(h, MG_EINVAL, "mg_stats: stats_dev is null");
    MG_HIP(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(mg_stats_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), h->stats, stats_dev,
                       h->launches);
    MG_HIP(h, hipGetLastError());
    return MG_OK;
}

}  // extern "C"
