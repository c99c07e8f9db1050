/*
 * mgrid.h - C ABI of libmgrid_hip.so: a batched MultiGrid simulator for AMD MI355X (gfx950).
 *
 * The reference (heng2j/multigrid) has no FFI seam on this path: its boundary is the Python
 * class surface of `MultiGridEnv` (multigrid/base.py:37).  Each entry point below names the
 * reference method it replaces for a whole batch of independent environments.  All pointers
 * suffixed `_dev` are device pointers owned by the caller (e.g. PyTorch-ROCm tensors); the
 * library owns only its internal state buffers.  Every launch is asynchronous on the
 * `stream` argument (a `hipStream_t` passed as `void*`; NULL = the default stream).
 *
 * Every function returns 0 on success or a negative MG_E* code and never throws;
 * `mg_last_error` describes the most recent failure.  Calls on one handle must be
 * serialised by the caller.
 *
 * Array layouts at the boundary are the reference's own:
 *   grid     uint8 [E][W][H][3]  (type, color, state) per cell, x-major  multigrid/core/grid.py:53
 *   agents   uint8 [E][A][8]     x y dir terminated carry_type carry_color carry_state color
 *                                (the AgentState row of multigrid/core/agent.py:266-275 without
 *                                its constant type column, position first)
 *   meta     uint8 [E][4]        RedBlueDoors: red_x red_y blue_x blue_y (multigrid/envs/
 *                                redbluedoors.py:161-168); LockedHallway with more than six rooms:
 *                                meta[0..1] = doors already rewarded (bit 2 * row + side), meta[3] = number
 *                                of distinct door colours (mg_upload_layouts counts them itself);
 *                                zeros for the other kinds
 *   image    uint8 [E][A][V][V][3]  obs["image"] of multigrid/base.py:430-436 (int64 there)
 *   onehot   uint8 [E][A][V][V][24] the same image one-hot encoded, multigrid/wrappers.py:114-146
 * Cell fields (and the carried object and colour of an agent row) are limited to type < 16, state < 16,
 * color < 16 (the reference uses 11 / 7 / 6); mg_upload_layouts and mg_set_state refuse anything larger.
 */
#ifndef MGRID_H
#define MGRID_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_ABI_VERSION 8

/* error codes */
#define MG_OK 0
#define MG_EINVAL (-1)  /* bad argument / unsupported configuration */
#define MG_EHIP (-2)    /* a HIP runtime call failed */
#define MG_ESTATE (-3)  /* call order violated (e.g. step before layouts were uploaded) */

/* environment kinds (which post-step hook runs) */
#define MG_KIND_EMPTY 0               /* multigrid/envs/empty.py            - no hook */
#define MG_KIND_REDBLUEDOORS 1        /* multigrid/envs/redbluedoors.py:170-187 */
#define MG_KIND_BLOCKEDUNLOCKPICKUP 2 /* multigrid/envs/blockedunlockpickup.py:160-169 */
#define MG_KIND_LOCKEDHALLWAY 3       /* multigrid/envs/locked_hallway.py:193-217 */
#define MG_KIND_COMPETITIVE 4         /* CompetativeRedBlueDoorEnvV3 with its default step handlers,
                                         multigrid/envs/competative_red_blue_door.py:372-536: colour-restricted pickup
                                         (:573-577), team win at the own door (:484-495), death-match eliminations
                                         (:498-523), pickup rewards and penalty (:425-472); at most 8 agents; agents
                                         act in `order_dev` order = the order of the caller's actions dict (:633-641),
                                         index order when order_dev is NULL; the step's info dict comes back through
                                         MgOut.info_dev, which also says whose pickup was a rewarded key or ball
                                         pickup (what the mission texts are rewritten from, :438-468).  Not
                                         reproduced: policy objects (`policies_map`). */

/* what happens to an environment whose episode ended in this step */
#define MG_AUTORESET_NONE 0      /* nothing: the caller resets with mg_reset (reference behaviour) */
#define MG_AUTORESET_SAME_STEP 1 /* state and the returned image/direction are those of the next
                                    episode's start, reward/terminated/truncated those of the step
                                    that ended (gymnasium SyncVectorEnv as used at
                                    multigrid/scripts/train_ppo_cleanrl.py:366,517) */
#define MG_AUTORESET_NEXT_STEP 2 /* terminal image returned, state reset before the next step */

typedef struct MgConfig {
    int32_t kind;                /* MG_KIND_* */
    int32_t width, height;       /* grid size W, H                      base.py:162-165 */
    int32_t num_agents;          /* A                                    base.py:172-178 */
    int32_t view_size;           /* V: 3, 5, 7, 9 or 11 (the reference takes any odd size; its default is 5, its
                                    registry entries never set it); MG_EINVAL otherwise.  9 and 11 run the
                                    general forms of the step kernel only (mg_step_n then queues single
                                    steps)                                                   base.py:96 */
    int32_t max_steps;           /*                                      base.py:94 */
    int32_t see_through_walls;   /*                                      base.py:95 */
    int32_t allow_agent_overlap; /*                                      base.py:97 */
    int32_t joint_reward;        /*                                      base.py:98 */
    int32_t success_any;         /* success_termination_mode == "any"    base.py:99 */
    int32_t failure_any;         /* failure_termination_mode == "any"    base.py:100 */
    int32_t num_rooms;           /* LockedHallway only, 1..14.  Up to six rooms every door has a colour of its
                                    own and rewarded doors are tracked by colour; beyond that the reference
                                    cycles the six colours, doors of one colour are separate doors, tracked by
                                    room (which takes its room-grid geometry: width = 3 (room_size - 1) + 1,
                                    height = num_rooms / 2 * (room_size - 1) + 1), and the all-doors ending
                                    comes at the number of distinct colours  locked_hallway.py:129,150-166,213 */
    int32_t num_envs;            /* E: environments held by this handle */
    int32_t device;              /* HIP device ordinal */
    int32_t autoreset;           /* MG_AUTORESET_* */
    int32_t env_id_offset;       /* global index of environment 0 (sharded runs); it seeds the
                                    layout choice and the on-device agent-order generator */
    int32_t death_match;         /* MG_KIND_COMPETITIVE: toggling an opponent eliminates it
                                    (competative_red_blue_door.py:124,498); needs the 16 x 8 grid */
    int32_t per_agent_outputs;   /* MG_KIND_COMPETITIVE: training_scheme "DTDE" / "CTDE" (dtde_step, :372-423,
                                    which reports no termination while only some agents are terminated,
                                    :375-378) rather than "CTCE" (ctce_step, :319-370; sum the rewards and AND
                                    the flags per team on the caller's side) */
    uint64_t seed;               /* seed of the on-device agent-order generator */
    int32_t batch_envs;          /* 0, or the size of the whole batch of which this handle holds one slice
                                    (mg_step_shards): the launch is then shaped as for that batch.  It changes
                                    no result. */
    int32_t reserved0;           /* must be 0 */
} MgConfig;

/* Caller-owned device output buffers of one step / reset.  Any pointer may be NULL to skip it. */
typedef struct MgOut {
    uint8_t *image_dev;      /* [E][A][V][V][3] */
    uint8_t *direction_dev;  /* [E][A]  obs["direction"]                 base.py:434 */
    float *reward_dev;       /* [E][A]                                   base.py:454,560-562 */
    uint8_t *terminated_dev; /* [E][A]                                   base.py:400 + hooks */
    uint8_t *truncated_dev;  /* [E][A]                                   base.py:401-402 */
    uint8_t *onehot_dev;     /* [E][A][V][V][24] the image as OneHotObsWrapper.observation returns
                                it (multigrid/wrappers.py:105-146: 11 type, 6 colour and 7 state
                                planes, wrappers.py:95), written by the same launch; 8-byte aligned */
    uint8_t *rows_dev;       /* [E][A][8] agent rows (layout at the top of this file) of the state the returned
                                observation shows: what `agent_states` reads after the call (base.py:587-596 `is_done`,
                                pettingzoo/__init__.py:48-52 `agents`); 8-byte aligned.  Single-step launches only:
                                with it mg_step_n takes one launch per step. */
    double *reward64_dev;    /* [E][A] the step's rewards as the doubles `_reward` computes (base.py:655-659), where
                                reward_dev holds them rounded to float32; 8-byte aligned; as rows_dev for mg_step_n.
                                Written by steps only. */
    uint8_t *info_dev;       /* [E][A] MG_KIND_COMPETITIVE steps only: the info dict of the step per agent
                                (competative_red_blue_door.py:380-388), packed: bit 0 door_open_done (:490-492), bit 1
                                got_eliminated_done (:502), bit 2 eliminated_opponents_done and bits 4-5
                                eliminated_opponent_num (:517-522; at most one per agent and step).  Under "CTCE" the
                                reference keys the dict by team (:320-328): or the flags and sum the counts over a
                                team's agents.  Bit 6 / bit 7: the agent's pickup in this step was a rewarded key /
                                ball pickup (:428-436 / :445-454), after which the reference rewrites the mission
                                texts of the agent and its team mates (:438-443 / :456-468).  Not touched for the
                                other kinds (their info dicts are empty, base.py:408). */
} MgOut;

typedef struct MgEnv *MgHandle;

int mg_abi_version(void);

/* Replaces MultiGridEnv.__init__ (multigrid/base.py:87-245) for E environments. */
int mg_create(const MgConfig *cfg, MgHandle *out);
int mg_destroy(MgHandle h);

/* Text of the last error on this handle (or of the last failed mg_create when h is NULL). */
const char *mg_last_error(MgHandle h);

/* Uploads N start states produced on the host by the `_gen_grid` layout generators
 * (multigrid/envs/empty.py:152, redbluedoors.py:141, blockedunlockpickup.py:136,
 * locked_hallway.py:146).  HOST pointers; copied, synchronous.  `meta` may be NULL except for
 * MG_KIND_REDBLUEDOORS, whose hook needs the door positions: there it must name a red door and a blue door
 * of the layout's grid (MG_EINVAL otherwise). */
int mg_upload_layouts(MgHandle h, const uint8_t *grids, const uint8_t *agents, const uint8_t *meta,
                      int32_t n);

/* MG_KIND_COMPETITIVE: the agents' reward schemes (`reward_schemes[agent.name]`, multigrid/envs/__init__.py:100-108),
 * HOST pointer to double [A][4]: eliminated_opponent_sparse_reward, key_pickup_sparse_reward,
 * ball_pickup_dense_reward, invalid_pickup_dense_penalty.  All zero until set.  Synchronous. */
int mg_set_reward_schemes(MgHandle h, const double *schemes);

/* Parameters of the `_gen_grid` procedures that MgConfig does not carry. */
typedef struct MgLayoutSpec {
    int32_t size;              /* Empty, RedBlueDoors: `size` (empty.py:112, redbluedoors.py:103);
                                  BlockedUnlockPickup, LockedHallway: `room_size` (blockedunlockpickup.py:105,
                                  locked_hallway.py:100) */
    int32_t random_start;      /* Empty: agent_start_pos is None, agents placed at random (empty.py:166-171) */
    int32_t start_x, start_y, start_dir; /* Empty: the fixed start otherwise (empty.py:113-114) */
    int32_t max_hallway_keys;  /* LockedHallway (locked_hallway.py:101) */
    int32_t max_keys_per_room; /* LockedHallway (locked_hallway.py:102) */
    uint8_t agent_color[64];   /* colour index of agent k = its team's colour (base.py:185-203) */
} MgLayoutSpec;

/* Replaces the host loop "reset() N times, then mg_upload_layouts": draws `n` start states on the
 * device with the environment class's own `_gen_grid` procedure (multigrid/envs/empty.py:152,
 * redbluedoors.py:141, blockedunlockpickup.py:136, locked_hallway.py:146; place_obj / place_agent
 * multigrid/base.py:661-749) and makes them the handle's layout pool.  Layout l depends only on
 * (seed, l).  The random numbers are a counter generator's, not numpy's: the layouts follow the
 * reference's distribution, not its sequences.  Synchronous; MG_EINVAL when the spec does not match the
 * handle's grid or a room cannot hold what has to be placed in it. */
int mg_generate_layouts(MgHandle h, const MgLayoutSpec *spec, int32_t n, uint64_t seed);

/* Replaces MultiGridEnv.reset (multigrid/base.py:310-362).  `layout_idx_dev` int32[E] picks the
 * start state per environment (NULL: environment e takes layout (env_id_offset + e +
 * 7919 * episode) mod N).  `mask_dev` uint8[E] restricts the reset to environments whose byte is
 * non-zero (NULL: all).  image/direction of the reset environments are written; reward,
 * terminated and truncated are not touched. */
int mg_reset(MgHandle h, const int32_t *layout_idx_dev, const uint8_t *mask_dev, const MgOut *out,
             void *stream);

/* Replaces MultiGridEnv.step (multigrid/base.py:364-408) plus the environment's own step hook.
 * `actions_dev` uint8[E][A] (multigrid/core/actions.py:4-15).  `order_dev` uint8[E][A] is the
 * order in which agents act, i.e. the argsort drawn at multigrid/base.py:457-461; NULL draws a
 * uniform random permutation per environment and step on the device. */
int mg_step(MgHandle h, const uint8_t *actions_dev, const uint8_t *order_dev, const MgOut *out,
            void *stream);

/* mg_step of ONE batch that is held as `num_shards` handles, each on a stream of its own (`streams[k]`, NULL =
 * all on the default stream).  The handles hold consecutive slices of the batch (handle k starts at environment
 * E_0 + ... + E_{k-1}; its env_id_offset must say so, its batch_envs should name the whole batch), all on one
 * device with the same num_agents and view_size; `actions_dev`, `order_dev` and the pointers of `out` are the
 * WHOLE batch's arrays, laid out as for mg_step, and every shard reads and writes its own slice of them.  The
 * results are those of one handle holding the whole batch (MultiGridEnv.step, multigrid/base.py:364-408, per
 * environment; the reference runs groups of environments side by side as rollout workers,
 * multigrid/utils/training_utilis.py:273 `num_rollout_workers`, and as a vector of environments,
 * multigrid/scripts/train_ppo_cleanrl.py:366).
 * Why: environments are independent, but launches on one stream are not - step t + 1 of every environment
 * waits for the slowest wavefront of step t.  Shards on separate streams wait only for themselves, so the
 * start and the drain of one shard's launch overlap the other's (two shards: 16 % less time per step of
 * 131072 x 8 agents on MI355X).  The caller orders each stream against the producers of its inputs and the
 * consumers of its outputs (events); nothing here synchronises.  Every shard is checked as mg_step checks
 * it before the first one is launched. */
int mg_step_shards(const MgHandle *shards, int32_t num_shards, const uint8_t *actions_dev, const uint8_t *order_dev,
                   const MgOut *out, void *const *streams);

/* `num_steps` consecutive steps by one call; for num_steps > 1 and no one-hot output they are taken by ONE kernel
 * launch that keeps the environments' state in on-chip memory between the steps (same results as
 * `num_steps` mg_step calls, less memory traffic and no launch gaps).  It stands for `num_steps`
 * consecutive mg_step calls (a rollout whose actions are known up
 * front, e.g. the random-action loop of multigrid/scripts/train_ppo_cleanrl.py:517-529 with the policy
 * replaced by a table).  Every array gains a leading [num_steps] dimension: actions_dev
 * uint8[T][E][A], order_dev uint8[T][E][A] or NULL, and each non-NULL pointer of `out`
 * ([T][E][A][V][V][3] images, [T][E][A][V][V][24] one-hot images, [T][E][A] scalars); step t reads slice t and writes slice t.  Stops at
 * the first step that fails and returns its code; steps before it have been queued. */
int mg_step_n(MgHandle h, int32_t num_steps, const uint8_t *actions_dev, const uint8_t *order_dev,
              const MgOut *out, void *stream);

/* Replaces MultiGridEnv.gen_obs (multigrid/base.py:410-438): image and direction of the current
 * state of every environment. */
int mg_gen_obs(MgHandle h, const MgOut *out, void *stream);

/* mg_gen_obs for a state that the CALLER holds, in the reference's layouts (grid_dev uint8 [E][W][H][3],
 * agents_dev uint8 [E][A][8], device pointers), instead of the state inside a handle; `cfg` gives num_envs,
 * width, height, num_agents, view_size, see_through_walls and device (the other fields are ignored), `out`
 * takes image / direction / onehot as in mg_gen_obs.
 * This entry point has no counterpart in the reference.  It exists for callers that cannot allocate device
 * memory: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It needs no
 * handle and calls no hipMalloc / hipFree: the packed copy of the state that the observation kernel reads
 * lives in a fixed 8 MiB buffer inside the library (one per GPU, held until the process ends).  A call
 * whose environments do not fit (num_envs * (header + grid bytes), e.g. 13442 environments of 16 x 16 cells
 * and 8 agents, 47661 of 8 x 8 cells and 2 agents) returns MG_EINVAL and mg_last_error(NULL) says how many
 * do.  Fields are taken modulo 16 (heading modulo 4) instead of being checked; an agent outside the grid is
 * neither drawn nor does it block sight, and its own view is taken from the nearest cell inside.  All calls on one GPU share
 * the buffer: calls that may overlap there (different streams, different threads) race and must be
 * serialised by the caller; calls on one stream are safe back to back. */
int mg_gen_obs_of_state(const MgConfig *cfg, const uint8_t *grid_dev, const uint8_t *agents_dev, const MgOut *out,
                        void *stream);

/* Debug / parity access to the full state in the reference's layouts (see the top of this file).
 * steps int32[E] = step_count (base.py:350,395); flags uint8[E]: bits 0-5 LockedHallway doors
 * already rewarded, by door colour (locked_hallway.py:204-205; with more than six rooms they are in meta); bit 7: RedBlueDoors blue door
 * object re-closed after a failure while its grid cell still reads open (redbluedoors.py:185).
 * MG_KIND_COMPETITIVE keeps there what the reference keeps in objects: flags bit 0 / 1 = the red / blue key
 * object is "picked up" (competative_red_blue_door.py:434-435) although the agent row still shows the copy
 * taken at pickup time; bit 2 / 3 = the red / blue door's cell was cleared under an eliminated agent (:503);
 * meta[0] / meta[1] = how often the red / blue ball's discount factor has been squared (:452-454).
 * Any pointer may be NULL.  mg_set_state first checks `grid_dev` and `agents_dev` against the field limits
 * at the top of this file and every agent's position against the grid (x < W, y < H), waiting for the stream,
 * and returns MG_EINVAL, leaving the state untouched, when a value does not fit. */
int mg_get_state(MgHandle h, uint8_t *grid_dev, uint8_t *agents_dev, uint8_t *meta_dev,
                 int32_t *steps_dev, uint8_t *flags_dev, void *stream);
int mg_set_state(MgHandle h, const uint8_t *grid_dev, const uint8_t *agents_dev,
                 const uint8_t *meta_dev, const int32_t *steps_dev, const uint8_t *flags_dev,
                 void *stream);

/* Episode statistics accumulated on the device since creation, copied to a DEVICE buffer of
 * MG_NUM_STATS int64: [0] finished episodes, [1] sum of their lengths, [2] episodes ended by
 * termination, [3] by truncation, [4] sum of all rewards paid, in units of 2^-24, [5] steps
 * taken (launches), [6..7] reserved.  (The reference keeps no such counters; its callers sum
 * them on the host, multigrid/utils/training_utilis.py:291-367.) */
#define MG_NUM_STATS 8
int mg_stats(MgHandle h, int64_t *stats_dev, void *stream);

/* Replaces PettingZooWrapper.agents (multigrid/pettingzoo/__init__.py:48-52) and MultiGridEnv.is_done
 * (multigrid/base.py:587-596) for every environment, by one launch: done_dev uint8[E] = the step count has reached
 * max_steps or every agent is terminated (the environment's own flags, agent_states.terminated, not the returned
 * dict); mask_dev uint8[E][A] = agent k is listed in `agents`, i.e. it is not terminated and the environment is not
 * done.  Either pointer may be NULL. */
int mg_agents_mask(MgHandle h, uint8_t *mask_dev, uint8_t *done_dev, void *stream);

/* Diagnostic: which instantiations of the step / reset / gen_obs kernel this process has launched so far, as text in
 * `buf` (HOST memory, `size` bytes, MG_EINVAL when too small): one line "V,SEE,MODE,WALLED,OH,MULTI,COMP,CROWD,STG:count"
 * per instantiation - view size, see_through_walls, mode (0 step, 1 gen_obs, 2 reset), grids known to be ringed by
 * walls, one-hot output, several steps per launch, CompetativeRedBlueDoor, cell-table act phase, staging width.
 * No counterpart in the reference; the tests use it to prove that every compiled instantiation has been run against
 * the oracle. */
int mg_launch_census(char *buf, int32_t size);

/* Per-environment episode statistics of a vector of environments after a step with MG_AUTORESET_SAME_STEP, by one
 * launch: replaces gym.wrappers.RecordEpisodeStatistics around every environment of the reference's SyncVectorEnv
 * (multigrid/scripts/train_ppo_cleanrl.py:173,366) and the `final_info` entries its training loop reads
 * (:418-427, :522-527).  reward / terminated / truncated are the step's outputs ([E][A]); return_acc double[E][A] and
 * length_acc int64[E] are the caller's running counters (in and out, zero at reset).  An environment's episode is
 * over when all its agents are terminated or it is truncated: done[e] = 1, episode_return[e][*] and
 * episode_length[e] take the counters (this step included) and the counters start again at zero; elsewhere done and
 * the episode outputs are zero.  All pointers are device pointers on the stream's device; needs no handle. */
int mg_episode_stats(int32_t num_envs, int32_t num_agents, const float *reward_dev, const uint8_t *terminated_dev,
                     const uint8_t *truncated_dev, double *return_acc_dev, int64_t *length_acc_dev, uint8_t *done_dev,
                     double *episode_return_dev, int64_t *episode_length_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MGRID_H */
