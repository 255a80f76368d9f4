/* so100_sim.h -- C ABI of libso100sim.so: the MI355X-native batched so100 simulator.
 *
 * This is the drop-in boundary for the ONE hot path of PieterBecking/so100-mujoco-rl: the per-env
 * step behind its Gymnasium / Stable-Baselines3 surface.  Nothing comparable exists in the reference
 * (its Python <-> C crossing is pybind11 `mujoco.mj_step(MjModel, MjData, nstep)`: opaque structs,
 * fp64, one env); each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch types.  All `*_dev` pointers are DEVICE pointers
 *     owned by the caller (e.g. tensor.data_ptr()); persistent sim state is owned by the handle.
 *   - every call returns 0 on success, a negative SO100_E_* code otherwise; the message is available
 *     from so100_last_error() (thread local).  Nothing throws across the ABI.
 *   - all work is enqueued on the caller's `hip_stream` (a hipStream_t; NULL = default stream);
 *     so100_step / so100_reset allocate nothing, free nothing and never synchronise
 *     (hipGraph-capturable).
 *   - one handle per process per device; a handle is not thread safe.
 *   - there is NO CPU fallback: without a usable HIP device so100_create fails.
 *
 * Reference files cited as "ref:" live under /root/reference/src/so100_mujoco_rl/.
 */
#ifndef SO100_SIM_H
#define SO100_SIM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SO100_ABI_VERSION 3

/* env kinds == the reference's registered ids Env01-v1 .. Env06-v1 (ref: __init__.py:5-45) */
#define SO100_ENV01 1   /* ref: envs/env01_v1.py  reach, random start pose              obs 15 */
#define SO100_ENV02 2   /* ref: envs/env02_v1.py  reach + re-randomise on reach         obs 15 */
#define SO100_ENV03 3   /* ref: envs/env03_v1.py  look-at, moving cube (bbox detector)    obs 8 */
#define SO100_ENV04 4   /* ref: envs/env04_v1.py  look-at, jumping cube (bbox detector)   obs 8 */
#define SO100_ENV05 5   /* ref: envs/env05_v1.py  look-at, analytic reprojection + noise  obs 8 */
#define SO100_ENV06 6   /* ref: envs/env06_v1.py  reach + close the gripper (env_base_06.py) obs 15 */

/* physics option flags (which MuJoCo constraint families are simulated) */
#define SO100_F_FRICTIONLOSS 1u   /* joint friction loss 0.1 (model/so_arm100_camera.xml:32)        */
#define SO100_F_LIMITS       2u   /* joint range limits (model/so_arm100_camera.xml:35-50)           */
#define SO100_F_FLOOR        4u   /* cube / floor box-plane contact (model/env01.xml:32,39)          */
#define SO100_F_CUBE_PINNED  8u   /* cube kinematic (BASELINE.json configs[1]: "contact disabled")   */
#define SO100_F_PADS_FLOOR  16u   /* the 8 finger-pad boxes vs the floor plane (so_arm100_camera.xml:108-111,120-123; env01.xml:39):
                                     live in the reference scene                                                           */
#define SO100_F_PADS_CUBE   32u   /* the finger pads vs the cube: excluded in the reference scene (env01.xml:47-48); lifting
                                     that exclusion is BASELINE.json configs[4] ("arm-cube collision")                    */
#define SO100_F_LINKS_FLOOR 64u   /* capsule proxies of the arm links' collision MESHES (so_arm100_camera.xml:58-59, 85, 92, 99, 106-107, 117-119: the STL files
                                     are not in the reference snapshot) vs the floor: one capsule per link 1..5, built by rule from the link frames and
                                     inertial boxes (csrc/so100_model_def.h).  A documented STAND-IN, not reference geometry; off in SO100_F_REFERENCE */
#define SO100_F_LINKS_CUBE 128u   /* the same kind of capsule on Rotation_Pitch and Upper_Arm vs the cube: the two arm bodies the reference scene does NOT exclude from
                                     colliding with block_a (env01.xml:44-48).  Stand-in geometry and a stand-in capsule-box narrowphase
                                     (oracle/so100_oracle.c: so100o_capsule_box); needs a dynamic cube; off in SO100_F_REFERENCE                        */
/* what the reference scene simulates, as far as it can be reproduced: the arm's mesh geoms (class "collision",
 * so_arm100_camera.xml:58-59) are not in the reference snapshot, so link-vs-floor / link-vs-link mesh contacts are absent */
#define SO100_F_REFERENCE (SO100_F_FRICTIONLOSS | SO100_F_LIMITS | SO100_F_FLOOR | SO100_F_PADS_FLOOR)
#define SO100_F_NOPADS    (SO100_F_FRICTIONLOSS | SO100_F_LIMITS | SO100_F_FLOOR)   /* round-1 "reference": no arm contact at all */
#define SO100_F_CONTACT5  (SO100_F_REFERENCE | SO100_F_PADS_CUBE)                   /* BASELINE.json configs[4]                   */
#define SO100_F_REFERENCE_LINKS (SO100_F_REFERENCE | SO100_F_LINKS_FLOOR)           /* + the link proxies: no arm link passes through the table */
#define SO100_F_REFERENCE_PROXIES (SO100_F_REFERENCE_LINKS | SO100_F_LINKS_CUBE)    /* + Rotation_Pitch / Upper_Arm vs the cube: every contact pair of the reference scene
                                                                                        that does not need two mesh shapes (link vs link stays out) */

#define SO100_E_INVALID  (-1)     /* bad argument                                   */
#define SO100_E_NODEVICE (-2)     /* no usable HIP device / HIP runtime failure      */
#define SO100_E_NOMEM    (-3)
#define SO100_E_LAUNCH   (-4)

#define SO100_NINJECT 16          /* per env: [0..7] step-phase uniforms, [8..15] reset-phase */

typedef struct so100_sim so100_sim;

typedef struct {
    int32_t  env_kind;            /* SO100_ENV01..06                                               */
    int32_t  num_envs;            /* N, any positive number                                        */
    int32_t  device;              /* HIP device ordinal                                            */
    uint32_t flags;               /* SO100_F_*                                                     */
    int32_t  solver_iters;        /* block-PGS sweeps over the arm rows (>= 1).  Measured in fp64 against the converged
                                     optimum: 2 sweeps 2e-8, 3 sweeps 5e-10, 4 sweeps 4e-12 => 2 is below fp32 round-off */
    int32_t  contact_iters;       /* max Newton iterations of the cube/floor block (6: 1 when resting or settling, more on impacts) */
    int32_t  frame_skip;          /* physics substeps per env step; 16 (ref: envs/env_base_01.py:45) */
    int32_t  max_episode_steps;   /* TimeLimit: 4000 Env01, 6000 others (ref: __init__.py:8,15); 0 = none */
    uint64_t seed;                /* Philox key                                                    */
    uint32_t env_id_offset;       /* global id of env 0 (rank * N): makes sharded runs reproducible */
    uint32_t envs_per_workgroup;  /* 0 = chosen by the library from N and the CU count (16 / 32 / 64 lanes of each 64-lane wave own an env in the
                                     multi-wave kernels; the pad-contact solve sums its records over the 4 / 2 / 1 lanes of an env in that
                                     order).  Results are bitwise reproducible for a given value; two handles that must agree bit for bit on
                                     the same envs although their N differs (a batch split into shards of another size) pass the same value:
                                     16, 32 or 64.  so100_envs_per_workgroup() reads it back. */
} so100_config;

/* One vectorised env step.  Replaces, for N envs at once, the reference chain
 *   SB3 DummyVecEnv.step_wait -> gymnasium TimeLimit.step -> EnvNN.step (ref: envs/env01_v1.py:15-37 and
 *   clones) -> mujoco.mj_step(model, data, nstep=16) -> _get_obs, including DummyVecEnv's auto-reset. */
/* Non-finite guard (product behaviour; the reference has none and MuJoCo would mj_resetData with a warning): if an env's
 * state, observation or reward is NaN / inf / beyond 1e10 after a step (e.g. a NaN action), that env's episode ends --
 * done = 1, trunc = 0, reward 0, terminal observation 0 -- it is auto-reset like any other finished episode and bit 128
 * of its "bits" state row is latched.  Other envs of the batch are unaffected. */
/* "_dev" pointers of THIS struct must be readable / writable by the GPU: device memory, or PINNED HOST memory (hipHostMalloc /
 * torch pin_memory: mapped into the GPU's address space) -- then the kernel moves the actions / results over the host link itself
 * and a numpy caller needs no copies at all (So100VecEnv's numpy path). */
typedef struct {
    const float* act_dev;          /* [N][6]  f32 in [-1,1] (ref: envs/env_base_01.py:77-83)          */
    float*       obs_dev;          /* [N][obs_dim] f32; post-auto-reset observation where done        */
    float*       rew_dev;          /* [N] f32                                                         */
    uint8_t*     done_dev;         /* [N] terminated || truncated                                     */
    uint8_t*     trunc_dev;        /* [N] info["TimeLimit.truncated"] = truncated && !terminated      */
    float*       terminal_obs_dev; /* [N][obs_dim] info["terminal_observation"], written where done; nullable */
    float*       ep_return_dev;    /* [N] Monitor's info["episode"]["r"], written where done; nullable */
    int32_t*     ep_length_dev;    /* [N] Monitor's info["episode"]["l"], written where done; nullable */
    const float* inject_dev;       /* [N][SO100_NINJECT] uniforms replacing the device RNG (parity tests); nullable */
    float*       rollout_row_dev;  /* [N][obs_dim+10] row of a rollout buffer (layout: so100_policy_io); the step writes
                                      reward -> column obs_dim+6 and done -> column obs_dim+7 (0 = running, 1 = terminated,
                                      2 = TimeLimit-truncated only: what SB3's collect_rollouts bootstraps); nullable */
} so100_step_io;

/* ---- rollout-side helper (caller of the hot path) ------------------------------------------
 * Fused SB3 "MlpPolicy" forward for a Box action space (ref: main.py:56-64 -> stable_baselines3 PPO("MlpPolicy"):
 * separate 2x64 tanh towers for pi and V, state-independent log_std), Gaussian sampling, log-prob, clip to [-1,1]
 * and the rollout-buffer write, in one launch.  Weights are PyTorch nn.Linear tensors (weight[out][in], row-major),
 * i.e. the policy's state_dict entries named in the comments. */
typedef struct {
    const float *pi_w0, *pi_b0;    /* mlp_extractor.policy_net.0.{weight[64][obs_dim], bias[64]} */
    const float *pi_w1, *pi_b1;    /* mlp_extractor.policy_net.2.{weight[64][64], bias[64]}      */
    const float *mu_w, *mu_b;      /* action_net.{weight[6][64], bias[6]}                        */
    const float *log_std;          /* log_std[6]                                                 */
    const float *vf_w0, *vf_b0;    /* mlp_extractor.value_net.0.{weight[64][obs_dim], bias[64]}  */
    const float *vf_w1, *vf_b1;    /* mlp_extractor.value_net.2.{weight[64][64], bias[64]}       */
    const float *v_w, *v_b;        /* value_net.{weight[1][64], bias[1]}                         */
} so100_policy_weights;

typedef struct {
    const float* obs_dev;          /* [N][obs_dim]                                                         */
    const float* noise_dev;        /* [N][6] standard normals replacing the device RNG; nullable            */
    float*       act_env_dev;      /* [N][6] clipped to [-1,1] (what so100_step consumes)                   */
    float*       act_raw_dev;      /* [N][6] unclipped sample (what SB3 stores); nullable                   */
    float*       value_dev;        /* [N]; nullable                                                         */
    float*       logp_dev;         /* [N]; nullable                                                         */
    float*       rollout_row_dev;  /* [N][obs_dim+10] = obs | raw action(6) | reward | done (0/1/2) | value | logp; nullable */
} so100_policy_io;

int  so100_policy_forward(so100_sim* sim, const so100_policy_weights* w, const so100_policy_io* io,
                          uint32_t step_counter, void* hip_stream);

/* T vectorised steps of {policy forward, sample, clip, env step, rollout-buffer write} in ONE launch (persistent
 * workgroups: env state stays in registers, weights in LDS; csrc/so100_rollout.hpp).  Equivalent, step for step, to
 * calling so100_policy_forward(step_counter0 + t) + so100_step T times.  ref: the loop body of stable_baselines3
 * OnPolicyAlgorithm.collect_rollouts driven from main.py:234-238. */
typedef struct {
    float*   rollout_dev;          /* [T][N][obs_dim+10] rows: obs | raw action(6) | reward | done (0/1/2) | value | logp */
    float*   obs_dev;              /* [N][obs_dim] in: current observation; out: observation after the last step     */
    float*   rew_dev;              /* [N] last step's reward                                                        */
    uint8_t* done_dev;             /* [N] last step's done                                                          */
    uint8_t* trunc_dev;            /* [N] last step's TimeLimit.truncated                                           */
    float*   terminal_obs_dev;     /* [N][obs_dim] written where an episode ended (latest); nullable                 */
    float*   ep_return_dev;        /* [N] nullable                                                                  */
    int32_t* ep_length_dev;        /* [N] nullable                                                                  */
    float*   terminal_obs_chunk_dev; /* [T][N][obs_dim] info["terminal_observation"] of EVERY episode end inside the chunk,
                                      written where done != 0 (other entries untouched): what the learner needs to bootstrap
                                      gamma * V(terminal_observation) on truncated steps; nullable                     */
} so100_rollout_io;

int  so100_rollout(so100_sim* sim, const so100_policy_weights* w, const so100_rollout_io* io, int32_t T,
                   uint32_t step_counter0, void* hip_stream);

int  so100_abi_version(void);
int  so100_obs_dim(int32_t env_kind);                     /* ref: env_base_01.py:63-75 (15), env_base_02.py:56-69 (8) */
int  so100_num_state_fields(void);                        /* rows of the [field][N] state matrix */
int  so100_state_field_index(const char* name);           /* e.g. "q0", "elapsed_steps"; -1 if unknown */
const char* so100_state_field_name(int32_t field);        /* inverse of the above; NULL if out of range (used by the
                                                             state save/restore wire format) */

/* ref: gym.make(id) -> EnvNN.__init__ -> mujoco.MjModel.from_xml_path + MujocoEnv.__init__
 * (envs/env_base_01.py:35-51).  The model is compiled in (csrc/so100_model_def.h). */
int  so100_create(const so100_config* cfg, so100_sim** out);
void so100_destroy(so100_sim* sim);
int  so100_envs_per_workgroup(const so100_sim* sim);      /* the value in use (see so100_config); < 0 on a null handle */

/* ref: MujocoEnv.reset -> mj_resetData -> EnvNN.reset_model (envs/env01_v1.py:39-63 and clones).
 * mask_dev: [N] bytes, non-zero = reset that env; NULL = all.  obs_dev rows of untouched envs are left alone. */
int  so100_reset(so100_sim* sim, const uint8_t* mask_dev, const float* inject_dev, float* obs_dev, void* hip_stream);

/* One launch: the 4-wave latency kernel for N <= 16384, the one-wave-per-64-envs throughput kernel above (results agree
 * to the last bit or two). */
int  so100_step(so100_sim* sim, const so100_step_io* io, void* hip_stream);

/* ref: reads / writes of data.qpos, data.qvel (MjData), SoA: qpos_dev [13][N], qvel_dev [12][N] */
int  so100_get_state(so100_sim* sim, float* qpos_dev, float* qvel_dev, void* hip_stream);
int  so100_set_state(so100_sim* sim, const float* qpos_dev, const float* qvel_dev, void* hip_stream);
/* any single row of the state matrix ([N] 4-byte words; integer rows are int32 bit patterns) */
int  so100_get_field(so100_sim* sim, int32_t field, void* out_dev, void* hip_stream);
int  so100_set_field(so100_sim* sim, int32_t field, const void* in_dev, void* hip_stream);

const char* so100_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SO100_SIM_H */
