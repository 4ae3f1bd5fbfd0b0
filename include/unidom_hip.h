/*
 * unidom_hip.h -- C ABI of libunidom_hip.so: MI355X (gfx950) kernels for the differentiable-physics
 * hot path of Kuroki1931/UniDOM (the substep that APG training drives through env.step_diff).
 *
 * The reference has no FFI for this path: it sits behind a Python/JAX functional API
 *     simulator.step_jax(state, action) -> (state, state)
 * consumed by lax.scan in step_diff and differentiated by jax.grad.  Each entry point below names the
 * reference interface it replaces (paths relative to /root/reference/DaXBench/daxbench/).
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer (HBM) unless marked "host"; float32, C-contiguous, AoS at the
 *     boundary ([B,P,3] etc., the reference's own layouts); the library transposes to SoA internally.
 *   - purely functional like the reference (NamedTuple._replace): inputs are never written, the caller owns
 *     every state / gradient / checkpoint buffer; a handle owns constant tables and, for the many-workgroup MPM and
 *     PLB paths, a scratch arena (HBM grid + active-cell lists) that is (re)allocated when a larger batch arrives.
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*); no allocation and no host synchronisation
 *     inside any rollout / step / loss call: handle-owned scratch is allocated in ud_*_create for conf.max_envs envs
 *     (the reference's per-handle constants, mpm_simulator.py:117-122,154: the batch size is known when the simulator
 *     is built) and a call with more envs than that returns UD_ERR_INVALID.  The only entry points that synchronise
 *     are the ud_*_poll_* / ud_*_reset calls, which say so.
 *   - return value: 0 = ok, negative = ud_status; ud_last_error() gives the text (thread-local).
 *   - a handle is bound to the device current at create time; not thread-safe per handle.
 */
#ifndef UNIDOM_HIP_H
#define UNIDOM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  UD_OK = 0,
  UD_ERR_INVALID = -1,      /* bad argument (null pointer, size out of range) */
  UD_ERR_UNSUPPORTED = -2,  /* configuration the kernels do not cover (stated in the message) */
  UD_ERR_HIP = -3,          /* a HIP runtime call failed (no device, launch error, ...) */
  UD_ERR_OVERFLOW = -4      /* a device-side capacity (MPM LDS cell table) was exceeded */
} ud_status;

const char* ud_last_error(void);
/* 3-part version + the gfx arch the code object was built for, e.g. "0.1.0 gfx950" */
const char* ud_version(void);

/* ------------------------------------------------------------------------------------------------
 * Cloth (mass-spring) -- replaces ClothSimulator.step_jax = vmap(jit(robot_step_wrapper))
 *   core/engine/cloth_simulator.py:26-70 (tables), :163-180 (robot_step), :257-337 (step),
 *   :198-226 (gripper), :182-196 (norm_grad), :228-255 (what is differentiated)
 * driven by lax.scan over the 40 macro actions of one step_diff (core/envs/basic/cloth_env.py:211).
 * ------------------------------------------------------------------------------------------------ */
typedef struct ud_cloth ud_cloth;

typedef struct {
  int N;           /* lattice size (80)                    fold_cloth1_env.py:17 */
  float gravity;   /* 0.5                                  :19 */
  float damping;   /* 2                                    :21 */
  float dt;        /* 2e-3                                 :22 */
  float max_v;     /* 2.0                                  :23 */
  float small_num; /* 1e-8                                 :24 */
  int substeps;    /* 50 = fori_loop bound                 cloth_simulator.py:176 */
  int mode;        /* 0 (default): forward in operation order "v2" -- the reference's formulas re-associated into fewer
                    *    IEEE operations, no FMA contraction; bit-identical to the CPU restatement compiled in the SAME
                    *    order (oracle cloth_substep_fwd_v2), and 1.5e-5 (v, one substep) / 9e-3 (50 substeps) away from
                    *    the reference-order restatement in f32, 1e-9 in f64 -- + restructured adjoint
                    *    kernel (one reduction round per substep);
                    * 1: forward in the reference's literal f32 operation order (bit-identical to the CPU restatement
                    *    of that order) AND reference-order adjoint (six reductions per substep);
                    * 2: v2 structure with fast-math (v_rsq, FMA) forward + restructured adjoint (f32 round-off
                    *    differences, which this stiff system amplifies over long rollouts -- see DESIGN.md).
                    * 3: forward in the reference's literal f32 operation order (mode 1's forward kernel: k*r/len*(len-L0)/L0,
                    *    cloth_simulator.py:264-268, and the friction block :281-306 as written; bit-identical to the CPU
                    *    restatement of THAT order) + the restructured adjoint of modes 0 / 2 (it reads the same checkpoint
                    *    records and re-derives the grasp sets from them).  Bodies of at most 512 particles
                    *    (UD_ERR_UNSUPPORTED above).
                    * Bodies above 1024 particles: modes 0 / 2 run the v2-order forward and the restructured adjoint on
                    *    SEVERAL workgroups per env (512 particles each, halo positions / force cotangents / block sums handed
                    *    over through HBM every substep; forward still bit-identical to the v2 restatement); mode 1 runs the
                    *    reference-order kernels on one 1024-lane workgroup per env. */
  int max_envs;    /* the largest B any call on this handle will pass (>= 1): handle-owned scratch (bodies above 1024 particles: the
                    * several-workgroup kernels' hand-off arena, or the one-workgroup adjoint's parking area) is allocated in
                    * ud_cloth_create for this many envs; a call with more returns UD_ERR_INVALID */
  int one_workgroup_per_env;   /* bodies above 1024 particles only.  != 0: keep the one-workgroup-per-env kernels where the library would
                    * run several workgroups per env (diagnostics and tests; fixed at create) */
} ud_cloth_conf;

/* mask: host pointer, N*N bytes, row-major, non-zero = cloth particle (create_cloth_mask,
 * fold_cloth1_env.py:48-53).  Particle order = row-major nonzero(mask) (cloth_simulator.py:52).
 * Limits: 1 <= P <= 4096 (P <= 1024: one workgroup per env, one particle per lane; above: ceil(P / 512) workgroups per
 * env, one particle per lane -- a call is cut into launches of ud_cloth_launch_envs() envs so that the workgroups that
 * wait for each other are resident together -- or, in mode 1 / when a spring spans more than 256 particle indices, one
 * workgroup per env with up to four particles per lane); the mask must not touch the lattice border
 * (UD_ERR_UNSUPPORTED otherwise). */
int ud_cloth_create(const ud_cloth_conf* conf, const uint8_t* mask, ud_cloth** out);
void ud_cloth_destroy(ud_cloth* h);
int ud_cloth_num_particles(const ud_cloth* h);
/* envs per kernel launch of a call with B envs (B itself unless the several-workgroup kernels run: then at most
 * 8 * floor((CUs / 8) / parts), every XCD holding whole envs) */
int ud_cloth_launch_envs(const ud_cloth* h, int B);
/* Several-workgroup kernels only: a part that waits for a sibling longer than ~seconds gives up, the env's outputs are
 * NaN and a device-side counter is raised.  This call SYNCHRONISES `stream`, returns the number of such workgroups
 * since the previous call (0 = none; also 0 for handles that never take that path; ud_last_error() holds the text
 * otherwise) and resets the counter.  The rollout calls themselves stay asynchronous and return UD_OK. */
int ud_cloth_poll_timeouts(ud_cloth* h, void* stream);
/* bytes of the per-substep checkpoint arena a forward rollout of T macro steps x B envs writes */
size_t ud_cloth_ckpt_bytes(const ud_cloth* h, int B, int T);

/* Forward: T macro steps (robot_step) of `substeps` substeps each, for B independent envs.
 *   x, v [B,P,3]; prim [B,2,4] = (primitive0, primitive1) = (pos xyz, radius); stiffness, mu [B];
 *   actions [T,B,8] (the scan xs: get_pnp_actions output, cloth_env.py:134-173).
 * Outputs: final x_out, v_out [B,P,3], prim_out [B,2,4]; optional per-macro-step state_list
 *   x_list, v_list [T,B,P,3], prim_list [T,B,2,4] (the scan ys, may be NULL);
 *   ckpt (may be NULL = no backward): ud_cloth_ckpt_bytes() bytes, opaque, consumed by ud_cloth_rollout_bwd;
 *   grasp (may be NULL, debug/tests): [T,substeps,B,2,P] bytes, 1 where the gripper mask was true (Q3). */
int ud_cloth_rollout_fwd(ud_cloth* h, int B, int T, const float* x, const float* v, const float* prim,
                         const float* stiffness, const float* mu, const float* actions, float* x_out,
                         float* v_out, float* prim_out, float* x_list, float* v_list, float* prim_list,
                         void* ckpt, uint8_t* grasp, void* stream);

/* Backward: the adjoint jax.grad produces through robot_step_wrapper / step_wrapper
 * (cloth_simulator.py:107-145, :228-255) including the per-substep norm_grad rescaling (:189-194) when
 * normalize != 0.  g_x, g_v [B,P,3], g_prim [B,2,4]: cotangents of the final state;
 * g_*_list (may be NULL): cotangents of the scan ys.  Outputs: cotangents of the initial state,
 * of actions [T,B,8] and the contributions to stiffness / mu [B] (the identity pass-through of those two
 * state fields is the caller's to add). */
int ud_cloth_rollout_bwd(ud_cloth* h, int B, int T, const void* ckpt, const float* stiffness, const float* mu,
                         const float* actions, const float* g_x, const float* g_v, const float* g_prim,
                         const float* g_x_list, const float* g_v_list, const float* g_prim_list, int normalize,
                         float* g_x0, float* g_v0, float* g_prim0, float* g_actions, float* g_stiffness,
                         float* g_mu, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MLS-MPM -- replaces SimpleMPMSimulator.step_jax = vmap(jit(step))
 *   core/engine/mpm_simulator.py:27-63 (constructor), :413-429 (step), :223-330 (substep),
 *   :178-221 (p2g_micro / g2p_micro), :365-373 (copy_frame), :375-411 (norm_grad_state / norm_grad),
 *   core/engine/svd_safe_batch.py:19-102 (svd + safe VJP),
 *   core/engine/primitives/primitives.py:185-239 (forward_kinematics, set_action, position_control_batch),
 *   core/engine/primitives/box.py:6-18 (box SDF)
 * driven by lax.scan over the macro actions of one step_diff (core/envs/basic/mpm_env.py:141).
 * One call = one `simulator.step` = conf.steps substeps for B independent envs, ONE kernel launch.
 *   core/engine/primitives/primitives.py:105-182 (inv_trans, sdf, finite-difference normal, collider velocity,
 *   collide_batch)
 *   core/engine/primitives/container.py:8-16 (container SDF)
 * Scope this round: one box primitive in position-control mode (whip_rope); 1..4 box or container primitives in
 * soft-contact mode (collide_batch: shape_rope, pour_water), applied one after the other (mpm_simulator.py:292-294).
 * With n_primitive = P > 1 every primitive array gains a primitive axis after the env axis -- position [B,P,steps,3],
 * rotation [B,P,steps,4], size [B,P,3], v / w [B,P,steps,3], action [B,6P] -- in both directions.  Materials 1 (elastic), 2 (plastic clamp) and 0 (liquid mu=0, la=1) in the particle pre-pass.
 * Position control with N <= 128 particles per env: one workgroup per env, the touched part of the `res` grid in an
 * LDS cell table, ONE launch per step.  N > 128, or soft contact: many workgroups per env, dense grid in HBM, a few
 * launches per substep (all on `stream`).
 * ------------------------------------------------------------------------------------------------ */
typedef struct ud_mpm ud_mpm;

typedef struct {
  int n_particles;
  int n_grid;                /* conf.n_grid (dx = 1/n_grid)                 whip_rope_env.py:47 */
  int res[3];                /* conf.res: allocated grid (index wrap/clamp)  :59 */
  int steps;                 /* conf.steps substeps per step                :51 */
  float dt;                  /* :48 */
  float p_mass, p_vol;       /* :62-63 */
  float gravity[3];          /* :64 */
  int use_position_control;  /* 1: position_control_batch (primitives.py:232-239), 0: collide_batch soft contact (:154-182) */
  float prim_friction;       /* PrimitiveState.friction  (create_primitive, mpm_env.py:201-207): collide_batch only */
  float prim_softness;       /* PrimitiveState.softness  (666 in every reference env): collide_batch only */
  int n_primitive;           /* conf.n_primitive (0 = 1).  > 1 (up to 4) in soft-contact mode only: pour_water_env.py:32 */
  int sdf_kind;              /* the SDF installed by set_sdf (primitives.py:26-28): 0 box (box.py:6-18), 1 container =
                                cut hollow sphere, size = (r, h, t) (container.py:8-16); soft-contact mode only */
  int grid_ckpt_cells;       /* many-workgroup path only.  0: the backward recomputes p2g + grid op per substep (the reference
                                rematerialises the whole substep, mpm_simulator.py:332-359).  K > 0: the forward also
                                checkpoints the active grid cells (32 B each) into a pool of K * n_particles records per
                                substep on average, inside the caller's checkpoint (ud_mpm_ckpt_bytes grows accordingly),
                                and the backward restores them.  A pool that runs out flags the env in status[] (1) */
  int sort_particles;        /* many-workgroup path only.  != 0: at every step the handle re-orders the particles by grid cell
                                (Morton key) internally, for bodies whose particles come in no spatial order (uniformly
                                sampled liquids, mpm_simulator.py:87-91); bodies above 8192 particles keep their order (the
                                sort runs in the LDS of one workgroup per env).  Invisible at this boundary: inputs, outputs and
                                gradients stay in the caller's order; only the summation order of the scatters changes */
  float prim_friction_each[4], prim_softness_each[4];   /* soft contact with several primitives: friction / softness of primitive i where
                                prim_softness_each[i] > 0 (create_primitive passes them per primitive, mpm_env.py:201-217); entries left
                                at 0 take prim_friction / prim_softness (every reference env passes 0.1 / 666 to all of them) */
  int deterministic;         /* != 0: ud_mpm_step_fwd sums every grid cell in the order of the reference's flattened scatter-add array
                                (mpm_simulator.py:178-194: [27 offsets][N particles] -- offsets in (i, j, k) order, the particles of an
                                offset in ascending index), in f32, and g2p adds its cells in (i, j, k) order; no float atomics, IEEE
                                arithmetic (no FMA contraction, correctly rounded divide / sqrt).  Two calls on the same inputs return
                                the same bits, and x, v, C, F equal the CPU build of the same source bit for bit (tests/test_mpm_det.py).
                                Per substep the particles are bucketed by base cell (one sort per env) and a cell walks the 27 buckets
                                that can reach it: 3.1x the default forward at 67 particles, 5.3x at 798 (profiles/r04*_det_cost.txt; rounds
                                2-3, every cell walking every particle: 13x / 332x).  Position control or soft contact (collide_batch of
                                box / container primitives: its exp is a plain-IEEE polynomial in this mode so that both builds agree; a
                                primitive's ROTATION still goes through the platform's sinf / cosf); at most 8192 particles
                                (UD_ERR_UNSUPPORTED otherwise); grid_ckpt_cells and sort_particles are ignored.  ud_mpm_step_bwd of such a
                                handle is deterministic too -- the grid recomputed by the deterministic forward's kernels, the g2p adjoint's
                                scatter an ordered sum per cell over (offset, particle), every per-env sum (ground friction, controlled
                                velocity, the collide adjoint's per-primitive cotangents, mu / lamda, the clip's norm) added in a fixed
                                order: two calls return the same bits (fast-math arithmetic, so no CPU build is bit-equal to it; checked
                                against the oracle by tolerance).  At most 32768 touched cells per env and substep (status[b] = 1 beyond) */
  int max_envs;              /* the largest B any call on this handle will pass (>= 1).  Every arena of the many-workgroup path (dense
                                grids, active lists, cotangent grids, the persistent forward's rotating grids, the deterministic mode's
                                scratch) is allocated in ud_mpm_create for this many envs: no step call allocates or synchronises the
                                host; a call with B > max_envs returns UD_ERR_INVALID.  (SimpleMPMSimulator's batch size,
                                mpm_simulator.py:27-63: known when the simulator is built) */
  /* Kernel selection of the many-workgroup path, fixed at create; 0 everywhere = the library's choice by measurement,
     which ud_mpm_launch_plan reports.  The other values exist for diagnostics and so that the tests can put every kernel before the
     oracle at sizes the oracle can follow: */
  int tune_lanes;            /* lanes per particle of the particle kernels: 0 = 4 below 100 000 particles per launch, 1 beyond; 1 / 4 force */
  int tune_cluster;          /* the persistent forward (one launch per step call): 0 = solids with one primitive in the four-lane regime;
                                1 = wherever its parts fit the chip; -1 = never (the multi-kernel forward) */
  int tune_cluster_part_lanes; /* its lanes per part: 0 = 128 (32 particles), 64 (16 particles: a part can never outgrow its cell table) */
  int tune_cluster_envs;     /* > 0: at most this many envs per persistent launch (several launches per call) */
  int tune_env_groups;       /* > 0: this many stream groups on the multi-kernel path (1 = everything on the caller's stream) */
  int tune_bwd_two_launch;   /* backward with the grid checkpoint: 0 = two launches per reverse substep where they apply (four lanes, one
                                primitive); -1 = always the four-kernel sequence */
  int tune_collide_records;  /* soft contact with the grid checkpoint: 0 = the forward's grid op leaves exp(-dist * softness) and the
                                finite-difference normal of every cell and primitive beside the checkpoint and the grid-op adjoint reads them
                                (16 B per cell, primitive and substep each way; same bits as evaluating the SDFs again); -1 = the adjoint
                                evaluates the SDFs again (7 per cell and primitive, three passes for two primitives) */
} ud_mpm_conf;

/* material, hardness: host arrays [n_particles] (SimpleMPMSimulator.material / .h, mpm_simulator.py:117-122) */
int ud_mpm_create(const ud_mpm_conf* conf, const int* material, const float* hardness, ud_mpm** out);
void ud_mpm_destroy(ud_mpm* h);
/* bytes of the checkpoint one ud_mpm_step_fwd call with B envs writes for its backward: the particle state of every substep; on
 * the many-workgroup path also the primitive rows, the spatial order, the grid-checkpoint pool and -- while B * n_particles is
 * below 100 000 (the regime where a launch waits for one wave's serial chain) -- the SVD factors of every substep's F, which
 * the backward reads instead of iterating again.  The layout is the library's; the same B must be passed to the backward */
size_t ud_mpm_ckpt_bytes(const ud_mpm* h, int B);
/* Which kernels a call with B envs runs (the choice is the library's, by measurement: DESIGN.md 3.2); for logs and benchmark
 * labels, nothing at this boundary depends on it.  0: one workgroup per env (bodies up to 128 particles, one box primitive).
 * Bit 0: the many-workgroup path.  Bit 1: its forward is one persistent launch per group of envs (several workgroups per
 * env that hand grid cells to each other through HBM), else four launches per substep.  Bit 2: its backward, restoring
 * the grid from the checkpoint (grid_ckpt_cells > 0), takes two launches per substep, else four (six when it recomputes).
 * Negative: bad arguments. */
int ud_mpm_launch_plan(const ud_mpm* h, int B);
/* Measurement aid (bench.py's algorithmic-byte count, tools/traffic_table.py): cells[b] = the number of grid-checkpoint records env b's forward
 * left in `ckpt` = its active grid cells summed over the substeps of that step call (SURVEY 8(d)'s G_act, per substep, is cells[b] / substeps).
 * `cells`: device array of B ints, written asynchronously on `stream`.  Handles without a grid checkpoint (one workgroup per env,
 * grid_ckpt_cells = 0, deterministic mode) write 0.  Same B as the forward that wrote `ckpt`. */
int ud_mpm_ckpt_cells(const ud_mpm* h, int B, const void* ckpt, int* cells, void* stream);
/* Puts every handle-owned arena back into its rest state, asynchronously on `stream` (no host synchronisation).  For the one case in which
 * a step call can leave them dirty: status bit 4 of the persistent forward (a workgroup gave up waiting for its siblings and skipped
 * the zeroing of its cells).  Until then later calls on the handle would sum into stale cells.  The Python mirror calls it when its
 * status check sees that bit.  The persistent forward needs every workgroup of a launch resident at once (launch size = occupancy x CUs
 * of an otherwise idle device): kernels of other streams or processes that occupy CUs for seconds are what can cause such a time-out. */
int ud_mpm_reset(ud_mpm* h, void* stream);

/* Forward `step`: state (x,v [B,N,3]; C,F [B,N,3,3]; J [B,N]), primitive 0 (position [B,steps,3], rotation
 * [B,steps,4] (w,x,y,z), size [B,3]), friction/mu/lamda [B], action [B,6] -> new state, primitive position /
 * rotation after copy_frame(steps,0) and the v,w [B,steps,3] written by set_action.
 * ckpt (may be NULL = no backward): ud_mpm_ckpt_bytes() bytes. status [B] int32, written asynchronously, the caller reads
 * it when it next synchronises: 0 ok.  One-workgroup path: 1 = LDS cell table overflow in that env (outputs invalid).
 * Many-workgroup path, a bit mask: 1 = the grid checkpoint of that env is incomplete -- its pool ran out, or (persistent forward) a
 * part of 32 particles touched more cells than its 512-slot table holds and sent the rest to the HBM grid directly (outputs valid; that
 * step's backward must recompute the grid: clip bit 1 of ud_mpm_step_bwd); 2 = a part's spill list overflowed as well (more than 896
 * cells for 32 particles: cannot happen with a 27-cell stencil; outputs invalid); 4 = a workgroup gave up waiting for a sibling
 * (outputs invalid; call ud_mpm_reset before the next step). */
int ud_mpm_step_fwd(ud_mpm* h, int B, const float* x, const float* v, const float* C, const float* F, const float* J,
                    const float* prim_position, const float* prim_rotation, const float* prim_size,
                    const float* friction, const float* mu, const float* lamda, const float* action, float* x_out,
                    float* v_out, float* C_out, float* F_out, float* J_out, float* prim_position_out,
                    float* prim_rotation_out, float* prim_v_out, float* prim_w_out, void* ckpt, int* status,
                    void* stream);

/* Backward `step`: cotangents of (x,v,C,F, primitive position, primitive rotation) at the step output ->
 * cotangents at the step input plus friction/mu/lamda [B] and action [B,6].  J receives no gradient (excluded
 * from the reference's substep loss, mpm_simulator.py:343-350).  g_prim_rotation (in) and g_prim_rotation0 (out)
 * [B,steps,4] may be NULL (= zero cotangent / not wanted).
 * Position control: nothing reaches the rotation array (g_prim_rotation0 = 0) and action[3:6] is reported as 0 --
 * the reference yields NaN there for w = 0 and launders it with nan_to_num at this boundary.
 * Soft contact: rotation and action[3:6] carry the chain rule as the reference writes it, NaN at w = 0 included
 * (d|w|/dw, primitives.py:86); clip != 0 launders it exactly like the reference.
 * clip bit 0 applies norm_grad_state / norm_grad (nan_to_num + global-norm clip to 1, :389-408); in soft-contact
 * mode the norm also covers the cotangents of the primitive's rotation, size, friction and action_scale leaves.
 * clip bit 1 (many-workgroup path with grid_ckpt_cells > 0): ignore the grid checkpoint and recompute p2g + grid op --
 * for a step whose forward flagged a pool overflow in status[]; the particle history in the checkpoint is always complete. */
int ud_mpm_step_bwd(ud_mpm* h, int B, const void* ckpt, const float* prim_size, const float* friction,
                    const float* mu, const float* lamda, const float* action, const float* g_x, const float* g_v,
                    const float* g_C, const float* g_F, const float* g_prim_position, const float* g_prim_rotation, int clip,
                    float* g_x0, float* g_v0, float* g_C0, float* g_F0, float* g_prim_position0, float* g_prim_rotation0,
                    float* g_friction, float* g_mu, float* g_lamda, float* g_action, int* status, void* stream);

/* ------------------------------------------------------------------------------------------------
 * PlasticineLab-style MLS-MPM, float64, von-Mises plasticity, sticky Sphere primitives (GenORM Torus task,
 * BASELINE config 5) -- replaces TaichiEnv.step -> MPMSimulator.step(is_copy=True) = `substeps` x substep
 *   GenORM/policy/pbm/plb/engine/mpm_simulator.py:438-449 (step), :256-268 (substep: clear_grid, compute_F_tmp,
 *   svd, p2g :166-195 with compute_von_mises :133-150, forward_kinematics, grid_op :200-232, g2p :234-253),
 *   engine/primitive/primitives.py:17-53 (Sphere), engine/primitive/primive_base.py:118-121,185-192.
 * Adjoint (ud_plb_step_bwd): substep_grad :271-289 with backward_svd :107-124; the differentiable leaves are the particle
 * state, primitive 0's action and position, and per env E, nu, yield_stress (PlasticineLab/sim2sim/plb/engine/
 * mpm_simulator.py:27-29,485-498: get_parameter_grad) and the ground friction (optimize_ground_friction, :57-58).
 * Losses (ud_plb_loss_*): engine/losses/loss.py:112-243 (density, SDF, soft / hard contact, weighted sum).
 * Parity for these entry points is UNPINNED: taichi is absent and the reference ships no recording of this path.
 * ------------------------------------------------------------------------------------------------ */
typedef struct ud_plb ud_plb;

typedef struct {
  int n_particles;
  int n_grid;            /* int(128 * quality * 0.5)                  mpm_simulator.py:14-18 */
  int substeps;          /* int(2e-3 // dt)                           :32 */
  double dt;             /* 0.5e-4 / (quality * 0.5)                  :21 */
  double gravity[3];     /* SIMULATOR.gravity (the kernel applies x30, :205) */
  double ground_friction;
  int n_primitives;      /* 1 or 2 Spheres; only primitive 0 is actuated (3 action dims) */
  double radius[2];
  double lower_bound[3], upper_bound[3];   /* primitive xyz_limit */
  int grid_ckpt_cells;   /* 0: ud_plb_step_bwd runs p2g again for every substep (substep_grad recomputes the whole substep,
                            mpm_simulator.py:271-289).  K > 0: a forward with a checkpoint also keeps the touched grid cells
                            (index, m, mv: 36 B each; up to min(27, K) * n_particles per substep, inside the caller's checkpoint:
                            ud_plb_ckpt_bytes grows accordingly) and the backward restores them; a substep that touched more
                            cells than that falls back to recomputing, per env, on the device.  K >= 27 can never fall back.
                            Multi-kernel path only (the persistent path always keeps its parts' cells: ud_plb_ckpt_bytes says what a
                            checkpoint takes) */
  int max_envs;          /* the largest B any call on this handle will pass (>= 1).  Every arena -- exchange grids, active lists, adjoint
                            and loss scratch -- is allocated in ud_plb_create for this many envs; no step or loss call allocates or
                            synchronises the host; a call with B > max_envs returns UD_ERR_INVALID */
  int path;              /* which kernels the handle runs, fixed at create (ud_plb_launch_plan reports it).  0: the library's choice --
                            ONE persistent launch per step call and direction (csrc/plb_cluster.hip: parts of 32 particles, one
                            workgroup each, the particle state in registers, the grid exchanged through HBM once per substep) where
                            every workgroup of a launch can be resident and the exchange grids fit (bodies up to ~8 000 particles per
                            env), else the multi-kernel path (2 launches per forward substep, 5 per reverse one).  1: the multi-kernel
                            path.  2: the persistent path (UD_ERR_UNSUPPORTED where it cannot run) */
  int lanes;             /* multi-kernel path: lanes per particle in the particle kernels.  0: by launch size (8 up to 16 000 particles
                            per launch, 4 below 100 000, 1 beyond); 1 / 4 / 8 force one mapping (how the tests put all three before
                            the restatement at their sizes) */
  int sort_every;        /* the handle orders each env's particles by grid cell internally (invisible at this boundary) on the first
                            call and every sort_every-th forward call after it; 0 = 8; negative = never */
} ud_plb_conf;

int ud_plb_create(const ud_plb_conf* conf, ud_plb** out);
void ud_plb_destroy(ud_plb* h);
/* 1: the multi-kernel path, 2: the persistent path (one launch per step call and direction); negative: bad arguments (B > max_envs) */
int ud_plb_launch_plan(const ud_plb* h, int B);
/* Persistent path only: a workgroup that waits for a sibling longer than ~seconds gives up, its env's outputs are NaN and a device-side
 * counter is raised.  This call SYNCHRONISES `stream`, returns the number of such workgroups since the previous call (0 = none, also on
 * the multi-kernel path; ud_last_error() holds the text otherwise) and, when it is not 0, puts the handle's exchange arena back into its
 * rest state so that later calls are valid again.  The step calls themselves stay asynchronous and return UD_OK.  Between a time-out
 * and the poll that clears it the handle is POISONED: every persistent launch sees the raised counter at entry, touches neither barrier
 * words nor exchange grids and writes NaN to all its outputs -- a time-out can never turn into finite-but-wrong results of later steps. */
int ud_plb_poll_timeouts(ud_plb* h, void* stream);
/* One env.step for B independent envs (float64 device arrays): x, v [B,N,3]; C, F [B,N,3,3]; prim_pos
 * [B,n_primitives,3]; softness [B,n_primitives]; action [B,3]; E, nu, yield_stress [B]. */
int ud_plb_step_fwd(ud_plb* h, int B, const double* x, const double* v, const double* C, const double* F,
                    const double* prim_pos, const double* softness, const double* action, const double* E,
                    const double* nu, const double* yield_stress, double* x_out, double* v_out, double* C_out,
                    double* F_out, double* prim_pos_out, void* ckpt, void* stream);
/* ckpt (may be NULL = no backward): ud_plb_ckpt_bytes(h, B) bytes, caller-owned, opaque: every substep's particle state,
 * the primitive trajectory and the internal spatial order of this call, consumed by ud_plb_step_bwd. */
size_t ud_plb_ckpt_bytes(const ud_plb* h, int B);

/* Adjoint of one env.step.  g_x, g_v [B,N,3], g_C, g_F [B,N,3,3], g_prim_pos [B,n_primitives,3]: cotangents of the
 * step's outputs (any may be NULL = zero).  Outputs: cotangents of the inputs x, v, C, F (required), of prim_pos
 * [B,n_primitives,3], of action [B,3], and per env of E, nu, yield_stress and the ground friction [B] (each may be NULL).
 * softness, action, E, nu, yield_stress: the forward call's inputs again. */
int ud_plb_step_bwd(ud_plb* h, int B, const void* ckpt, const double* softness, const double* action, const double* E,
                    const double* nu, const double* yield_stress, const double* g_x, const double* g_v, const double* g_C,
                    const double* g_F, const double* g_prim_pos, double* g_x0, double* g_v0, double* g_C0, double* g_F0,
                    double* g_prim_pos0, double* g_action, double* g_E, double* g_nu, double* g_yield_stress,
                    double* g_ground_friction, void* stream);

/* Loss of a particle state (engine/losses/loss.py): x [B,N,3], prim_pos [B,n_primitives,3], target_density and target_sdf
 * [n_grid^3] (shared by the envs), weights [3] = (contact, density, sdf) -- all device arrays.
 *   density = sum_I |grid_mass_I - target_density_I|, sdf = sum_I target_sdf_I grid_mass_I  with grid_mass = the p2g of the
 *   particle masses (compute_grid_m_kernel); contact = sum_primitives min_dist^2 with, soft_contact != 0: min_dist =
 *   sum_i d_i w(d_i) / sum_i w(d_i), w(d) = 1 / (1 + 1e4 d^2), d_i = max(sdf(x_i), 0); soft_contact == 0: min_i d_i.
 * loss [B] = weights . (contact, density, sdf); parts [B,3] (may be NULL) = the three terms.
 * ud_plb_loss_bwd: g_loss [B] -> g_x [B,N,3], g_prim_pos [B,n_primitives,3] (may be NULL). */
int ud_plb_loss_fwd(ud_plb* h, int B, const double* x, const double* prim_pos, const double* target_density,
                    const double* target_sdf, const double* weights, int soft_contact, double* loss, double* parts, void* stream);
int ud_plb_loss_bwd(ud_plb* h, int B, const double* x, const double* prim_pos, const double* target_density,
                    const double* target_sdf, const double* weights, int soft_contact, const double* g_loss, double* g_x,
                    double* g_prim_pos, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Cloth-env arithmetic either side of the rollout (the reference jit-fuses it into step_diff; here one forward and
 * one backward kernel per piece, one workgroup per env, instead of ~180 elementwise launches per step_diff).
 *   ud_chamfer_*    core/utils/util.py:138-153  calc_chamfer(x [B,P,3], y [Q,3]) -> [B]:
 *                   d(p,q) = sqrt(mean_xyz((x_p - y_q)^2)); out = mean_q min_p d + mean_p min_q d.
 *                   idx_xy [B,P] / idx_yx [B,Q] (int32) are the argmins the backward routes the cotangent through.
 *   ud_cloth_pnp_*  core/envs/basic/cloth_env.py:134-173 get_pnp_actions (actions [B,6], primitive0 [B,4]) ->
 *                   macro_actions [40,B,8], and :206-209 contact_distance [B] = min_p |actions[:, :3] - x_p|.
 * P, Q <= 4096.  The backward entry points return what jax.grad returns for the same expressions.
 * ------------------------------------------------------------------------------------------------ */
int ud_chamfer_fwd(int B, int P, int Q, const float* x, const float* y, float* out, int* idx_xy, int* idx_yx, void* stream);
int ud_chamfer_bwd(int B, int P, int Q, const float* x, const float* y, const int* idx_xy, const int* idx_yx,
                   const float* g_out, float* g_x, void* stream);
int ud_cloth_pnp_fwd(int B, int P, const float* actions, const float* primitive0, const float* x, float* macro_actions,
                     float* contact_distance, int* contact_idx, void* stream);
/* g_contact_distance may be NULL (no auxiliary reward); g_x [B,P,3] is written in full (zero except the contact row) */
int ud_cloth_pnp_bwd(int B, int P, const float* actions, const float* x, const float* contact_distance,
                     const int* contact_idx, const float* g_macro_actions, const float* g_contact_distance,
                     float* g_actions, float* g_primitive0, float* g_x, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MPM-env arithmetic either side of simulator.step (core/envs/basic/mpm_env.py), one workgroup per env.
 *   ud_mpm_focus_*   pre_step :99-114: shift [B,3] = (cx - mean_n x.x, 0, cz - mean_n x.z) with (cx, cz) = res/2/n_grid;
 *                    x_out = x + shift; prim_pos_out[p] = prim_pos[p] + shift for the n_prim (<= 4) primitive
 *                    trajectories [B,S,3].  prim_pos / prim_pos_out are HOST arrays of n_prim device pointers.
 *   ud_mpm_finish_*  post_step :116-125 (x - shift, prim_pos - shift; shift NULL = no focus), nan_to_num on x v C F J
 *                    :150-154, reward_func :90-94 = e ** (-10 * mean_n sqrt(mean_xyz((x - goal)^2))) with goal [Q,3], Q = N or 1
 *                    (calc_l2 broadcasts; a missing goal file is zeros((1,3)), mpm_env.py:46-48),
 *                    get_obs :57-76: obs [B, 6N + 3S] = x | v | primitive 0 trajectory.
 * Backward entry points: any cotangent pointer (and any entry of g_prim_pos_out) may be NULL = that output was unused.
 * nan_to_num passes a cotangent only where its argument was finite (jnp.where selection).  ud_mpm_focus_bwd does not
 * write primitive cotangents: they pass through unchanged (g_prim_pos[p] = g_prim_pos_out[p]).
 * ------------------------------------------------------------------------------------------------ */
int ud_mpm_focus_fwd(int B, int N, int n_prim, int S, float cx, float cz, const float* x, const float* const* prim_pos,
                     float* x_out, float* const* prim_pos_out, float* shift, void* stream);
int ud_mpm_focus_bwd(int B, int N, int n_prim, int S, const float* g_x_out, const float* const* g_prim_pos_out,
                     const float* g_shift, float* g_x, void* stream);
int ud_mpm_finish_fwd(int B, int N, int n_prim, int S, int Q, const float* x, const float* v, const float* C, const float* F,
                      const float* J, const float* shift, const float* const* prim_pos, const float* goal, float* x_out,
                      float* v_out, float* C_out, float* F_out, float* J_out, float* const* prim_pos_out, float* reward,
                      float* obs, void* stream);
/* shift and g_shift are both NULL or both given; g_shift [B,3] = -(sum of every shifted cotangent) */
int ud_mpm_finish_bwd(int B, int N, int n_prim, int S, int Q, const float* x, const float* v, const float* C, const float* F,
                      const float* shift, const float* goal, const float* reward, const float* g_x_out, const float* g_v_out,
                      const float* g_C_out, const float* g_F_out, const float* const* g_prim_pos_out, const float* g_reward,
                      const float* g_obs, float* g_x, float* g_v, float* g_C, float* g_F, float* const* g_prim_pos,
                      float* g_shift, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNIDOM_HIP_H */
