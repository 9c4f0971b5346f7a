/* tsim.h — C ABI of the MI355X-native batched tactile-simulation step (libtsim_hip.so).
 *
 * This is the drop-in boundary for ONE path of eanswer/TactileSimulation: what its python layer calls on
 * `redmax_py.Simulation` (pybind11 over the un-vendored DiffRedMax C++) from
 * envs/redmax_torch_functions.py.  Every entry point below names the reference call it replaces
 * (paths relative to the reference repo).  Plain pointers and sizes only — no torch types.
 *
 * One `tsim_batch` = B independent environments of one model, resident on one GPU.  The reference's
 * one-environment `Simulation` object is B = 1.  All array arguments are DEVICE pointers of the batch's
 * real type (float for TSIM_F32, double for TSIM_F64), env-major ([B][dim], C order) unless noted.
 * TSIM_F32 batches are mixed precision inside: positions, link poses and penetration depths are double, everything
 * else float; the interface stays float.
 * `stream` is a hipStream_t (NULL = default stream).  Functions return 0 on success, non-zero on error
 * (message via tsim_last_error()).  Calls on one batch must be serialised by the caller (the reference's
 * Simulation is single-threaded as well: algorithms/gd.py:30 torch.set_num_threads(1)).
 */
#ifndef TSIM_H
#define TSIM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tsim_batch tsim_batch;
enum { TSIM_F32 = 0, TSIM_F64 = 1 };

/* redmax.Simulation(model_path)                       envs/redmax_torch_env.py:33
 * The XML is compiled on the host (tactilesimulation_amd/model/compiler.py) to the flat blob of
 * include/tsim_blob.h; I / F are HOST pointers.  tape_capacity = max number of recorded sub-steps
 * between reset() and backward (TactilePush: 100 env-steps x 5 = 500). */
int tsim_batch_create(const int32_t* I, const double* F, int B, int tape_capacity, int dtype, int device,
                      tsim_batch** out);
void tsim_batch_destroy(tsim_batch* b);

/* sim.ndof_r / ndof_u / ndof_var / ndof_tactile / options.h   envs/redmax_torch_env.py:35-37,
 * envs/redmax_torch_functions.py:161, envs/tactile_push_env.py:67 */
int tsim_ndof_r(const tsim_batch* b);
int tsim_ndof_u(const tsim_batch* b);
int tsim_ndof_var(const tsim_batch* b);
int tsim_ndof_tactile(const tsim_batch* b);
int tsim_batch_size(const tsim_batch* b);
int tsim_dtype(const tsim_batch* b);
double tsim_timestep(const tsim_batch* b);
int tsim_tape_len(const tsim_batch* b);          /* recorded sub-steps since the last reset */

/* update_* model edits (envs/dclaw_rotate_env.py:173-178, envs/tactile_insertion_env.py:254-279,
 * envs/stable_grasp_env.py:122): the host recompiles the blob and re-uploads it.  Topology (counts,
 * offsets) must be unchanged. */
int tsim_update_model(tsim_batch* b, const int32_t* I, const double* F, void* stream);

/* Batched counterpart of the update_* randomisers (envs/tactile_insertion_env.py:238-281 draws new contact / tactile
 * parameters per episode; with B environments each one gets its own): `tables` is a DEVICE array [B][tsim_table_size]
 * of the batch's real type whose rows are copies of the leading `tsim_table_size` reals of the blob's F[] (all numeric
 * tables: links, dofs, motors, variables, pairs, sensors) with the randomised entries overwritten. NULL reverts to the
 * shared model. Point arrays (contact points, taxels) are always shared.  The float header of a row (time step, gravity, Newton tolerance:
 * F[0 .. TSIM_FH_SIZE)) is NOT per environment: it is overwritten with the shared model's (the reference's randomisers never touch it).
 * On an fp32 batch whose model has a compiled-in structure this call checks on the device that every row keeps it and reads one int back: it
 * synchronises `stream` then.  Inside a stream capture the check cannot run: the batch then takes the generic kernels for as long as these tables
 * are set (tsim_kernel_variant says so) — set the tables before capturing to keep the compiled-in instantiation in the graph. */
int tsim_set_env_tables(tsim_batch* b, const void* tables, void* stream);
int tsim_table_size(const tsim_batch* b);

/* sim.set_state_init(q, qdot) + sim.reset(backward_flag)      envs/redmax_torch_functions.py:39-41,
 * envs/tactile_push_env.py:138,154.  q0 / qd0: [B][ndof_r].  Restarts the tape and zeroes the carried
 * adjoint. */
int tsim_reset(tsim_batch* b, const void* q0, const void* qd0, int backward_flag, void* stream);

/* The same for the environments with mask[env] != 0 only (DEVICE int32[B]); the others keep their state.  This is what
 * a batched roll-out collector does when single episodes end (the reference resets one `Simulation` at a time:
 * envs/dclaw_rotate_env.py:181-199 reset() per env).  Forward-only batches only: the tape is shared by the batch.  On a
 * BDF2 model the environment restarts with a constant-velocity history (q_-1 = q0 - h qd0). */
int tsim_reset_masked(tsim_batch* b, const void* q0, const void* qd0, const int32_t* mask, void* stream);

/* sim.set_u(u); sim.forward(num_steps, ...); sim.get_q(); sim.get_variables();
 * sim.get_tactile_force_vector()                               envs/redmax_torch_functions.py:131-136
 * (and :48-57 with num_steps = 1).  u: [B][ndof_u], held for num_steps implicit sub-steps.
 * Outputs (any may be NULL): q_out, qd_out [B][ndof_r]; var_out [B][ndof_var]; tac_out [B][ndof_tactile]
 * (taxel-major: shear0, shear1, normal); status [B] int32 = number of sub-steps whose Newton solve did not
 * reach tol (bit 30 set if a non-finite value appeared, in the state or in u: a NaN / inf control is flagged here rather than
 * swallowed by the motor law's clamp). */
int tsim_step(tsim_batch* b, const void* u, int num_steps, void* q_out, void* qd_out, void* var_out,
              void* tac_out, int32_t* status, void* stream);

/* sim.get_q() / get_qdot() / get_variables() / get_tactile_force_vector() at the current state, e.g.
 * right after reset (envs/tactile_push_env.py:157) or after forward() where the pad is too large to be read out inside the step
 * (examples/RollingBallExp/test_sim_speed.py:77-80: 40 000 taxels).  tsim_readout is two launches — kinematics of the current state,
 * then one kernel whose lanes are taxels — or, for pads of >= 4096 taxels right after a forward launch, the second one alone: that
 * launch leaves the pose records of the state it ends in.  Same results either way; any other change of state (reset, masked reset,
 * new model tables, cache pop) and any HIP-graph capture of this batch's launches switch back to two. */
int tsim_get_state(tsim_batch* b, void* q_out, void* qd_out, void* stream);
int tsim_readout(tsim_batch* b, void* var_out, void* tac_out, void* stream);

/* backward_info.set_flags / df_dq / df_dvar / df_dtactile; sim.backward_steps(n);
 * backward_results.df_du                                       envs/redmax_torch_functions.py:151-170
 * Adjoint of the newest n recorded sub-steps, newest first, continuing the adjoint carried from earlier
 * calls; pops them from the tape.
 *   seed_mode 0: df_dq [B][ndof_r], df_dvar [B][ndof_var], df_dtac [B][ndof_tactile] are the partials w.r.t.
 *                the outputs after the LAST of the n sub-steps (what StepSimFunction.backward builds by
 *                zero-padding, :153-163);
 *   seed_mode 1: [B][n][dim], step-major, oldest first (the general layout of the reference).
 * Any seed pointer may be NULL (= zeros).  df_du: [B][n][ndof_u], oldest first (reshape(num_steps, ndof_u)
 * at :170). */
int tsim_backward_steps(tsim_batch* b, int n, int seed_mode, const void* df_dq, const void* df_dvar,
                        const void* df_dtac, void* df_du, void* stream);

/* The open-loop episode of EpisodicSimFunction.forward             envs/redmax_torch_functions.py:46-57
 *   for t in range(T): sim.set_u(actions[t]); sim.forward(n); q, variables = getters;
 *                      if tactile_masks[t]: tactile = get_tactile_force_vector()
 * as ONE launch: frame f applies u[f] for num_steps sub-steps and writes its outputs to slot f.
 *   u [num_frames][B][ndof_u];  q_out, qd_out [num_frames][B][ndof_r], var_out [num_frames][B][ndof_var],
 *   tac_out [num_masked][B][ndof_tactile] (any may be NULL);  status [B]: non-converged sub-steps of the whole call.
 *   tactile_slot: DEVICE int32[num_frames], slot of frame f in tac_out or < 0 for "no tactile read-out at this frame"
 *   (the tactile_masks of :55-57 as an exclusive prefix count); NULL = every frame, slot f.
 * Results are bit-identical to num_frames calls of tsim_step; no environment waits for the slowest one of the batch
 * between env-steps, which is where the per-step launches lose their time. */
int tsim_rollout(tsim_batch* b, const void* u, int num_frames, int num_steps, const int32_t* tactile_slot,
                 void* q_out, void* qd_out, void* var_out, void* tac_out, int32_t* status, void* stream);

/* sim.backward() of EpisodicSimFunction.backward                  envs/redmax_torch_functions.py:77-92
 * Adjoint of the newest num_frames * num_steps sub-steps in one launch.  Seeds are the partials w.r.t. the outputs of
 * each frame (time-major like tsim_rollout's outputs): df_dq [num_frames][B][ndof_r], df_dvar [num_frames][B][ndof_var],
 * df_dtac [num_masked][B][ndof_tactile] (tactile_slot as in tsim_rollout), any may be NULL.
 * df_du [num_frames][B][ndof_u] = gradient w.r.t. u[f]
 * (summed over the frame's sub-steps).  Continues / leaves the carried adjoint like tsim_backward_steps. */
int tsim_backward_episode(tsim_batch* b, int num_frames, int num_steps, const int32_t* tactile_slot,
                          const void* df_dq, const void* df_dvar, const void* df_dtac, void* df_du, void* stream);

/* sim.backward() results df_dq0 / df_dqdot0                    envs/redmax_torch_functions.py:92-100
 * = the carried adjoint once the whole tape has been popped. [B][ndof_r] each. */
int tsim_get_adjoint(tsim_batch* b, void* df_dq0, void* df_dqd0, void* stream);

/* sim.saveBackwardCache() / popBackwardCache() / clearBackwardCache()
 *                                     envs/redmax_torch_functions.py:65,81; envs/tactile_insertion_env.py:226
 * LIFO of tapes so that several episodes can be forwarded before their backwards.  The tapes are swapped by pointer:
 * save parks the live tape buffer on the stack and continues on a spare one (the current state is carried over), pop
 * makes the newest saved tape the live one again (its length and recording flag with it; the carried adjoint restarts at
 * zero) and returns the buffer it replaces to the pool.  No tape copy, no synchronisation; the only allocation is the first
 * use of a stack depth, which tsim_cache_reserve(depth) moves out of the hot path.  clear drops the saved tapes (their
 * buffers go back to the pool; two spares are kept).  All calls of one batch on ONE stream. */
int tsim_cache_save(tsim_batch* b, void* stream);
int tsim_cache_pop(tsim_batch* b, void* stream);
int tsim_cache_clear(tsim_batch* b);
int tsim_cache_reserve(tsim_batch* b, int depth);     /* no reference counterpart: pre-allocate `depth` tape buffers */
int tsim_cache_depth(const tsim_batch* b);            /* saved tapes on the stack */

/* Diagnostics (no reference counterpart): one residual evaluation g(q1; q0, qd0, u) and its Newton matrix
 * H = dg/dq1 for env 0..B-1; g_out [B][nr], H_out [B][nr][nr] (row-major). Used by the parity tests.
 * cycles (device int64 [B][32], may be NULL): shader-clock stamps taken inside the evaluation (models with ndof_r <= 8). */
int tsim_debug_eval(tsim_batch* b, const void* q1, const void* q0, const void* qd0, const void* u,
                    void* g_out, void* H_out, long long* cycles, void* stream);

/* Diagnostics (no reference counterpart; SURVEY.md §7 "Non-smoothness"): branch signature of the taped sub-steps
 * t_first+1 .. t_first+n (1 = the first sub-step after reset) of every environment: which contact points and taxels
 * penetrate and on which smooth piece of the penalty law each of them is (stick / slip, face of the primitive), recomputed
 * from the taped state exactly as the adjoint re-evaluates it.  out: DEVICE uint32 [n][B][2] = (number of penetrating
 * items, sum over them of mix(item, branch) mod 2^32; the oracle states the same two numbers).  Two runs with equal
 * signatures went through the same smooth pieces of the dynamics — the fp32 / fp64 gradient comparison is made on those
 * environments, and the fraction that differs is reported (tests/test_gpu_configs.py).  Needs reset(backward_flag=True). */
int tsim_debug_signature(tsim_batch* b, int t_first, int n, uint32_t* out, void* stream);

/* Diagnostics: the adjoint kernels of this batch record shader-clock stamps of the first sub-steps of wavefront 0 in `cycles`
 * (DEVICE int64[32], zero it first; NULL switches it off): loop top | tape record in LDS | phase 1 | output partials | solve |
 * contacts (3 stamps per pair group) | projection | mass product. */
int tsim_debug_stamps(tsim_batch* b, long long* cycles);

/* launch statistics of the most recent kernels (HIP events are the caller's business; this only reports
 * static launch geometry of the forward / backward kernels): out[0] = LDS bytes per block, out[1] = threads per block,
 * out[2] = blocks, out[3] = lanes per environment (64 / 32 / 16: a 64-lane block carries 1 / 2 / 4 environments; chosen
 * from the batch size and the device's SIMD count, override with the environment variable TSIM_LPE at
 * tsim_batch_create). out = int32[4]. */
int tsim_launch_info(const tsim_batch* b, int32_t* out);
/* Measurement (no reference counterpart; what bench.py's per-kernel roofline is timed with): while enabled, every launch of the simulation
 * kernels by this batch — k_forward, the tactile read-out kernel that follows it (k_taxels / k_taxels_small), k_backward — is bracketed by a
 * pair of HIP events recorded on the stream the kernel is launched on (never inside a stream capture).  tsim_kernel_times waits for the
 * recorded pairs, returns per kind the summed milliseconds and the number of launches since the previous call (HOST double[TSIM_KT_COUNT],
 * int32[TSIM_KT_COUNT]) and forgets them; at most 65 536 launches are kept between two calls (later ones are not timed). */
enum { TSIM_KT_FORWARD = 0, TSIM_KT_TAXELS = 1, TSIM_KT_BACKWARD = 2, TSIM_KT_COUNT = 3 };
int tsim_kernel_timing(tsim_batch* b, int enable);
int tsim_kernel_times(tsim_batch* b, double* ms_sum, int32_t* launches);
/* Force 16 / 32 / 64 lanes per environment (0 = automatic again).  For callers that split a batch into groups on several
 * streams: each group is then small, but the groups together should still fill the device. */
int tsim_set_lanes_per_env(tsim_batch* b, int lanes);   /* host-side only: takes effect with the next launch */
/* Statically known models.  The library carries, next to the generic kernels, instantiations of the forward / adjoint kernels for models
 * whose compiled blob it was built with (csrc/tsim_static.h; round 4: TactilePush, envs/assets/pusher/pusher.xml): tree, joint types, joint
 * frames and axes, contact pairs, dof records are compile-time constants there and the whole residual evaluation is one register-resident
 * pass (csrc/tsim_static_eval.h).  They are used when the
 * batch's blob equals the compiled-in one (checked at tsim_batch_create / tsim_update_model: ints and float records bit for bit — for an fp32
 * batch the records AS FLOATS, which is all its kernels ever see of them; round 5 added fp64 instantiations) and the batch has no
 * per-environment tables; results equal the generic kernels' to fp32 rounding (tests/test_gpu_static_model.py).
 * tsim_static_model returns the id of the instantiation the NEXT launch will use (0: generic, 1: TactilePush); tsim_set_static(b, 0)
 * keeps a batch on the generic kernels (also: environment variable TSIM_NO_STATIC at creation), tsim_set_static(b, 1) allows them again. */
int tsim_static_model(const tsim_batch* b);
int tsim_set_static(tsim_batch* b, int allow);
/* Name of the kernel instantiation the NEXT forward / adjoint launch of this batch uses, for measurement records (bench.py
 * `kernel_instantiation`): "generic", "static:<model>" (every float of the model a compile-time constant) or "param:<model>" (round 5: the
 * model's STRUCTURE — tree, joint types, contact pairs, which joint frames are identities and which axes unit vectors — is compiled in, its
 * parameters are read from the batch's float records, shared or per environment: the instantiation survives tsim_update_model and
 * tsim_set_env_tables as long as the structure does).  Returns a pointer into static storage. */
const char* tsim_kernel_variant(const tsim_batch* b);
/* Host-side switches of a batch (take effect with the next launch).  No reference counterpart.
 *   TSIM_OPT_PAIR_CULL  (default 1; environment variable TSIM_NO_PAIR_CULL=1 at creation: 0)  a contact pair whose points' bounding sphere is
 *                       out of reach of its primitive in every environment of a wavefront is skipped by the residual evaluation, and the
 *                       generic fp32 kernels test a point's fp32 distance before its double-precision one.  Exact: what is skipped would
 *                       have contributed zeros (tests/test_gpu_exact_options.py asserts equal outputs with the option off). */
/*   TSIM_OPT_VALUE_TRIALS (default 2; environment variable TSIM_VALUE_TRIALS=n at creation; 0: off)  a line-search trial only needs ||g||: after
 *                       this many rejected trials of a Newton iteration the further ones evaluate the residual WITHOUT its tangents (about half
 *                       the work) whenever no environment of the wavefront needs a Newton matrix from that round; a trial that is taken is
 *                       re-evaluated in full first.  Exact: iterates, convergence flags and the taped matrices are those of the loop without
 *                       the option (tests/test_gpu_exact_options.py); tsim_last_evals counts trial points either way. */
/*   TSIM_OPT_TRIAL_HELPERS (default 1; environment variable TSIM_NO_TRIAL_HELPERS=1 at creation: 0)  a launch lasts as long as its slowest
 *                       environment's chain of evaluations, and those chains are long line searches.  The slots of a wavefront whose own
 *                       environments are finished evaluate the NEXT trial points (dlbase + 2^-t dq: known in advance) of a slot that is still
 *                       in a line search; the owner judges the results in the order and with the decision code of the sequential loop.
 *                       Exact: iterates, convergence flags, taped matrices and tsim_last_evals are those of the loop without helpers
 *                       (tests/test_gpu_exact_options.py); a line search of n trials takes ceil(n / (1 + helpers)) rounds. */
/*   TSIM_OPT_VALUE_FIRST (default 1; environment variable TSIM_NO_VALUE_FIRST=1 at creation: 0)  launches that record no tape (roll-out collection:
 *                       tsim_reset with backward_flag 0) never use the Newton matrix of a sub-step's FINAL iterate.  Where the previous sub-step
 *                       of an environment converged in one Newton step, the first trial of the next one evaluates the residual without its
 *                       tangents; it is taken as it is if it ends the sub-step, and evaluated again in full otherwise.  Exact, as above. */
/* Read-only through tsim_get_option (set them with tsim_set_solver_options): TSIM_OPT_CROSS_KINKS, TSIM_OPT_EVAL_BUDGET; and TSIM_OPT_ALL_DEFAULT = 1 iff
 * every solver / scheduling option of the batch is at its default (cross_kinks 1, eval_budget 0, value_trials 2, trial_helpers 1, value_first 1) — the
 * forward launch of an fp32 batch on a compiled-in model at four environments per wavefront then runs an instantiation that has them as compile-time
 * constants (same results bit for bit, ~2 % faster; environment variable TSIM_NO_DEFAULT_OPTS=1 at creation: never). */
enum { TSIM_OPT_PAIR_CULL = 1, TSIM_OPT_VALUE_TRIALS = 2, TSIM_OPT_TRIAL_HELPERS = 3, TSIM_OPT_VALUE_FIRST = 4, TSIM_OPT_CROSS_KINKS = 5, TSIM_OPT_EVAL_BUDGET = 6,
       TSIM_OPT_ALL_DEFAULT = 7 };
int tsim_set_option(tsim_batch* b, int option, int value);
int tsim_get_option(const tsim_batch* b, int option);

/* residual evaluations each environment spent in the most recent tsim_step (HOST int32[B]); synchronises. */
int tsim_last_evals(tsim_batch* b, int32_t* host_out);
/* ... and how many of those trial points were evaluated by a helper slot (TSIM_OPT_TRIAL_HELPERS) instead of in a round of the environment's own
 * (HOST int32[B]); synchronises.  No reference counterpart (diagnostics: tests assert that the helper path ran, profiles report its share). */
int tsim_last_helper_trials(tsim_batch* b, int32_t* host_out);

/* The Newton iteration of a sub-step is the loop the model's <solver_option tol max_iter max_ls> states (envs/assets/pusher/pusher.xml:4)
 * and nothing else: up to max_iter iterations, each halving its step until ||g|| decreases, at most max_ls times, taking the last trial
 * if none did; converged when ||g||_2 < tol.  fp64 batches run exactly that by default.  Two options around it, neither of which the
 * reference has (its arithmetic is fp64 and it runs one environment at a time):
 *   cross_kinks  (default: 1 for TSIM_F32, 0 for TSIM_F64)  a line search whose max_ls + 1 trials all fail is a non-smooth local minimum
 *                of ||g|| at a contact / friction kink.  The literal loop gets across it with its 2^-max_ls step and ~150 more
 *                evaluations in fp64; in fp32 it often does not (step lengths of 1e-6 drown in rounding) and runs to max_iter while the
 *                rest of the batch waits.  With the option, and only when ||g|| < TSIM_KINK_FACTOR x tol (close to convergence, where the
 *                Newton step is small), a trial still rejected after TSIM_KINK_LS halvings is followed by the full Newton step across
 *                the kink, at most TSIM_KINK_MAX times per sub-step: the same root, ~20 evaluations.  Everywhere else the loop is the
 *                literal one.
 *   eval_budget  (default 0 = none)  residual evaluations one sub-step may take; a sub-step cut short is flagged in status exactly like
 *                one that hit max_iter.  For throughput-minded roll-out collection on stiff models (a TactileInsertion grasp can make
 *                plain backtracking creep for ~2000 evaluations, and a batch waits for its slowest environment).
 * Host-side only: takes effect with the next launch. */
#define TSIM_KINK_FACTOR 1000.0
#define TSIM_KINK_LS 4
#define TSIM_KINK_MAX 6
int tsim_set_solver_options(tsim_batch* b, int cross_kinks, int eval_budget);
/* largest ||g||_2 any sub-step of the most recent forward launch ended with, per environment (HOST float[B]); synchronises. */
int tsim_last_gnorm(tsim_batch* b, float* host_out);

const char* tsim_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
