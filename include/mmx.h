/*
 * mmx.h -- C ABI of the MI355X-native batched inverse-kinematics hot path.
 *
 * This is the drop-in boundary for ONE path of facebookresearch/momentum: the
 * per-iteration  FK -> Jacobian/residual assembly -> (JtJ + lambda I) d = Jtr
 * -> theta update  loop, batched over independent characters.  Every entry point
 * cites the reference interface it replaces (paths relative to the reference
 * checkout).  Plain pointers and sizes only; no C++/torch types; status-code
 * returns, no exceptions across the ABI (the C++ shell in
 * include/momentum_amd/ converts non-zero into std::runtime_error to match
 * MT_CHECK/MT_THROW, momentum/common/checks.h:36-45, exception.h:24-66).
 *
 * Conventions shared with the reference:
 *   - quaternions are (x,y,z,w) (Eigen storage order; tensor API
 *     pymomentum/tensor_ik/tensor_error_function_utility.h:58-66),
 *   - joint parameter order per joint is tx,ty,tz,rx,ry,rz,scale
 *     (momentum/character/types.h:21-27, kParametersPerJoint = 7),
 *   - joints are listed parent-before-child (momentum/character/skeleton.cpp:16-22),
 *   - the dense Jacobian of one instance is COLUMN-MAJOR M x P
 *     (Eigen default; momentum/math/resizeable_matrix.h:18,32-34), rows ordered
 *     [position block (3 rows / constraint)] then [orientation block (9 rows /
 *     constraint)] = order of addErrorFunction
 *     (momentum/character_solver/skeleton_solver_function.cpp:217-261).
 *
 * Threading: one handle = one device + caller-provided stream; calls on one
 * handle must be serialised by the caller (the reference's solver objects are
 * not thread-safe either, momentum/math/resizeable_matrix.h:36-38).
 *
 * Streams and graphs: mmx_solve* / mmx_eval_* enqueue kernels and device-to-device
 * copies on `stream` and return; they neither wait for the stream nor read results
 * back (exceptions, each said at its declaration: the *_host conveniences, the
 * timed debug entry points, MMX_STEP_TRUST_REGION on the wide route).  After one
 * warm-up call on a handle (scratch buffers, LDS limits) such a call may therefore
 * be recorded into a HIP graph (hipStreamBeginCapture / torch.cuda.CUDAGraph) and
 * replayed on new values in the same buffers -- tests/test_gpu_graph.py; the
 * library issues kernel nodes only, no memset nodes.
 */
#ifndef MMX_H_
#define MMX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMX_ABI_VERSION 11 /* 6: per-instance characters and constraint parents, MMX_STEP_TRUST_REGION (+ mmx_gn_options::
                             trust_region_radius), mmx_comm_* (RCCL), MMX_LIMIT_MINMAX_JOINT_PASSIVE, row-major J
                             7: mmx_tuning / mmx_problem_set_tuning / mmx_problem_last_route (replace the MMX_* environment
                             switches of earlier builds: the library reads no environment variable on the solve path)
                             8: columns in elimination order + tile-sparse factor on the wide route
                                (mmx_host_elimination_order, mmx_host_tile_structure, mmx_problem_tile_structure)
                             9: status[] is a bit set (MMX_SOLVE_DAMPING_FLOORED, informational), sized form of the
                                constraint call (mmx_problem_set_constraints_sized), mmx_eval_skeleton_state_host.
                             10: mmx_gn_options grows by `precision` / `precision_bound` (MMX_PRECISION_*: single, double, or
                                single with the elements whose precision estimate exceeds the bound re-solved in double),
                                status bits MMX_SOLVE_PRECISION_SUSPECT / MMX_SOLVE_ESCALATED_F64, MMX_SOLVE_FAILED(),
                                mmx_solve_with_step_history (damping and gain ratio per iteration of the LM schedule),
                                mmx_problem_solve_diagnostics.
                             11: MMX_PRECISION_MIXED (double forward kinematics / residuals / g = J^T r / linear-solve
                                residual around the single-precision factor: the double instantiation's answers at close to
                                the single-precision rate), which is also what MMX_PRECISION_AUTO now escalates to where it
                                applies; status bit MMX_SOLVE_MIXED; mmx_tuning grows mixed_tolerance / mixed_max_cg out of
                                its reserved words (same size).
                             A caller MUST compare mmx_abi_version() with the MMX_ABI_VERSION it was compiled against
                             before any other call: the structs below grow at their end from version to version. */
#define MMX_PARAMS_PER_JOINT 7 /* momentum/character/types.h:21 */
#define MMX_INVALID_PARENT (-1) /* kInvalidIndex, momentum/character/types.h:182 */
#define MMX_MAX_MODEL_PARAMS 2048 /* kMaxModelParams, momentum/math/types.h:426 */

typedef enum mmx_status {
  MMX_OK = 0,
  MMX_ERR_INVALID_ARGUMENT = 1, /* MT_CHECK failure class */
  MMX_ERR_SIZE_MISMATCH = 2, /* solver.cpp:77, parameter_transform.cpp:112-121 */
  MMX_ERR_DEVICE = 3, /* HIP runtime error (message in mmx_last_error) */
  MMX_ERR_UNSUPPORTED = 4, /* shape outside what the kernels were built for */
  MMX_ERR_OUT_OF_MEMORY = 5,
  MMX_ERR_NO_DEVICE = 6 /* no gfx950 device: the product path never falls back to CPU */
} mmx_status;

/* Per-instance solve status (mmx_solve: status[B]). */
#define MMX_SOLVE_OK 0
#define MMX_SOLVE_NONFINITE 1 /* NaN/Inf result -> theta reverted to theta_init
                                 (pymomentum/tensor_ik/tensor_ik.cpp:168-173) */
#define MMX_SOLVE_NOT_PD 2 /* a Cholesky pivot came out non-positive in single precision (J rank deficient and
                              lambda below the rounding of J^T J).  Informational: a pivot at or below 2^-18 of its
                              row's H_jj + lambda drops that column from the iteration's step (zero step in that
                              parameter, the others solve the reduced system) and the step IS taken -- like the
                              reference, which never checks LLT::info() (gauss_newton_solver.cpp:251); its double
                              instantiation does not meet such pivots, its float instantiation hands Eigen's aborted
                              factor to solve().  The objective converges like the reference's; the components of
                              theta that J does not determine differ from the double solver's minimum-norm ones
                              (tests/test_gpu_weak_damping.py).  Since ABI 8 every single-precision
                              FACTOR is damped by at least 1e-5 of the mean diagonal of J^T J (the refinement measures
                              its residual with the caller's lambda through J, so the step is the caller's wherever J
                              determines it): the drop rule is the last resort and this status is rare. */

#define MMX_SOLVE_DAMPING_FLOORED 4 /* (bit, informational; since ABI 9 status[] is a bit set and MMX_SOLVE_NONFINITE /
                                       MMX_SOLVE_NOT_PD are its error bits, MMX_SOLVE_ERROR_MASK) in at least one
                                       iteration the caller's damping was below the floor of the single-precision
                                       FACTOR, 1e-5 of the mean diagonal of J^T J (the note above): what was factored is
                                       J^T J + floor I, the refinement pulled the step towards the caller's damping
                                       wherever J determines it, and the predicted decrease of the LM schedule / trust
                                       region was formed with the caller's value.  Where J is rank deficient or barely
                                       determined (BASELINE configs[0], configs[1] below lambda ~ 1e-3) such a solve
                                       is NOT within 1e-5 of the reference's double instantiation on the pose
                                       parameters -- no single-precision normal-equation solver is; the answer there
                                       is mmx_solve_f64.  mmx_solve_f64 never sets this bit. */
#define MMX_SOLVE_PRECISION_SUSPECT 8 /* (bit, informational, ABI 10) the single-precision solve's own estimate of its
                                         distance from the same solve in double -- 0.042 eps x max over the iterations of
                                         sqrt(error of the iteration / error of the first) / (smallest Cholesky pivot
                                         ratio d_jj / (H_jj + lambda) of the iteration) ~ eps x cond(J^T J + lambda I) x the
                                         residual's share (ABI 11; ABI 10 weighted every iteration with 1),
                                         calibrated as the 98th percentile of the relative distance on the BASELINE shapes;
                                         mmx_problem_solve_diagnostics returns it -- exceeds mmx_gn_options::precision_bound
                                         (default 1e-5, north_star's parity bound: 1 / ratio = 2000).  Unlike
                                         MMX_SOLVE_DAMPING_FLOORED, which only says that the factor's damping floor engaged,
                                         this follows the conditioning that loses the digits; it is a property of the problem
                                         class (rig, constraint set, lambda) more than of the instance: a class in which a
                                         few per cent of the instances leave the bound is marked as a whole.  With MMX_PRECISION_AUTO such elements (and the ones with an
                                         error bit) are solved again by the double instantiation from the initial parameters.
                                         One-launch and wide routes; the explicit-Jacobian route keeps
                                         MMX_SOLVE_DAMPING_FLOORED as its cue. */
#define MMX_SOLVE_ESCALATED_F64 16 /* (bit, informational, ABI 10) MMX_PRECISION_AUTO: this element's result comes from the
                                      double instantiation (its other status bits are that run's, plus the
                                      MMX_SOLVE_PRECISION_SUSPECT that sent it there) */
#define MMX_SOLVE_MIXED 32 /* (bit, informational, ABI 11) this element's result comes from the mixed-precision instantiation
                              (MMX_PRECISION_MIXED, or MMX_PRECISION_AUTO's second pass).  With it, MMX_SOLVE_PRECISION_SUSPECT
                              means that some iteration's conjugate gradients ran into mixed_max_cg before they met
                              mixed_tolerance (the single-precision factor was no preconditioner any more: cond x eps ~ 1). */
#define MMX_SOLVE_ERROR_MASK 3 /* (status & MMX_SOLVE_ERROR_MASK) == 0: the solve of that element is sound.  Mind the
                                  parentheses: in C `status & MMX_SOLVE_ERROR_MASK == 0` parses as status & (3 == 0). */
#define MMX_SOLVE_FAILED(status) (((status) & MMX_SOLVE_ERROR_MASK) != 0)

/* mmx_gn_options::precision (ABI 10). */
#define MMX_PRECISION_F32 0 /* the single-precision kernels (GaussNewtonSolverT<float>; what BASELINE's metric times) */
#define MMX_PRECISION_F64 1 /* mmx_solve with float parameters in and out, every element solved by the double instantiation
                               (GaussNewtonSolverT<double>, momentum/solver/gauss_newton_solver.cpp:315-316) */
#define MMX_PRECISION_AUTO 2 /* single precision first; the elements it marks MMX_SOLVE_PRECISION_SUSPECT (or whose solve
                                failed) are compacted and solved again in double from the initial parameters, on the same
                                stream, without a host round trip.  The batched driver's defaults (lambda = 0.01,
                                pymomentum/tensor_ik/solver_options.h:28-37) on marginally determined problems are where this
                                matters (profiles/r05_weak_damping.json).  ABI 11: the second pass is the
                                mixed-precision instantiation wherever that applies (MMX_SOLVE_MIXED instead of
                                MMX_SOLVE_ESCALATED_F64), and a problem class the FIRST factorisation already marks leaves the
                                single-precision pass at once (no ten iterations thrown away). */
#define MMX_PRECISION_MIXED 3 /* (ABI 11) one launch per solve like MMX_PRECISION_F32, with everything the answer's digits
                                 depend on in double: theta, forward kinematics, residual rows, g = J^T r and the residual of
                                 the linear solve -- all O(joints + constraints) tree passes -- while H = J^T J, its Cholesky
                                 factor and the triangular solves stay single precision and act as the preconditioner of a
                                 conjugate-gradient iteration whose operator is applied in double through the tree (classic
                                 mixed-precision iterative refinement; accurate while cond(J^T J + lambda I) x 6e-8 < 1).
                                 Follows GaussNewtonSolverT<double> (momentum/solver/gauss_newton_solver.cpp:315-316) to ~1e-7
                                 on theta -- at the batched driver's lambda = 0.01 and far below -- at a multiple of the double
                                 instantiation's rate.  Scope: the one-launch route's problems (up to 128 solved parameters,
                                 256 joints) with position / orientation constraints, limits on model / joint parameters and
                                 the model-parameter prior, every step rule but the trust region;
                                 anything else is solved by the double instantiation exactly as under MMX_PRECISION_F64 (no
                                 MMX_SOLVE_MIXED bit on the status). */

/* Where the caller's bulk arrays live. */
#define MMX_MEM_HOST 0
#define MMX_MEM_DEVICE 1

/* Jacobian layouts of mmx_eval_jacobian. */
#define MMX_LAYOUT_COL_MAJOR 0 /* J[b][p*M + i]   (the reference's layout) */
#define MMX_LAYOUT_ROW_MAJOR 1 /* J[b][i*P + p]   (torch-style [B][M][P]; assembled column-major, then transposed:
                                  one extra read + write of J; mmx_eval_jacobian[_host] only) */

/* Solver step rule. */
#define MMX_STEP_GN_FIXED_LAMBDA 0 /* GaussNewtonSolverT, constant regularization
                                      (momentum/solver/gauss_newton_solver.cpp:224-280) */
#define MMX_STEP_LM_SCHEDULE 1 /* gain-ratio lambda schedule, a lambda-form of
                                  momentum/character_solver/trust_region_qr.cpp:244-268
                                  (BASELINE configs[2]; no direct reference implementation) */
#define MMX_STEP_TRUST_REGION 2 /* TrustRegionQRT::doIteration (momentum/character_solver/trust_region_qr.cpp:
                                   52-270; LinearSolverType::TrustRegionQR of the batched driver, tensor_ik.cpp:
                                   150-152): per iteration up to ten trial steps, each after up to three Newton
                                   updates of the damping that pull |step| towards the trust radius (:180-231),
                                   gain ratio against the quadratic model (:246-247), radius x 0.25 / x 2 (cap 10)
                                   (:256-262), a step with rho <= 0 is rejected (:265-269).  Every change of the
                                   damping costs a factorisation (the reference appends rows to its QR).  Runs
                                   inside the one-launch solve where the problem fits it (<= 224 solved parameters,
                                   position / orientation constraints, limits, model prior) and on the wide route
                                   otherwise (driven from the host: factor / decide / trial kernels per trust step;
                                   larger systems, further joint error functions, ellipsoid limits); not on
                                   MMX_ROUTE_EXPLICIT_JACOBIAN.  mmx_solve_f64 runs the rule in double (one LL^T of
                                   J^T J + damping per value of the damping; 1e-7 on the oracle's double run where J
                                   has full column rank).  do_line_search and regularization are not read by this rule.
                                   DEVIATION from TrustRegionQRT: the reference's Householder QR of J takes the rule's
                                   (almost) zero damping on rank-deficient / under-determined J; an fp32 Cholesky of
                                   J^T J cannot, so the factor is damped by at least 1e-5 of the mean diagonal of J^T J
                                   (the refinement keeps the rule's damping): in the directions J does not determine
                                   the step is the more damped one -- the trust-region logic (radius, gain ratio,
                                   accept / reject) runs on that step; objective and iteration counts follow the
                                   oracle's on the reference's fixtures (tests/test_gpu_trust_region.py), the
                                   undetermined components of theta need not. */

/*
 * Static rig = Skeleton + ParameterTransform of a momentum::Character
 * (momentum/character/character.h:32-125; only .skeleton and
 * .parameterTransform are read on this path,
 * skeleton_solver_function.cpp:30-33).  All pointers are HOST pointers and are
 * copied by mmx_rig_create.
 */
typedef struct mmx_rig_desc {
  int32_t num_joints; /* J  = skeleton.joints.size() */
  int32_t num_params; /* P  = parameterTransform.numAllModelParameters() */
  const int32_t* parent; /* [J]  Joint::parent, -1 for kInvalidIndex (joint.h:18-36) */
  const float* pre_rotation; /* [J][4] Joint::preRotation (x,y,z,w) */
  const float* translation_offset; /* [J][3] Joint::translationOffset */
  /* ParameterTransform::transform, SparseRowMatrix 7J x P == CSR
     (momentum/character/parameter_transform.h:62-95, math/types.h:191) */
  const int32_t* pt_outer; /* [7J+1] outerIndexPtr */
  const int32_t* pt_inner; /* [nnz]  innerIndexPtr (column = model parameter) */
  const float* pt_value; /* [nnz]  valuePtr */
  const float* pt_offsets; /* [7J]   ParameterTransform::offsets (may be NULL = zeros) */
} mmx_rig_desc;

/*
 * Per-instance constraint payload = the std::vector<PositionDataT>/
 * <OrientationDataT> of one PositionErrorFunction + one OrientationErrorFunction
 * per batch element (momentum/character_solver/position_error_function.h:16-29,
 * orientation_error_function.h:16-36, error_function_types.h:34-44).  The parent
 * joint of each constraint is given at mmx_problem_create (batch-shared) or per
 * element with mmx_problem_set_instance_parents.  Orientation quaternions are normalised on ingest like the
 * OrientationDataT constructor does (orientation_error_function.h:33-35).
 */
/*
 * One entry of Character::parameterLimits restricted to the limit types that act on model or
 * joint PARAMETERS (momentum/character/parameter_limits.h:20-31,33-99,125-136): the rows of
 * LimitErrorFunctionT that need no joint transforms.  Ellipsoid limits travel in
 * mmx_ellipsoid_limit; the passive MinMaxJointPassive type is accepted and ignored like in the reference.
 */
#define MMX_LIMIT_MINMAX 0 /* LimitType::MinMax    : index0 = parameterIndex ; v = {min, max} */
#define MMX_LIMIT_MINMAX_JOINT 1 /* LimitType::MinMaxJoint : index0 = 7 * jointIndex + jointParameter (a row of the
                                    parameter transform) ; v = {min, max} on that JOINT parameter */
#define MMX_LIMIT_MINMAX_JOINT_PASSIVE 2 /* LimitType::MinMaxJointPassive: accepted and IGNORED -- LimitErrorFunctionT gives it
                                            neither an error term nor a Jacobian row (limit_error_function.cpp:836-837,
                                            1051-1052); such entries do not count in mmx_problem_num_rows */
#define MMX_LIMIT_LINEAR_JOINT 4 /* LimitType::LinearJoint : index0 = 7 * referenceJointIndex + referenceJointParameter,
                                    index1 = 7 * targetJointIndex + targetJointParameter ;
                                    v = {scale, offset, rangeMin, rangeMax} */
#define MMX_LIMIT_LINEAR 3 /* LimitType::Linear    : index0 = referenceIndex, index1 = targetIndex ;
                              v = {scale, offset, rangeMin, rangeMax}  (p_ref = scale * p_target - offset) */
#define MMX_LIMIT_HALFPLANE 6 /* LimitType::HalfPlane : index0 = param1, index1 = param2 ;
                                 v = {normal[0], normal[1], offset} */
#define MMX_LOSS_WELSCH (-3.402823466e+38f) /* GeneralizedLossT::kWelsch = numeric_limits<float>::lowest() */

typedef struct mmx_parameter_limit {
  int32_t type; /* MMX_LIMIT_* (values of momentum::LimitType) */
  int32_t index0;
  int32_t index1;
  float weight; /* ParameterLimit::weight */
  float v[4];
} mmx_parameter_limit;

/*
 * LimitType::Ellipsoid of Character::parameterLimits (momentum/character/parameter_limits.h:77-84):
 * the point `offset` of joint `parent` is pulled onto the ellipsoid (unit sphere mapped by `ellipsoid`)
 * defined in the frame of joint `ellipsoid_parent`.  Three rows per entry, evaluated like
 * computeEllipsoidError / computeEllipsoidJacobian (limit_error_function.cpp:173-195,702-790): the
 * Jacobian walks parent -> ellipsoid_parent (exclusive) and treats the projected point as constant,
 * weight kLimitWeight * weight_ * kPositionWeight(1e-4) * weight.  Affine maps are 3 x 4 row-major
 * [linear | translation].  Batch-shared, HOST array.  NB: no test of the reference exercises this
 * limit type, so parity for it rests on the restatement alone.
 */
typedef struct mmx_ellipsoid_limit {
  float ellipsoid[12]; /* LimitEllipsoid::ellipsoid */
  float ellipsoid_inv[12]; /* LimitEllipsoid::ellipsoidInv (its inverse) */
  float offset[3]; /* LimitEllipsoid::offset */
  float weight; /* ParameterLimit::weight */
  int32_t ellipsoid_parent;
  int32_t parent;
} mmx_ellipsoid_limit;

/*
 * The other JointErrorFunctionT specialisations: one block = one error
 * function object with `count` constraints per batch element.  FuncDim rows per constraint.
 *   type                      reference (momentum/character_solver/...)            rows  payload used
 *   MMX_JC_PLANE              PlaneErrorFunctionT(above=false) plane_error_function.cpp:52-71   1  local_point=offset, global=normal (normalised), plane_d
 *   MMX_JC_HALF_PLANE         PlaneErrorFunctionT(above=true)  (f clamped to <= 0, :63-70)      1  as above
 *   MMX_JC_AIM_DIST           AimDistErrorFunctionT aim_error_function.cpp:15-36                3  local_point, local_dir (normalised), global=globalTarget
 *   MMX_JC_AIM_DIR            AimDirErrorFunctionT  aim_error_function.cpp:39-67                3  as above
 *   MMX_JC_FIXED_AXIS_DIFF    FixedAxisDiffErrorFunctionT  fixed_axis_error_function.cpp:15-26  3  local_dir=localAxis, global=globalAxis (both normalised)
 *   MMX_JC_FIXED_AXIS_COS     FixedAxisCosErrorFunctionT   :28-39                               1  as above
 *   MMX_JC_FIXED_AXIS_ANGLE   FixedAxisAngleErrorFunctionT :41-66                               1  as above
 *   MMX_JC_NORMAL             NormalErrorFunctionT normal_error_function.cpp:14-31              1  local_point, local_dir=localNormal (normalised), global=globalPoint
 * Vectors are normalised on ingest exactly where the reference's data constructors do
 * (plane_error_function.h:30, aim_error_function.h:34, fixed_axis_error_function.h:29-30,
 * normal_error_function.h:34).
 */
#define MMX_JC_PLANE 0
#define MMX_JC_HALF_PLANE 1
#define MMX_JC_AIM_DIST 2
#define MMX_JC_AIM_DIR 3
#define MMX_JC_FIXED_AXIS_DIFF 4
#define MMX_JC_FIXED_AXIS_COS 5
#define MMX_JC_FIXED_AXIS_ANGLE 6
#define MMX_JC_NORMAL 7
#define MMX_MAX_JOINT_BLOCKS 8

typedef struct mmx_joint_constraint_block {
  int32_t type; /* MMX_JC_* */
  int32_t count; /* constraints per batch element */
  const int32_t* parent; /* [count] HOST, batch-shared: ConstraintData::parent */
  const float* local_point; /* [B][count][3] or NULL when the type has no point */
  const float* local_dir; /* [B][count][3] or NULL when the type has no direction */
  const float* global; /* [B][count][3] */
  const float* plane_d; /* [B][count] (plane types only) */
  const float* weight; /* [B][count] ConstraintData::weight */
  float function_weight; /* SkeletonErrorFunction::weight_ */
  float loss_alpha, loss_c; /* GeneralizedLossT(alpha, c); c <= 0: default L2, c = 1 */
} mmx_joint_constraint_block;

typedef struct mmx_constraint_data {
  const float* pos_offset; /* [B][Kp][3] */
  const float* pos_target; /* [B][Kp][3] */
  const float* pos_weight; /* [B][Kp]    ConstraintData::weight */
  const float* ori_offset; /* [B][Ko][4] (x,y,z,w) */
  const float* ori_target; /* [B][Ko][4] (x,y,z,w) */
  const float* ori_weight; /* [B][Ko] */
  float pos_function_weight; /* SkeletonErrorFunction::weight_ of the position block
                                (skeleton_error_function.h:44-141, setWeight) */
  float ori_function_weight; /* ... of the orientation block */
  int32_t memory; /* MMX_MEM_HOST: copied; MMX_MEM_DEVICE: borrowed, caller keeps alive */
  /* ---- optional parameter-space blocks (all zero / NULL = absent); rows follow the
     orientation rows: [3 Kp][9 Ko][num_limits][P if model_target != NULL] */
  /* ModelParametersErrorFunctionT::setTargetParameters (model_parameters_error_function.h:48-51) */
  const float* model_target; /* [B][P] targetParameters_, same memory kind as above */
  const float* model_weights; /* [B][P] targetWeights_ */
  float model_function_weight; /* weight_ of that block */
  /* LimitErrorFunctionT::setLimits (limit_error_function.h:88-89): batch-shared, ALWAYS a host
     pointer (copied) */
  int32_t num_limits;
  const mmx_parameter_limit* limits; /* [num_limits] */
  float limit_function_weight; /* weight_ of that block */
  /* ---- robust loss of the two joint-constraint blocks: GeneralizedLossT(alpha, c)
     (momentum/math/generalized_loss.h:46-101; constructor arguments lossAlpha / lossC of
     PositionErrorFunctionT / OrientationErrorFunctionT, position_error_function.h:41-48).
     alpha = 2: L2, 1: L1 / pseudo-Huber, 0: Cauchy, MMX_LOSS_WELSCH: Welsch, other: Barron's
     general form.  c <= 0 (e.g. a zero-initialised struct) selects the default L2 loss with c = 1. */
  float pos_loss_alpha, pos_loss_c;
  float ori_loss_alpha, ori_loss_c;
  /* ---- optional further joint-constraint blocks (see mmx_joint_constraint_block); their rows
     follow the orientation rows and precede the limit rows:
     [3 Kp][9 Ko][block 0]...[block n-1][num_limits][P if model_target != NULL].
     `joint_blocks` is a HOST array (copied); the payload arrays inside follow `memory`.
     Problems with such blocks are solved by the explicit-Jacobian kernels. */
  int32_t num_joint_blocks; /* <= MMX_MAX_JOINT_BLOCKS */
  const mmx_joint_constraint_block* joint_blocks;
  /* ---- Ellipsoid entries of the limit block (same weight_ = limit_function_weight as `limits`);
     three rows each, placed between the joint-block rows and the other limit rows:
     [3 Kp][9 Ko][blocks][3 num_ellipsoid_limits][num_limits][P].  HOST array (copied). */
  int32_t num_ellipsoid_limits;
  const mmx_ellipsoid_limit* ellipsoid_limits;
  /* ---- per-element error-function weights = errorFunctionWeights[iBatch][weightsMap[iErr]] of the batched driver
     (pymomentum/tensor_ik/tensor_ik.cpp:100-101,137-138; tensor_ik_utility.cpp:162-177: every error function of
     element iBatch gets setWeight(that entry)).  NULL = none.  [B][num_function_weights] floats, same memory kind as
     the constraint arrays (device: borrowed), columns:
        0 position block   1 orientation block   2 limit block (parameter + ellipsoid limits)
        3 model-parameter block   4 + i joint block i        (columns past num_function_weights count as 1)
     Element b's weight_ of a block = the block's scalar function weight above x function_weights[b][column]; the
     driver's weights go here with the scalars left at 1.  A product <= 0 switches the block off for that element
     (skeleton_solver_function.cpp:223-231), like weightsMap[iErr] < 0 does (weight 0, tensor_ik_utility.cpp:176). */
  const float* function_weights;
  int32_t num_function_weights;
} mmx_constraint_data;

/*
 * POD mirror of SolverOptions + GaussNewtonSolverOptions
 * (momentum/solver/solver.h:19-34, gauss_newton_solver.h:17-59).
 */
/*
 * Backtracking rules of the reference's Gauss-Newton solvers (tau = 0.5, at most 10 trial steps, the
 * last trial stays when none passes):
 *   MMX_LINE_SEARCH_GAUSS_NEWTON  GaussNewtonSolverT::updateParameters (gauss_newton_solver.cpp:283-313):
 *                                 accept when  e - e(alpha) >= alpha * 1e-3 * e
 *   MMX_LINE_SEARCH_DIRECTIONAL   SubsetGaussNewtonSolverT::doIteration (subset_gauss_newton_solver.cpp:
 *                                 117-142) and GaussNewtonSolverQRT::doIteration (gauss_newton_solver_qr.cpp:
 *                                 126-149) -- the two solvers the batched driver builds
 *                                 (pymomentum/tensor_ik/tensor_ik.cpp:142-158): accept when
 *                                 e - e(alpha) >= 1e-4 * alpha * (J^T r . delta)
 */
#define MMX_LINE_SEARCH_NONE 0
#define MMX_LINE_SEARCH_GAUSS_NEWTON 1
#define MMX_LINE_SEARCH_DIRECTIONAL 2

typedef struct mmx_gn_options {
  int32_t min_iterations; /* SolverOptions::minIterations (default 1) */
  int32_t max_iterations; /* SolverOptions::maxIterations (default 2) */
  float threshold; /* SolverOptions::threshold (default 1): converged when
                      |e_prev-e|/(|e|+FLT_MIN) <= threshold*FLT_EPSILON, solver.cpp:98-99 */
  float regularization; /* GaussNewtonSolverBaseOptions::regularization (default 0.05) */
  int32_t do_line_search; /* MMX_LINE_SEARCH_* (default 0 = GaussNewtonSolverBaseOptions::doLineSearch false) */
  int32_t step_rule; /* MMX_STEP_* */
  /* LM schedule knobs (only read when step_rule == MMX_STEP_LM_SCHEDULE) */
  float lm_lambda_min; /* default 1e-6 */
  float lm_lambda_max; /* default 1e6 */
  float lm_up; /* lambda *= lm_up   when rho < 0.25 or the step is rejected (default 4) */
  float lm_down; /* lambda *= lm_down when rho > 0.75 (default 0.5) */
  /* trust region (only read when step_rule == MMX_STEP_TRUST_REGION) */
  float trust_region_radius; /* TrustRegionQROptions::trustRegionRadius_ (default 1; <= 0 selects the default) */
  /* ABI 10 */
  int32_t precision; /* MMX_PRECISION_* (default MMX_PRECISION_F32); read by mmx_solve / mmx_solve_with_history /
                        mmx_solve_with_step_history / mmx_solve_host, not by mmx_solve_f64 */
  float precision_bound; /* estimate above which an element is MMX_SOLVE_PRECISION_SUSPECT (default 1e-5; <= 0 selects the
                            default) */
} mmx_gn_options;

typedef struct mmx_rig mmx_rig; /* opaque: device-resident rig constants */
typedef struct mmx_problem mmx_problem; /* opaque: one batch of B instances on one device */

/* Sensible defaults == the reference's struct initialisers cited above. */
void mmx_gn_options_default(mmx_gn_options* opt);

/* ABI / build info. */
int32_t mmx_abi_version(void);
/* Message of the last failing call on this thread ("" if none). */
const char* mmx_last_error(void);
/* Number of visible HIP devices (0 when there is none; never throws). */
int32_t mmx_device_count(void);

/*
 * Replaces: constructing Skeleton + ParameterTransform for
 * SkeletonSolverFunctionT (skeleton_solver_function.cpp:25-39).  Validates the
 * parent-before-child invariant (skeleton.cpp:16-22) and CSR bounds, uploads the
 * rig and its derived integer tables to `device`.
 */
int32_t mmx_rig_create(const mmx_rig_desc* desc, int32_t device, mmx_rig** out);
void mmx_rig_destroy(mmx_rig* rig);
int32_t mmx_rig_num_joints(const mmx_rig* rig);
int32_t mmx_rig_num_params(const mmx_rig* rig);

/*
 * Replaces: one SkeletonSolverFunctionT + PositionErrorFunctionT +
 * OrientationErrorFunctionT per batch element, as built per task in
 * pymomentum/tensor_ik/tensor_ik.cpp:136-141.  pos_parent / ori_parent are HOST
 * arrays of joint indices (ConstraintData::parent).  All device scratch is sized
 * here, once.
 */
int32_t mmx_problem_create(
    mmx_rig* rig,
    int32_t batch,
    int32_t num_pos,
    const int32_t* pos_parent,
    int32_t num_ori,
    const int32_t* ori_parent,
    mmx_problem** out);
void mmx_problem_destroy(mmx_problem* problem);

/*
 * Per-instance characters of the same topology: the batched driver solves element iBatch on
 * *characters[iBatch] (pymomentum/tensor_ik/tensor_ik.cpp:129,140), in practice one skeleton scaled per
 * subject -- same parents and parameter transform, different Joint::translationOffset / preRotation.
 * translation_offset [B][J][3], pre_rotation [B][J][4] (x,y,z,w); either may be NULL (= the rig's own
 * values for every element); both NULL restores the shared rig.  `memory`: MMX_MEM_HOST (copied) or
 * MMX_MEM_DEVICE (borrowed, caller keeps alive).  Every entry point of the problem (FK, J assembly,
 * solve) then reads element b's constants.
 */
int32_t mmx_problem_set_instance_rig(
    mmx_problem* problem,
    const float* translation_offset,
    const float* pre_rotation,
    int32_t memory,
    void* stream);

/*
 * Per-instance constraint parents: ConstraintData::parent of element iBatch's constraint k, the way the
 * tensor error functions read `parents` per batch element (pymomentum/tensor_ik/
 * tensor_marker_error_function.cpp:97-98,186).  pos_parent [B][Kp], ori_parent [B][Ko] (joint indices);
 * either may be NULL (= the batch-shared list given at mmx_problem_create); both NULL restores the
 * shared lists.  `memory` as above.  The integer bookkeeping that depends on where constraints sit
 * (structurally zero columns, the solve list) is rebuilt for the UNION of the batch's parents -- a column
 * that is zero for one element only gets an exact zero step there, as in the reference.
 * Joint indices are validated (MT_CHECK joint_error_function-inl.h:230).
 */
int32_t mmx_problem_set_instance_parents(
    mmx_problem* problem,
    const int32_t* pos_parent,
    const int32_t* ori_parent,
    int32_t memory,
    void* stream);

/*
 * Which kernels mmx_solve runs.  The library picks by problem size (MMX_ROUTE_AUTO); a caller -- in practice a parity
 * test that wants every route exercised on the same inputs, or a benchmark -- can pin the route per problem handle.
 * A pinned route the problem does not fit makes mmx_solve return MMX_ERR_UNSUPPORTED; it never falls through silently.
 *   MMX_ROUTE_FUSED              one launch, one workgroup per instance, the system in LDS (<= 224 solved parameters)
 *   MMX_ROUTE_WIDE               normal equations from the tree moments, left-looking blocked Cholesky over the
 *                                structurally non-zero 16 x 16 tiles (columns in elimination order) -- the factor resident
 *                                in LDS while its tiles fit half a CU (<= ~75 tiles), in HBM beyond --, refinement through
 *                                the tree (<= 512 solved parameters; the default from 129 on)
 *   MMX_ROUTE_EXPLICIT_JACOBIAN  dense J in HBM -> J^T J (matrix cores; VALU beyond 384 columns) -> Cholesky step; the route
 *                                for problems outside the tree kernels' scope, among them systems of 513 ... 2048 solved
 *                                parameters (kMaxModelParams, momentum/math/types.h:426-429: a rig cannot have more)
 * The route does not change WHAT is computed (same algorithm, same refinement); results of different routes agree to
 * rounding (tests/test_gpu_weak_damping.py, tests/test_gpu_fuzz.py).
 */
#define MMX_ROUTE_AUTO 0
#define MMX_ROUTE_FUSED 1
#define MMX_ROUTE_WIDE 2
#define MMX_ROUTE_EXPLICIT_JACOBIAN 3
typedef struct mmx_tuning {
  int32_t route; /* MMX_ROUTE_* */
  int32_t max_refinement_steps; /* iterative-refinement steps a solve may take per iteration on top of the fp32 Cholesky
                                   solve (each measures its residual through J itself): 0 = the default, up to three (a
                                   further one only while the last correction exceeded 1e-3 of the step); 1..3 = at most
                                   that many; -1 = none (a measurement switch: north_star's 1e-5 needs the refinement) */
  float mixed_tolerance; /* (ABI 11) MMX_PRECISION_MIXED: the conjugate gradients of an iteration stop when the predicted size of
                            the next correction falls below this fraction of the step (0 = the default, 3e-9: north_star's 1e-5
                            then holds on every element whose double run amplifies a 1e-12 perturbation by less than 1e5; 1e-7
                            costs 15 % less and holds it where the amplification stays below 1e2) */
  int32_t mixed_max_cg; /* ... and after at most this many operator applications (0 = the default, 12) */
  int32_t reserved[4]; /* must be zero */
} mmx_tuning;
int32_t mmx_problem_set_tuning(mmx_problem* problem, const mmx_tuning* tuning);
/* MMX_ROUTE_* the last mmx_solve / mmx_solve_with_history on this handle took (MMX_ROUTE_AUTO before the first). */
int32_t mmx_problem_last_route(const mmx_problem* problem);

/* M = 3*Kp + 9*Ko + rows of the further blocks and parameter-space blocks
 * (JointErrorFunctionT::getJacobianSize, joint_error_function-inl.h:300-302). */
int32_t mmx_problem_num_rows(const mmx_problem* problem);
int32_t mmx_problem_batch(const mmx_problem* problem);

/*
 * Replaces SolverT::setEnabledParameters -> SkeletonSolverFunctionT::
 * setEnabledParameters (solver.cpp:40-47, skeleton_solver_function.cpp:45-61):
 * enabled[P] (HOST, 0/1) is the ParameterSet; derives activeJointParams
 * (parameter_transform.cpp:97-107) and the compacted enabled list
 * (gauss_newton_solver.cpp:57-66).  Default after create: all enabled.
 */
int32_t mmx_problem_set_enabled(mmx_problem* problem, const uint8_t* enabled);

/* Replaces PositionErrorFunctionT::setConstraints / OrientationErrorFunctionT::
 * setConstraints + setWeight for every batch element. */
int32_t mmx_problem_set_constraints(
    mmx_problem* problem,
    const mmx_constraint_data* data,
    void* stream);
/* The same with sizeof(mmx_constraint_data) AS THE CALLER WAS COMPILED: mmx_constraint_data grows at its end (ABI 8
 * appended function_weights / num_function_weights), and a caller built against an older header hands over a shorter
 * struct.  Fields past data_size read as zero / NULL (= absent); a data_size larger than this library's struct is
 * MMX_ERR_INVALID_ARGUMENT (the caller is newer than the library).  Bindings should call this form
 * (the C++ shell and the ctypes plumbing do); mmx_problem_set_constraints trusts the full current layout. */
int32_t mmx_problem_set_constraints_sized(
    mmx_problem* problem,
    const mmx_constraint_data* data,
    size_t data_size,
    void* stream);

/*
 * The graded kernel and the parity hook.  Replaces, per batch element,
 * SkeletonSolverFunctionT::initializeJacobianComputation + computeJacobianBlock
 * for both blocks (skeleton_solver_function.cpp:200-261 ->
 * joint_error_function-inl.h:179-297) into a pre-zeroed dense Jacobian
 * (gauss_newton_solver.cpp:166).  theta_dev [B][P], jac_dev [B][M*P] in `layout`,
 * res_dev [B][M], err_dev [B] (double, may be NULL) are DEVICE pointers.
 * Every element of jac_dev is written (zeros included).
 */
int32_t mmx_eval_jacobian(
    mmx_problem* problem,
    const float* theta_dev,
    float* jac_dev,
    float* res_dev,
    double* err_dev,
    int32_t layout,
    void* stream);

/*
 * Profiling aid: mmx_eval_jacobian with a pair of HIP events attached to the dispatch packet of the
 * J-assembly kernel itself (hipExtLaunchKernelGGL): *kernel_ms receives the duration of that kernel
 * on `stream`, without launch latency or the gap to neighbouring dispatches -- the quantity a
 * rocprofv3 kernel trace reports.  Synchronises the stream.  bench.py's roofline uses it.
 */
int32_t mmx_eval_jacobian_timed(
    mmx_problem* problem,
    const float* theta_dev,
    float* jac_dev,
    float* res_dev,
    double* err_dev,
    int32_t layout,
    void* stream,
    float* kernel_ms);

/*
 * Profiling aid: the store pattern of the J-assembly kernel on its own -- every element of
 * jac_dev [B][M*P] is written with the same stores, workgroup shape and column-major layout, but no
 * kinematics.  *kernel_ms = duration of that kernel (events on its dispatch packet).  The bandwidth
 * it reaches is what the write pattern allows on the box at hand; bench.py reports it next to the
 * J-assembly figure (roofline.store_pattern_gbs).  Problems with position / orientation rows only.
 */
int32_t mmx_debug_store_pattern(mmx_problem* problem, float* jac_dev, void* stream, float* kernel_ms);

/*
 * Forward pass only.  Replaces SkeletonStateT<T>(params, skeleton)
 * (skeleton_state.cpp:22-28,87-121).  state_dev [B][J][8] =
 * (tx,ty,tz, qx,qy,qz,qw, s) world transforms (the layout of
 * pymomentum's skel_state tensors).
 */
int32_t mmx_eval_skeleton_state(
    mmx_problem* problem,
    const float* theta_dev,
    float* state_dev,
    void* stream);

/*
 * Normal equations of one GN step without solving them: H = J[:,E]^T J[:,E]
 * (full symmetric n x n, n = |E| enabled parameters, row-major == col-major),
 * g = J[:,E]^T r.  Replaces GaussNewtonSolverT::computeJtJFromJacobianBlocks
 * (gauss_newton_solver.cpp:110-221).  jtj_dev [B][n*n], jtr_dev [B][n].
 */
int32_t mmx_eval_normal_equations(
    mmx_problem* problem,
    const float* theta_dev,
    float* jtj_dev,
    float* jtr_dev,
    double* err_dev,
    void* stream);

/*
 * Replaces, for every batch element, SolverT::solve with a GaussNewtonSolverT
 * (solver.cpp:50-128, gauss_newton_solver.cpp:224-313) exactly as the batched
 * driver does per task (tensor_ik.cpp:127-177), including its NaN/Inf revert.
 * theta_dev [B][P] in/out (DEVICE).  Optional DEVICE outputs (NULL to skip):
 *   final_error[B]   double  the value solve() returns (error at the theta
 *                            before the last step, solver.cpp:126-127)
 *   iterations[B]    int32   errorHistory_.size()
 *   status[B]        int32   MMX_SOLVE_* (a bit set; MMX_SOLVE_FAILED(status[b]) tests the error bits)
 *   error_history    double [B][max_iterations] (unused tail = 0)
 */
int32_t mmx_solve(
    mmx_problem* problem,
    const mmx_gn_options* options,
    float* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    void* stream);

/*
 * The double instantiation: SolverT<double>::solve with GaussNewtonSolverT<double> for every element
 * (momentum/solver/gauss_newton_solver.cpp:315-316; solveTensorIKProblem is templated on T,
 * pymomentum/tensor_ik/tensor_ik.cpp:95-103).  theta_dev is DOUBLE [B][P]; the constraint payload and the joint
 * constants stay float (Joint is float in the reference even for double solves, skeleton_state.cpp:89) and are
 * widened per use; every other argument as in mmx_solve.  Normal equations from the explicit Jacobian, LL^T and
 * two substitutions in double: agrees with the reference's double solver to rounding (tests: 1e-10 against the
 * oracle), no refinement step.  Built for exactness, not for the roofline: position / orientation
 * constraints and the further joint error functions (mmx_joint_constraint_block) with their losses, parameter limits
 * incl. ellipsoid limits, the model-parameter prior, per-element error-function weights, per-instance characters and
 * parents, fixed lambda, the LM schedule or the trust region (MMX_STEP_TRUST_REGION), both line-search rules.
 */
int32_t mmx_solve_f64(
    mmx_problem* problem,
    const mmx_gn_options* options,
    double* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    void* stream);

/*
 * mmx_solve with SolverT::setStoreHistory(true) (momentum/solver/solver.cpp:53-72,101-110): additionally
 *   parameter_history  float [B][max_iterations][P]  iterationHistory_["parameters"]: the parameters after
 *                      iteration i; rows from an element's iteration count on stay zero (setZero(), :70)
 * (error_history is iterationHistory_["error"], `iterations` iterationHistory_["iterations"]).  The "jtj"
 * history of GaussNewtonSolverT (gauss_newton_solver.cpp:265-278) is not stored -- n^2 floats per element and
 * iteration --; it is J^T J at the parameters BEFORE iteration i, which mmx_eval_normal_equations returns for
 * row i-1 of parameter_history (theta_init for i = 0); the host mirror rebuilds it that way on request
 * (momentum_amd/capi.py Problem.jtj_history: lower triangle + regularization on the diagonal, like :249 leaves
 * hessianApprox_).
 */
int32_t mmx_solve_with_history(
    mmx_problem* problem,
    const mmx_gn_options* options,
    float* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    float* parameter_history,
    void* stream);

/*
 * mmx_solve_with_history + the LM schedule's decisions (MMX_STEP_LM_SCHEDULE; MMX_ERR_INVALID_ARGUMENT for the other step
 * rules when step_history is not null):
 *   step_history  double [B][max_iterations][2]  (lambda iteration i factored with, its gain ratio rho = actual / predicted
 *                 decrease -- the quantity TrustRegionQRT compares with 0.25 / 0.75 / nu,
 *                 momentum/character_solver/trust_region_qr.cpp:244-268; -1 when no step could be computed); rows from an
 *                 element's iteration count on stay zero.
 * A step is accepted iff rho > 0; lambda is multiplied by lm_up when !(rho >= 0.25), by lm_down when rho > 0.75.  With this
 * output a caller (and tests/test_gpu_baseline_parity.py) can tell an element whose answer differs from another precision's
 * because a gain ratio sat on a threshold from one that differs for any other reason.  Every route; parameter_history may be
 * null.
 */
int32_t mmx_solve_with_step_history(
    mmx_problem* problem,
    const mmx_gn_options* options,
    float* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    float* parameter_history,
    double* step_history,
    void* stream);

/*
 * Numerical diagnostics of the LAST single-precision solve on this handle (one-launch and wide routes;
 * MMX_ERR_UNSUPPORTED after a solve on the explicit-Jacobian route or before the first solve): diag_dev [B][4] floats
 *   [0] the precision estimate MMX_SOLVE_PRECISION_SUSPECT is decided on (relative to |theta|)
 *   [1] smallest Cholesky pivot ratio d_jj / (H_jj + lambda) met in any iteration (~ 1 / cond(J^T J + lambda I))
 *   [2] largest |last refinement correction| / |step| of any iteration
 *   [3] |theta| of the result
 * Elements that MMX_PRECISION_AUTO re-solved in double keep the single-precision run's figures.
 */
int32_t mmx_problem_solve_diagnostics(mmx_problem* problem, float* diag_dev, void* stream);

/*
 * Parity hook of the fused solve kernel: its normal equations at theta (first iteration), over
 * the kernel's SOLVE list = enabled parameters whose Jacobian column is not structurally zero
 * (the others get an exact zero step, like in the reference where H row/col and g vanish).
 * jtj_dev [B][n*n] (J^T J without lambda), jtr_dev [B][n]; solve_list_host [<= P] receives the
 * parameter index of each compacted column, *num_solved = n.  Pass null device pointers to query
 * n / the list only.  theta is not modified.
 */
int32_t mmx_debug_fused_normal_equations(
    mmx_problem* problem,
    const float* theta_dev,
    float* jtj_dev,
    float* jtr_dev,
    int32_t* solve_list_host,
    int32_t* num_solved,
    void* stream);

/*
 * Parity hook of the wide route's first stage: H = J^T J (LOWER triangle of the n x n system written, the rest zero)
 * and g = J^T r from the tree moments (treeNormalEquationsKernel), in the layout of mmx_eval_normal_equations but with
 * the columns in the solvers' ELIMINATION order (the solve list mmx_debug_fused_normal_equations reports: the enabled
 * parameters, every subtree's ahead of those of the joints above it), so that the two can be compared entry by entry
 * after that permutation.  Problems without structurally zero columns, inside the tree kernels' scope;
 * MMX_ERR_UNSUPPORTED otherwise.
 */
int32_t mmx_debug_tree_normal_equations(mmx_problem* problem, const float* theta_dev, float* jtj_dev, float* jtr_dev, void* stream);

/* Host-buffer convenience wrappers (the reference's boundary hands over host
 * memory): copy in, run on the handle's stream, copy out, synchronise. */
int32_t mmx_solve_host(
    mmx_problem* problem,
    const mmx_gn_options* options,
    float* theta_host,
    double* final_error_host,
    int32_t* iterations_host,
    int32_t* status_host);
int32_t mmx_solve_f64_host(
    mmx_problem* problem,
    const mmx_gn_options* options,
    double* theta_host,
    double* final_error_host,
    int32_t* iterations_host,
    int32_t* status_host);
int32_t mmx_eval_jacobian_host(
    mmx_problem* problem,
    const float* theta_host,
    float* jac_host,
    float* res_host,
    double* err_host,
    int32_t layout);
/* mmx_eval_skeleton_state with host buffers: theta_host [B][P] -> state_host [B][J][8] (tx,ty,tz, qx,qy,qz,qw, s). */
int32_t mmx_eval_skeleton_state_host(mmx_problem* problem, const float* theta_host, float* state_host);

/*
 * Multi-GPU: instances are independent, so a batch shards into contiguous blocks, one problem handle per
 * GPU, with NO collective on the data path (the reference runs one independent task per element,
 * pymomentum/tensor_ik/tensor_ik.cpp:127-177).  The one exchange is the per-batch residual norms -- (sum of
 * final errors, sum of iterations, number of failed instances), three doubles -- all-reduced once per solve
 * with RCCL over xGMI.  The communicator below is that all-reduce and nothing else; RCCL is called directly
 * (librccl is looked up at run time, so a single-GPU process never loads it).
 *   multi-process (one rank per GPU):  rank 0 calls mmx_comm_unique_id and hands the 128 bytes to the other
 *     ranks (any side channel: a torch.distributed / MPI broadcast, a file); every rank calls mmx_comm_create.
 *   single process (one host thread per GPU):  mmx_comm_create_all fills one communicator per device; the
 *     threads then call mmx_comm_all_reduce_norms concurrently (the C++ shell's BatchedMultiGpuSolver).
 */
#define MMX_COMM_ID_BYTES 128 /* NCCL_UNIQUE_ID_BYTES */
typedef struct mmx_comm mmx_comm;
int32_t mmx_comm_unique_id(uint8_t id[MMX_COMM_ID_BYTES]);
int32_t mmx_comm_create(const uint8_t id[MMX_COMM_ID_BYTES], int32_t world_size, int32_t rank, int32_t device, mmx_comm** out);
int32_t mmx_comm_create_all(int32_t num_devices, const int32_t* devices, mmx_comm** out /* [num_devices] */);
int32_t mmx_comm_world_size(const mmx_comm* comm);
int32_t mmx_comm_rank(const mmx_comm* comm);
/* In-place sum over the ranks of norms_dev[3] (DEVICE doubles) on `stream`; asynchronous like any kernel. */
int32_t mmx_comm_all_reduce_norms(mmx_comm* comm, double* norms_dev, void* stream);
/* The same for three HOST doubles: staged through a device buffer and a stream the communicator owns;
 * returns when norms_host holds the sums.  (Callers without device code, e.g. the C++ shell's threads.) */
int32_t mmx_comm_all_reduce_norms_host(mmx_comm* comm, double norms_host[3]);
/* (sum final_error, sum iterations, #elements with MMX_SOLVE_FAILED(status)) of a solve's outputs -> norms_dev[3], on `stream`. */
int32_t mmx_residual_norms(int32_t batch, const double* final_error, const int32_t* iterations, const int32_t* status, double* norms_dev, void* stream);
void mmx_comm_destroy(mmx_comm* comm);

/*
 * Host-side integer bookkeeping, exposed so that it can be checked bit-exactly
 * without a GPU (north_star: "bit-exact on joint-index bookkeeping").
 * All outputs are HOST arrays owned by the caller.
 *   level[J]            depth of each joint (root = 0)
 *   tin[J], tout[J]     DFS pre-order interval: a is an ancestor-or-self of j
 *                       iff tin[a] <= tin[j] < tout[a]
 *   active_joint_params[7J]  ParameterTransformT::computeActiveJointParams(enabled)
 *   enabled_list[P], *num_enabled  GaussNewtonSolverT::updateEnabledParameters
 */
int32_t mmx_host_tables(
    const mmx_rig_desc* desc,
    const uint8_t* enabled,
    int32_t* level,
    int32_t* tin,
    int32_t* tout,
    uint8_t* active_joint_params,
    int32_t* enabled_list,
    int32_t* num_enabled);

/*
 * The solvers' column order and the tile structure of their factor (host-side integer bookkeeping, no GPU needed).
 * The reference factors a dense J (QR); the step it computes does not depend on a column order, so the solvers here
 * number the columns of (J^T J + lambda I) in an ELIMINATION order: the enabled parameters sorted by the post-order
 * position (children before their parent, smaller subtrees first) of the last joint each drives.  Two columns of J
 * overlap only when a joint of the one is an ancestor-or-self of a joint of the other, so in that order the Cholesky
 * factor has no fill outside that pattern, and the blocked solvers of the wide route skip its all-zero 16 x 16 tiles.
 *   mmx_host_elimination_order: order[P] receives the enabled parameters in that order, *num_enabled their count.
 *   mmx_host_tile_structure: symbolic factorisation on the tile grid for an n x n pattern (related[n*n], row-major,
 *     entry (row, col), row > col, non-zero = H(row, col) can be non-zero; n <= 512): row_mask[32] (bit J of word I:
 *     tile (I, J <= I) of the factor is structurally non-zero), col_mask[32] (bit I of word k: tile (I >= k, k) is),
 *     *products = tile products L(I,j) L(k,j)^T the masked factorisation performs.
 *   mmx_problem_tile_structure: the masks the problem's wide route runs with (all ones when it keeps the dense
 *     structure: further joint error functions / ellipsoid limits present, or outside the tree kernels' scope).
 */
int32_t mmx_host_elimination_order(const mmx_rig_desc* desc, const uint8_t* enabled, int32_t* order, int32_t* num_enabled);
int32_t mmx_host_tile_structure(int32_t n, const uint8_t* related, uint32_t* row_mask, uint32_t* col_mask, int64_t* products);
int32_t mmx_problem_tile_structure(mmx_problem* problem, uint32_t* row_mask, uint32_t* col_mask, int32_t* num_blocks, int32_t* num_tiles, int64_t* products);
/*
 *   mmx_host_tile_level_schedule: the order the resident factor kernel takes the block columns of that structure in --
 *     steps of mutually independent columns (no tile in each other's rows: different subtrees of the elimination tree),
 *     each column on its own waves of the 4-wave workgroup.  steps[0] = number of steps S, then 4 words per step:
 *     k | first_wave << 8 | num_waves << 12 (num_waves = 15: the whole workgroup, a panel beyond 208 rows) or -1.
 *     steps must hold 1 + 4 * 32 words.  Host-only bookkeeping (additive in ABI 9).
 */
int32_t mmx_host_tile_level_schedule(int32_t n, const uint8_t* related, int32_t* steps);
/*
 *   mmx_host_f64_assembly_list: what mmx_solve_f64 builds once per problem for its resident form -- per chunk of
 *     units_per_chunk constraint vectors (num_pos points, then three per orientation constraint) the entries (column c of
 *     solve_list, unit) of J that have an applicable source, with those sources' indices in the kernel's packed table
 *     (prefix sums of the columns' source counts in solve_list order + position in the column).  groups: two words per
 *     entry, c | unit-in-chunk << 12 | count << 18 and (count == 1 ? the source index : offset into extra);
 *     chunk_start: [chunks + 1] first group of a chunk, then [chunks] the chunk's mask of 16-column blocks with an entry.
 *     In: *num_groups / *num_extra = capacities of groups (in entries) / extra; out: the sizes (arrays are filled when
 *     they fit; call with null arrays to query).  chunk_start must hold 2 * chunks + 1 words.  Host-only (additive in ABI 9).
 */
int32_t mmx_host_f64_assembly_list(
    const mmx_rig_desc* desc,
    const int32_t* solve_list,
    int32_t n,
    const int32_t* pos_parent,
    int32_t num_pos,
    const int32_t* ori_parent,
    int32_t num_ori,
    int32_t units_per_chunk,
    uint32_t* groups,
    int32_t* num_groups,
    int32_t* extra,
    int32_t* num_extra,
    int32_t* chunk_start,
    int32_t* num_chunks);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* MMX_H_ */
