"""The pymomentum.solver2-shaped surface (momentum_amd/solver2.py): object model
on CPU; on the GPU the reference's own Python test of this surface, pymomentum/test/test_solver2.py:
135-200 (test_ik_basic: targets from theta* = 0.5 * rand, GN 200 iterations, lambda = 1e-5 => joint
positions allclose(1e-4), error history decreasing and reproduced exactly by a second solve)."""
import numpy as np
import pytest

from momentum_amd import make_test_character, solver2 as pym_solver2


def test_object_model_matches_the_reference_binding():
    character = pym_solver2.Character(make_test_character(4))
    n_joints = character.skeleton.size
    assert n_joints == 4 and character.parameter_transform.size == 11
    pos_error = pym_solver2.PositionErrorFunction(character)
    pos_error.add_constraint(parent=0, offset=np.array([1.0, 0.0, 0.0]), target=np.array([10.0, 0.0, 0.0]), weight=1.0)
    assert len(pos_error.constraints) == 1
    assert np.isclose(pos_error.constraints[0].data["offset"], [1.0, 0.0, 0.0]).all()
    assert np.isclose(pos_error.constraints[0].data["target"], [10.0, 0.0, 0.0]).all()
    pos_error.clear_constraints()
    assert len(pos_error.constraints) == 0
    pos_error.add_constraints(parent=np.arange(n_joints), target=np.zeros((n_joints, 3)))
    assert len(pos_error.constraints) == n_joints
    with pytest.raises(RuntimeError):
        pos_error.add_constraint(parent=n_joints, target=np.zeros(3))  # invalid joint index
    fn = pym_solver2.SkeletonSolverFunction(character, [pos_error])
    assert len(fn.error_functions) == 1
    other = pym_solver2.Character(make_test_character(5))
    with pytest.raises(RuntimeError):
        fn.add_error_function(pym_solver2.PositionErrorFunction(other))  # different character
    opt = pym_solver2.GaussNewtonSolverOptions()
    assert (opt.min_iterations, opt.max_iterations, opt.threshold, opt.regularization, opt.do_line_search) == (1, 2, 1.0, 0.05, False)
    plane = pym_solver2.PlaneErrorFunction(character, above=True, weight=2.0)
    plane.add_constraints(normal=np.tile([0.0, 1.0, 0.0], (2, 1)), d=np.array([0.5, 1.0]), parent=[1, 3])
    blk = plane.block(3)
    assert blk.type == 1 and blk.rows == 2 and blk.plane_d.shape == (3, 2) and blk.function_weight == 2.0
    assert np.allclose(blk.plane_d[2], [0.5, 1.0])
    with pytest.raises(RuntimeError):
        pym_solver2.ModelParametersErrorFunction(character).set_target_parameters(np.zeros(5))
    solver = pym_solver2.GaussNewtonSolver(fn, opt)
    with pytest.raises(RuntimeError):
        solver.set_enabled_parameters(np.ones(3, bool))


@pytest.mark.gpu
def test_ik_basic_like_the_reference_python_test():
    character = pym_solver2.Character(make_test_character(4))
    n_joints = character.skeleton.size
    n_params = character.parameter_transform.size
    np.random.seed(42)
    model_params_init = np.zeros(n_params, dtype=np.float32)
    model_params_target = (0.5 * np.random.rand(n_params)).astype(np.float32)
    skel_state_target = pym_solver2.model_parameters_to_skeleton_state(character, model_params_target)
    assert skel_state_target.shape == (n_joints, 8)
    pos_error = pym_solver2.PositionErrorFunction(character)
    pos_error.add_constraints(parent=np.arange(n_joints), target=skel_state_target[:, :3])
    solver_function = pym_solver2.SkeletonSolverFunction(character, [pos_error])
    solver_options = pym_solver2.GaussNewtonSolverOptions()
    solver_options.max_iterations = 200
    solver_options.regularization = 1e-5
    solver = pym_solver2.GaussNewtonSolver(solver_function, solver_options)
    model_params_final = solver.solve(model_params_init)
    skel_state_final = pym_solver2.model_parameters_to_skeleton_state(character, model_params_final)
    assert np.allclose(skel_state_final[:, :3], skel_state_target[:, :3], rtol=1e-4, atol=1e-4)
    assert len(solver.per_iteration_errors) > 1
    assert solver.per_iteration_errors[-1] < solver.per_iteration_errors[0]
    prev = solver.per_iteration_errors
    solver.solve(model_params_init)
    assert solver.per_iteration_errors == prev  # deterministic
    # wrong size => exception like the binding (solver2_pybind.cpp:889-892)
    with pytest.raises(RuntimeError):
        solver.solve(np.zeros(n_params + 1, np.float32))
    # residual / Jacobian / gradient of the solver function
    r, J = solver_function.get_jacobian(model_params_init)
    assert J.shape == (3 * n_joints, n_params) and r.shape == (3 * n_joints,)
    assert np.isclose(r @ r, solver_function.get_error(model_params_init), rtol=1e-5)
    assert np.allclose(solver_function.get_gradient(model_params_init), 2 * J.T @ r)
    # SolverT::setStoreHistory / getHistory (solver.cpp:53-72; "jtj": gauss_newton_solver.cpp:262-279): the first block
    # of the jtj history is the damped lower triangle of J^T J at the initial parameters
    assert solver.get_history() == {}
    solver_options.max_iterations = 5
    solver = pym_solver2.GaussNewtonSolver(solver_function, solver_options)
    solver.set_store_history(True)
    final = solver.solve(model_params_init)
    hist = solver.get_history()
    its = int(hist["iterations"])
    assert hist["parameters"].shape == (5, n_params) and hist["jtj"].shape == (5, n_params, n_params) and 1 <= its <= 5
    assert np.array_equal(hist["parameters"][its - 1], final)
    want = np.tril(J.T @ J) + 1e-5 * np.eye(n_params)
    assert np.allclose(hist["jtj"][0], want, rtol=1e-4, atol=1e-4 * np.abs(want).max())


@pytest.mark.gpu
def test_batched_solve_with_floor_limits_and_prior(orc):
    """Batch of poses with a floor (half-plane), an aim and a fixed-axis error function, parameter limits
    and a model-parameter prior, line search on: parity with the oracle through the same description."""
    from momentum_amd import _abi, humanoid72_landmark_joints, make_humanoid72
    from momentum_amd._abi import GnOptions, ParameterLimit
    from tests.helpers import make_problem

    rig = make_humanoid72(unit=0.01)
    character = pym_solver2.Character(rig)
    lm = humanoid72_landmark_joints(rig)
    B = 4
    cons, th0, _ = make_problem(rig, lm, lm, B, seed=99, perturb=0.3)
    rng = np.random.default_rng(1)
    pos = pym_solver2.PositionErrorFunction(character, weight=0.9)
    pos.add_constraints(parent=lm, target=cons.pos_target.reshape(B, -1, 3), offset=cons.pos_offset.reshape(B, -1, 3))
    ori = pym_solver2.OrientationErrorFunction(character)
    ori.add_constraints(target=cons.ori_target.reshape(B, -1, 4), parent=lm, offset=cons.ori_offset.reshape(B, -1, 4))
    floor = pym_solver2.PlaneErrorFunction(character, above=True, weight=3.0)
    feet = [j for j in range(rig.num_joints) if rig.parent[j] >= 0][-6:]
    floor.add_constraints(normal=np.tile([0.0, 1.0, 0.0], (len(feet), 1)), d=np.full(len(feet), -0.2), parent=feet)
    aim = pym_solver2.AimDistErrorFunction(character, weight=0.2)
    aim.add_constraint(local_point=[0, 0, 0], local_dir=[0, 0, 1], global_target=rng.uniform(-1, 1, (B, 3)), parent=int(lm[3]))
    axis = pym_solver2.FixedAxisCosErrorFunction(character, alpha=0.0, c=0.5)
    axis.add_constraint(local_axis=[0, 1, 0], global_axis=[0, 1, 0], parent=int(lm[0]), weight=0.7)
    limits = [ParameterLimit.minmax(8, -0.1, 0.1, 1.0), ParameterLimit.linear(10, 11, 1.0, 0.0, weight=0.5)]
    lim = pym_solver2.LimitErrorFunction(character, limits, weight=0.8)
    prior = pym_solver2.ModelParametersErrorFunction(character, np.zeros(rig.num_params), np.full(rig.num_params, 0.3), weight=0.5)
    fn = pym_solver2.SkeletonSolverFunction(character, [pos, ori, floor, aim, axis, lim, prior])
    opts = pym_solver2.GaussNewtonSolverOptions()
    opts.min_iterations = opts.max_iterations = 8
    opts.do_line_search = True
    solver = pym_solver2.GaussNewtonSolver(fn, opts)
    theta = solver.solve(th0)
    assert theta.shape == (B, rig.num_params) and len(solver.per_iteration_errors) == B
    # the same problem described to the oracle
    blocks = [floor.block(B), aim.block(B), axis.block(B)]
    full = orc.Constraints(
        cons.pos_parent, cons.pos_offset, cons.pos_target, 0.9 * cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target, cons.ori_weight,
        limits=limits, limit_function_weight=0.8, model_target=np.zeros((B, rig.num_params), np.float32),
        model_weights=np.full((B, rig.num_params), 0.3, np.float32), model_function_weight=0.5, joint_blocks=blocks,
    )  # fmt: skip
    ref = orc.solve_batch(rig, full, th0, GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05, do_line_search=True), dtype="f64")
    rel = np.linalg.norm(theta - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    assert (rel <= 3e-5).all(), rel
    h = np.array(solver.per_iteration_errors)
    assert np.abs(h - ref["error_history"]).max() <= 1e-4 * max(1.0, np.abs(ref["error_history"]).max())
    assert (np.diff(h, axis=1) <= 1e-6 * np.abs(h[:, :-1]) + 1e-12).all()  # the line search keeps it monotone
