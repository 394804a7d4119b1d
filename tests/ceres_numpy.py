"""Independent numpy restatement of the Ceres 1.14 solve that MSF_LOAM's scan matchers run
(mapping_scan_matcher.cc:77-97,250-259; odometry_scan_matcher.cc:67-74,269-274):

    ceres::Problem with one 7-double parameter block (PoseLocalParameterization), one
    LidarEdgeFactorSE3 / LidarPlaneFactorSE3 residual block per correspondence, ceres::HuberLoss(0.1)
    shared by all blocks, Solver::Options{max_num_iterations = 6}, everything else default.

TEST INFRASTRUCTURE.  Written from the published behaviour of Ceres Solver 1.14 (docs: "Solving
Non-linear Least Squares" — TrustRegionMinimizer, LEVENBERG_MARQUARDT strategy, Solver::Options
defaults, LossFunction / Corrector; source files named per function below) WITHOUT consulting
oracle/msfl_oracle.c, and deliberately built differently from it: the stacked Jacobian J (m x 6)
and residual vector are formed explicitly per evaluation, the damped step is the least-squares
solution of the augmented system [J; D] s = [-r; 0] (numpy lstsq, SVD based) instead of a 6 x 6
Cholesky on accumulated normal equations, rotations go through rotation matrices.  Agreement of
the two trust-region TRAJECTORIES (candidate cost, radius, step quality, accept / reject, stop
reason per iteration) is what tests/test_oracle_lm_trajectory.py asserts.
"""
import numpy as np

DBL_MIN = np.finfo(np.float64).tiny


# ---- Solver::Options defaults that matter (ceres/solver.h, 1.14) -------------------------------
class Options:
    max_num_iterations = 6                 # the reference overrides the default of 50
    initial_trust_region_radius = 1e4
    max_trust_region_radius = 1e16
    min_trust_region_radius = 1e-32
    min_relative_decrease = 1e-3
    min_lm_diagonal = 1e-6
    max_lm_diagonal = 1e32
    max_num_consecutive_invalid_steps = 5
    function_tolerance = 1e-6
    gradient_tolerance = 1e-10
    parameter_tolerance = 1e-8
    jacobi_scaling = True
    huber_delta = 0.1


def quat_to_R(q):
    """Eigen::Quaterniond(x, y, z, w).toRotationMatrix()."""
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def plus(x, delta):
    """PoseLocalParameterization::Plus (imu_fusion/pose_local_parameterization.cc:6-21) with
    Utility::deltaQ (utility.h:7-31): t += d[0:3]; q = (q * dq(d[3:6])).normalized()."""
    t = x[:3] + delta[:3]
    v = np.asarray(delta[3:6], dtype=np.float64)
    theta = np.sqrt(v @ v)
    if theta < 1e-6:
        k = 0.5 - theta ** 2 / 48.0 + theta ** 4 / 3840.0
    else:
        k = np.sin(0.5 * theta) / theta
    dq = np.array([k * v[0], k * v[1], k * v[2], np.cos(0.5 * theta)])
    ax, ay, az, aw = x[3:7]
    bx, by, bz, bw = dq
    q = np.array([aw * bx + ax * bw + ay * bz - az * by,
                  aw * by + ay * bw + az * bx - ax * bz,
                  aw * bz + az * bw + ax * by - ay * bx,
                  aw * bw - ax * bx - ay * by - az * bz])
    return np.concatenate([t, q / np.sqrt(q @ q)])


def huber(s, a):
    """ceres::HuberLoss::Evaluate (loss_function.cc): rho(s) and rho'(s) for s = |r|^2."""
    b = a * a
    if s > b:
        r = np.sqrt(s)
        return 2 * a * r - b, max(DBL_MIN, a / r), -max(DBL_MIN, a / r) / (2 * s)
    return s, 1.0, 0.0


def evaluate(corr, x, opt, want_jacobian=True):
    """ResidualBlock::Evaluate for every block + Corrector (corrector.cc): cost = 1/2 sum rho(|r|^2); with
    rho'' <= 0 (Huber) residuals and Jacobians are scaled by sqrt(rho').  Jacobians are taken in the
    tangent space: the 3x7 / 1x7 analytic blocks of lidar_factor.cc:7-44 times the 7x6 'identity on top'
    ComputeJacobian of PoseLocalParameterization (pose_local_parameterization.cc:23-27)."""
    R = quat_to_R(x[3:7])
    t = x[:3]
    res, jac, cost = [], [], 0.0
    for c in corr:
        kind = int(c["kind"])
        if kind == 0:
            continue
        p, C, N = np.asarray(c["p"]), np.asarray(c["C"]), np.asarray(c["N"])
        w = R @ p + t
        if kind == 1:                                   # LidarEdgeFactorSE3: r = N x (R p + t - C)
            r = np.cross(N, w - C)
            J = np.hstack([skew(N), -skew(N) @ R @ skew(p)])
        else:                                           # LidarPlaneFactorSE3: r = N . (R p + t - C)
            r = np.array([N @ (w - C)])
            J = np.hstack([N.reshape(1, 3), -(N.reshape(1, 3) @ R @ skew(p))])
        rho0, rho1, rho2 = huber(float(r @ r), opt.huber_delta)
        cost += 0.5 * rho0
        s = np.sqrt(rho1)                               # Corrector: rho'' <= 0  ->  alpha = 0, plain sqrt(rho') scaling
        res.append(s * r)
        jac.append(s * J)
    if not res:
        return 0.0, np.zeros(0), np.zeros((0, 6))
    return cost, np.concatenate(res), (np.vstack(jac) if want_jacobian else None)


def gradient_max_norm(x, g, plus_fn=None):
    """TrustRegionMinimizer::ComputeGradientMaxNorm... (trust_region_minimizer.cc): for an unconstrained problem
    the projected-gradient form |x - Plus(x, -g)|_inf."""
    return np.max(np.abs(x - (plus_fn or plus)(x, -g)))


class Trace:
    def __init__(self):
        self.cost, self.radius, self.rel_decrease, self.step_norm, self.accepted = [], [], [], [], []
        self.iterations = 0
        self.successful_steps = 0
        self.termination = None
        self.initial_cost = self.final_cost = 0.0


def solve(corr, x0, opt=Options, evaluate_fn=None, plus_fn=None, n_tangent=6):
    """TrustRegionMinimizer::Minimize (trust_region_minimizer.cc) with LevenbergMarquardtStrategy
    (levenberg_marquardt_strategy.cc) and TrustRegionStepEvaluator (monotonic steps).

    `evaluate_fn(corr, x, opt) -> (cost, residuals, jacobian)`, `plus_fn(x, delta)` and `n_tangent` replace the scan
    matcher's problem by any other least-squares problem: the loop itself is then checked against a run Ceres
    publishes (tests/test_oracle_lm_trajectory.py, Powell's function from the Ceres tutorial)."""
    evaluate = evaluate_fn or globals()["evaluate"]
    plus = plus_fn or globals()["plus"]
    tr = Trace()
    x = np.array(x0, dtype=np.float64)
    if evaluate_fn is None and not any(int(c["kind"]) != 0 for c in corr):
        tr.termination = "empty"                        # Problem without residual blocks: parameters untouched
        return x, tr
    # ---- Init(): iteration 0 ----
    cost, r, J = evaluate(corr, x, opt)
    tr.initial_cost = tr.final_cost = cost
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0))) if opt.jacobi_scaling else np.ones(n_tangent)     # fixed after iteration 0
    g = J.T @ r                                          # gradient in the unscaled tangent space
    Js = J * scale
    x_norm = np.sqrt(x @ x)
    gmax = gradient_max_norm(x, g, plus)
    if gmax <= opt.gradient_tolerance:
        tr.termination = "gradient"
        return x, tr
    radius, decrease_factor, reuse_diagonal, diagonal = opt.initial_trust_region_radius, 2.0, False, None
    invalid, step_successful = 0, True
    while True:
        # ---- FinalizeIterationAndCheckIfMinimizerCanContinue ----
        if tr.iterations >= opt.max_num_iterations:
            tr.termination = "max_iterations"; break
        if step_successful and gmax <= opt.gradient_tolerance:
            tr.termination = "gradient"; break
        if radius < opt.min_trust_region_radius:
            tr.termination = "min_radius"; break
        tr.iterations += 1
        # ---- LevenbergMarquardtStrategy::ComputeStep ----
        if not reuse_diagonal:
            diagonal = np.clip((Js * Js).sum(0), opt.min_lm_diagonal, opt.max_lm_diagonal)
        D = np.sqrt(diagonal / radius)
        reuse_diagonal = True
        A = np.vstack([Js, np.diag(D)])
        b = np.concatenate([-r, np.zeros(n_tangent)])
        step, *_ = np.linalg.lstsq(A, b, rcond=None)
        valid = bool(np.all(np.isfinite(step)))
        used_radius = radius
        model_cost_change = 0.0
        if valid:
            m = Js @ step
            model_cost_change = -float(m @ (r + 0.5 * m))
            valid = model_cost_change > 0.0
        if not valid:                                    # HandleInvalidStep / StepIsInvalid
            invalid += 1
            tr.cost.append(cost); tr.radius.append(used_radius); tr.rel_decrease.append(0.0); tr.step_norm.append(0.0); tr.accepted.append(-1)
            if invalid >= opt.max_num_consecutive_invalid_steps:
                tr.termination = "invalid_steps"; break
            radius *= 0.5
            step_successful = False
            continue
        invalid = 0
        delta = step * scale
        cand = plus(x, delta)
        cand_cost, cand_r, cand_J = evaluate(corr, cand, opt)
        step_norm = float(np.sqrt((x - cand) @ (x - cand)))
        if step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance):      # ParameterToleranceReached
            tr.cost.append(cand_cost); tr.radius.append(used_radius); tr.rel_decrease.append(0.0); tr.step_norm.append(step_norm); tr.accepted.append(0)
            tr.termination = "parameter"; break
        cost_change = cost - cand_cost
        if abs(cost_change) <= opt.function_tolerance * cost:                               # FunctionToleranceReached
            tr.cost.append(cand_cost); tr.radius.append(used_radius); tr.rel_decrease.append(0.0); tr.step_norm.append(step_norm); tr.accepted.append(0)
            tr.termination = "function"; break
        rel = cost_change / model_cost_change                                                # StepQuality, monotonic
        ok = rel > opt.min_relative_decrease
        tr.cost.append(cand_cost); tr.radius.append(used_radius); tr.rel_decrease.append(rel); tr.step_norm.append(step_norm); tr.accepted.append(int(ok))
        if ok:                                           # HandleSuccessfulStep / StepAccepted
            x, cost, r, J = cand, cand_cost, cand_r, cand_J
            Js = J * scale
            g = J.T @ r
            x_norm = np.sqrt(x @ x)
            gmax = gradient_max_norm(x, g, plus)
            radius = min(radius / max(1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) ** 3), opt.max_trust_region_radius)
            decrease_factor, reuse_diagonal, step_successful = 2.0, False, True
            tr.successful_steps += 1
            tr.final_cost = cost
        else:                                            # HandleUnsuccessfulStep / StepRejected
            radius /= decrease_factor
            decrease_factor *= 2.0
            reuse_diagonal, step_successful = True, False
    return x, tr
