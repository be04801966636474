#!/usr/bin/env python
"""Pin the oracle to the REFERENCE: replay the committed golden fixtures through ossamaAhmed/blackbox_mpc itself.

`tests/golden/*.npz` are outputs of this repo's NumPy oracle under injected standard noise (the reference ships no
golden vectors and cannot be imported in the build container: every hot-path module imports tensorflow, which is not
installed).  Until the fixtures have been reproduced by the reference, parity is "unpinned" (DESIGN.md section 2).
This script is what closes that gap on any machine that has the reference's own environment:

    pip install tensorflow==2.0.0 tensorflow-probability==0.8.0rc0 gym      # reference setup.py:12-14
    python tools/pin_oracle.py --reference /path/to/blackbox_mpc_checkout

It imports the reference, replaces `tf.random.truncated_normal / uniform / normal` by functions that hand out the
fixtures' injected standard draws in call order (scaled and shifted exactly as TF's own would be:
`mean + stddev * z`, `minval + (maxval - minval) * u`), runs the reference's own RandomSearch / CEM / PI2 / PSO /
SPSA / CMA-ES optimizers and DeterministicTrajectoryEvaluator on the fixtures' inputs, and diffs every stored output against
`tests/golden/`.  Exit codes: 0 = every fixture reproduced within tolerance (parity PINNED: record the printed
summary in DESIGN.md), 1 = a mismatch (the oracle is wrong somewhere: the diff says where), 3 = tensorflow or the
reference is not importable here -- said LOUDLY, nothing is compared, parity stays unpinned.

Nothing in the product or the test suite imports this file; the reference's sources are never copied -- they are
imported from the checkout the caller points at.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
F = np.float32


def skip(msg):
    bar = "=" * 100
    print("%s\nPIN SKIPPED -- PARITY REMAINS UNPINNED: %s\n%s" % (bar, msg, bar), file=sys.stderr)
    sys.exit(3)


class NoiseQueue:
    """Hands the fixtures' standard draws to the reference in the order its graph asks for them."""

    def __init__(self, tf):
        self.tf = tf
        self.q = {"trunc": [], "uniform": [], "normal": []}
        self.orig = (tf.random.truncated_normal, tf.random.uniform, tf.random.normal)

    def push(self, kind, arr):
        self.q[kind].append(np.asarray(arr, F))

    def _pop(self, kind, shape):
        """A graph-time-safe draw: the reference's optimizers are @tf.function graphs whose iteration loop is a
        tf.while_loop traced ONCE, so the queue must be popped when the op RUNS (every loop iteration), not when the
        Python body is traced -- hence tf.numpy_function."""
        tf = self.tf
        static = []
        for d in (shape if isinstance(shape, (list, tuple)) else [shape]):
            v = tf.get_static_value(d) if tf.is_tensor(d) else d
            static.append(None if v is None else int(v))

        def take():
            if not self.q[kind]:
                raise RuntimeError("the reference asked for a %s draw %s that the fixture does not hold" % (kind, static))
            z = self.q[kind].pop(0)
            if None not in static and tuple(z.shape) != tuple(static):
                raise RuntimeError("%s draw: fixture holds %s, the reference asked for %s" % (kind, z.shape, static))
            return z
        z = tf.numpy_function(take, [], tf.float32)
        z.set_shape(static)
        return z

    def install(self):
        tf = self.tf

        def truncated_normal(shape, mean=0.0, stddev=1.0, dtype=tf.float32, seed=None, name=None):
            return tf.cast(mean, dtype) + tf.cast(stddev, dtype) * self._pop("trunc", shape)

        def uniform(shape, minval=0, maxval=None, dtype=tf.float32, seed=None, name=None):
            if dtype in (tf.int32, tf.int64):        # SPSA's Rademacher draw (spsa.py:73-75): fixture holds +-1
                return tf.cast((self._pop("uniform", shape) + 1.0) / 2.0, dtype)
            u = self._pop("uniform", shape)
            return tf.cast(minval, dtype) + (tf.cast(maxval, dtype) - tf.cast(minval, dtype)) * u

        def normal(shape, mean=0.0, stddev=1.0, dtype=tf.float32, seed=None, name=None):
            return tf.cast(mean, dtype) + tf.cast(stddev, dtype) * self._pop("normal", shape)

        tf.random.truncated_normal, tf.random.uniform, tf.random.normal = truncated_normal, uniform, normal

    def restore(self):
        self.tf.random.truncated_normal, self.tf.random.uniform, self.tf.random.normal = self.orig

    def assert_drained(self, what):
        left = {k: len(v) for k, v in self.q.items() if v}
        if left:
            raise RuntimeError("%s: the reference did not consume every injected draw: %s" % (what, left))


class Box:
    def __init__(self, low, high):
        self.low, self.high = np.asarray(low, F), np.asarray(high, F)
        self.shape = self.low.shape


class Report:
    def __init__(self):
        self.rows, self.failed = [], False

    def check(self, fixture, item, got, want, rtol=0.0, atol=0.0, exact=False):
        got, want = np.asarray(got), np.asarray(want)
        if got.shape != want.shape:
            self.rows.append((fixture, item, "SHAPE %s vs %s" % (got.shape, want.shape)))
            self.failed = True
            return
        if exact:
            ok = bool(np.array_equal(got, want))
            err = 0.0 if ok else float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64))))
        else:
            err = float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64)))) if got.size else 0.0
            ok = bool(np.allclose(got, want, rtol=rtol, atol=atol))
        self.rows.append((fixture, item, "%s max|diff| = %.3g" % ("ok  " if ok else "FAIL", err)))
        self.failed |= not ok


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", default=os.environ.get("BBMPC_REFERENCE", "/root/reference"),
                    help="checkout of ossamaAhmed/blackbox_mpc (v0.3)")
    args = ap.parse_args()
    try:
        import tensorflow as tf
    except Exception as ex:                                     # noqa: BLE001
        skip("tensorflow is not importable (%s: %s)" % (type(ex).__name__, ex))
    if not os.path.isdir(os.path.join(args.reference, "blackbox_mpc")):
        skip("no reference checkout at %s (pass --reference)" % args.reference)
    sys.path.insert(0, args.reference)
    try:
        from blackbox_mpc.dynamics_functions.deterministic_mlp import DeterministicMLP
        from blackbox_mpc.dynamics_handlers.system_dynamics_handler import SystemDynamicsHandler
        from blackbox_mpc.optimizers.cem import CEMOptimizer
        from blackbox_mpc.optimizers.cma_es import CMAESOptimizer
        from blackbox_mpc.optimizers.spsa import SPSAOptimizer
        from blackbox_mpc.optimizers.pi2 import PI2Optimizer
        from blackbox_mpc.optimizers.pso import PSOOptimizer
        from blackbox_mpc.optimizers.random_search import RandomSearchOptimizer
        from blackbox_mpc.trajectory_evaluators.deterministic import DeterministicTrajectoryEvaluator
        from blackbox_mpc.utils.pendulum import PendulumTrueModel, pendulum_reward_function
    except Exception as ex:                                     # noqa: BLE001
        skip("the reference is not importable (%s: %s)" % (type(ex).__name__, ex))
    sys.path.insert(0, os.path.join(args.reference, "tutorials", "mujoco"))
    from cost_func import reward_function as cheetah_reward     # tutorials/mujoco/cost_func.py:5-22

    nq = NoiseQueue(tf)
    nq.install()
    rep = Report()
    load = lambda n: dict(np.load(os.path.join(GOLDEN, n + ".npz")))
    act_space, obs_space = Box([-2.0], [2.0]), Box([-1.0, -1.0, -8.0], [1.0, 1.0, 8.0])

    def pend_evaluator():
        h = SystemDynamicsHandler(env_action_space=act_space, env_observation_space=obs_space,
                                  dynamics_function=PendulumTrueModel(), true_model=True)
        return DeterministicTrajectoryEvaluator(reward_function=pendulum_reward_function, system_dynamics_handler=h)

    def call(opt, states):
        a, n, r = opt(tf.constant(states, tf.float32), tf.constant(0, tf.int32), tf.constant(False))
        return a.numpy(), n.numpy(), r.numpy()

    RT, AT = 2e-4, 2e-3            # H-step rewards (tests/test_gpu_pendulum.py); refits 2e-5; single steps 1e-5
    # ---- cfg1: RandomSearch (random_search.py:38-48) -----------------------------------------------------------
    g = load("cfg1")
    opt = RandomSearchOptimizer(act_space, obs_space, planning_horizon=20, population_size=200, num_agents=1)
    opt.set_trajectory_evaluator(pend_evaluator())
    nq.push("uniform", g["uniform"])
    a, n, r = call(opt, g["states"])
    nq.assert_drained("cfg1")
    rep.check("cfg1", "action", a, g["action"], exact=True)
    rep.check("cfg1", "next_state", n, g["next_state"], rtol=1e-5, atol=1e-5)
    rep.check("cfg1", "reward", r, g["reward"], rtol=1e-5, atol=1e-5)
    ev = pend_evaluator()
    seq = g["uniform"] * 4.0 - 2.0
    rep.check("cfg1", "rewards[N,A]", ev(tf.constant(g["states"]), tf.constant(seq.astype(F)), tf.constant(0)).numpy(),
              g["rewards"], rtol=RT, atol=AT)
    # ---- cfg2: CEM (cem.py:74-136), 3 iterations ---------------------------------------------------------------
    g = load("cfg2")
    opt = CEMOptimizer(act_space, obs_space, planning_horizon=30, max_iterations=3, population_size=500, num_elite=50,
                       num_agents=1, alpha=0.25)
    opt.set_trajectory_evaluator(pend_evaluator())
    for it in range(3):
        nq.push("trunc", g["trunc"][it])
    a, n, r = call(opt, g["states"])
    nq.assert_drained("cfg2")
    rep.check("cfg2", "action", a, g["action"], atol=2e-5)
    rep.check("cfg2", "next_state", n, g["next_state"], rtol=1e-5, atol=1e-5)
    rep.check("cfg2", "reward", r, g["reward"], rtol=1e-5, atol=1e-5)
    # ---- cfg3: PI2 (pi2.py:58-96), two control steps with the shift-left warm start ----------------------------
    g = load("cfg3")
    opt = PI2Optimizer(act_space, obs_space, planning_horizon=30, max_iterations=2, population_size=256, num_agents=4,
                       lamda=1.0)
    opt.set_trajectory_evaluator(pend_evaluator())
    for step in range(2):
        for it in range(2):
            nq.push("trunc", g["trunc"][step][it])
        a, _, _ = call(opt, g["states"])
        nq.assert_drained("cfg3 step %d" % step)
        rep.check("cfg3", "action[step %d]" % step, a, g["action"][step], atol=5e-3)
        rep.check("cfg3", "prev_mean[step %d]" % step, opt._previous_solution.numpy(), g["prev_mean"][step], atol=5e-3)
    # ---- cfg4: evaluator on the learned MLP (deterministic.py:26-127, deterministic_mlp.py:27-51) ---------------
    g = load("cfg4")
    sys.path.insert(0, ROOT)
    from tests.golden.make_golden import cheetah_problem        # the fixture's weights / statistics recipe (data)
    ws, bs, stats, _ = cheetah_problem(int(g["mlp_seed"]))
    S, U = 20, 6
    c_act, c_obs = Box([-1.0] * U, [1.0] * U), Box([-10.0] * S, [10.0] * S)
    mlp = DeterministicMLP(layers=[S + U, 200, 200, S], activation_functions=[tf.math.tanh, tf.math.tanh, None])
    mlp(tf.zeros([1, S + U]), tf.constant(False))               # build the Dense layers
    for layer, w, b in zip(mlp.layers, ws, bs):
        layer.set_weights([w, b])
    h = SystemDynamicsHandler(env_action_space=c_act, env_observation_space=c_obs, dynamics_function=mlp,
                              true_model=False, is_normalized=True)
    (h._mean_states, h._std_states, h._mean_actions, h._std_actions, h._mean_targets, h._std_targets) = stats
    ev = DeterministicTrajectoryEvaluator(reward_function=cheetah_reward, system_dynamics_handler=h)
    rep.check("cfg4", "rewards[N,A]", ev(tf.constant(g["states"]), tf.constant(g["seq"]), tf.constant(0)).numpy(),
              g["rewards"], rtol=1e-3, atol=3e-2)
    rep.check("cfg4", "predict_next_state", ev.predict_next_state(tf.constant(g["step_states"]),
                                                                  tf.constant(g["step_actions"])).numpy(),
              g["step_next"], rtol=2e-5, atol=2e-5)
    # ---- cfg5: PSO after reset() (pso.py:70-160) -----------------------------------------------------------------
    g = load("cfg5")
    opt = PSOOptimizer(act_space, obs_space, planning_horizon=8, max_iterations=3, population_size=96, num_agents=2)
    opt.set_trajectory_evaluator(pend_evaluator())
    nq.push("uniform", g["reset_pos"])                          # :147-149 then :151-152
    nq.push("uniform", g["reset_vel"])
    opt.reset()
    nq.assert_drained("cfg5 reset")
    for it in range(3):                                         # :107-109 two scalar normals per iteration
        nq.push("normal", np.asarray(g["normal2"][it][0]).reshape(()))
        nq.push("normal", np.asarray(g["normal2"][it][1]).reshape(()))
    nq.push("trunc", g["trunc"])
    nq.push("uniform", g["uniform"])
    a, _, _ = call(opt, g["states"])
    nq.assert_drained("cfg5")
    rep.check("cfg5", "action", a, g["action"], atol=2e-5)
    rep.check("cfg5", "pos", opt._particle_positions.numpy(), g["pos"], atol=2e-5)
    rep.check("cfg5", "vel", opt._particle_velocities.numpy(), g["vel"], atol=2e-5)
    rep.check("cfg5", "gbest", opt._global_best_known_position.numpy(), g["gbest"], atol=2e-5)
    # ---- cfg6: SPSA (spsa.py:61-117), two control steps with the shift-left warm start (:114-115) --------------------
    g = load("cfg6")
    opt = SPSAOptimizer(act_space, obs_space, planning_horizon=10, max_iterations=3, population_size=64, num_agents=3)
    opt.set_trajectory_evaluator(pend_evaluator())
    for step in range(2):
        for it in range(3):                                     # :73-75 the int32 {0, 1} draw, mapped to -1 / +1 by the reference
            nq.push("uniform", g["rademacher"][step][it])
        a, _, _ = call(opt, g["states"])
        nq.assert_drained("cfg6 step %d" % step)
        rep.check("cfg6", "action[step %d]" % step, a, g["action"][step], atol=1e-4)
        rep.check("cfg6", "params[step %d]" % step, opt._current_parameters.numpy(), g["params"][step], atol=1e-4)
    # ---- cfg7: CMA-ES (cma_es.py:129-213), ONE iteration from the constructor state: B = D = I, so the samples and the
    # (m, p_sigma, sigma, p_C, C) update do not depend on tf.linalg.svd's conventions; D = sqrt(singular values) neither
    g = load("cfg7")
    opt = CMAESOptimizer(act_space, obs_space, planning_horizon=6, max_iterations=1, population_size=96, num_elite=12,
                         num_agents=2)
    opt.set_trajectory_evaluator(pend_evaluator())
    nq.push("normal", g["normal"][0])
    a, n, r = call(opt, g["states"])
    nq.assert_drained("cfg7")
    rep.check("cfg7", "action", a, g["action"], atol=2e-5)
    rep.check("cfg7", "next_state", n, g["next_state"], rtol=1e-5, atol=1e-5)
    rep.check("cfg7", "x_sorted[:k] (samples in rank order)", opt._x_sorted.numpy()[:12], g["samples"][g["order"]], atol=2e-5)
    rep.check("cfg7", "m", opt._m.numpy(), g["m"], atol=2e-5)
    rep.check("cfg7", "sigma", opt._sigma.numpy(), g["sigma"], rtol=2e-5, atol=1e-6)
    rep.check("cfg7", "p_sigma", opt._p_sigma.numpy(), g["p_sigma"], rtol=1e-4, atol=2e-5)
    rep.check("cfg7", "p_C", opt._p_C.numpy(), g["p_C"], rtol=1e-4, atol=2e-5)
    rep.check("cfg7", "C", opt._C.numpy(), g["C"], rtol=1e-4, atol=2e-5)
    rep.check("cfg7", "diag(D)", np.diag(opt._D.numpy()), g["D"], rtol=1e-4, atol=2e-5)
    nq.restore()

    w = max(len(r[1]) for r in rep.rows)
    print("reference: %s   tensorflow %s" % (args.reference, tf.__version__))
    for fx, item, res in rep.rows:
        print("  %-5s %-*s %s" % (fx, w, item, res))
    if rep.failed:
        print("RESULT: MISMATCH -- the oracle disagrees with the reference on the rows marked FAIL")
        sys.exit(1)
    print("RESULT: every golden fixture reproduced by the reference within the stated tolerances -> parity PINNED "
          "(record this output in DESIGN.md section 2)")


if __name__ == "__main__":
    main()
