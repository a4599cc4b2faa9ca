"""LinDS task dicts -> device tables (the host half of `LinearDSEnv.set_task`).

Reference: xenoverse/linds/linds_env.py set_task :40-65 and build_dynamics_matrices :67-76 (zero-order-hold
discretisation with scipy.linalg.expm, once per task, fp64).  Task dict schema: SURVEY.md §8(a) L1.
The device computes in fp32; every table below is the fp64 host value rounded once to fp32, stored
transposed (k-major) because that is the order both the scalar-broadcast kernel and the MFMA kernel read.
"""
import numpy as np
from scipy.linalg import expm

KMAX = 6          # RandomFourier: first term (order 0) + at most max_item=5 further terms
NS_MAX, NA_MAX, NO_MAX = 32, 16, 32


def build_dynamics_matrices(ld_A, ld_B, ld_X, dt):
    """Phi = e^{A dt}, Gamma = (int_0^dt e^{A tau} d tau) B, Xt = X dt   (linds_env.py:67-76)."""
    ns = ld_A.shape[0]
    M = np.block([[ld_A, np.eye(ns)], [np.zeros((ns, 2 * ns))]])
    E = expm(M * dt)
    return E[:ns, :ns], E[:ns, ns:] @ ld_B, ld_X * dt


def fourier_terms(command):
    """(orders f64[n], coeffs f64[n, no, 2], period) of a RandomFourier-like object (utils/random_nn.py:346-368:
    attributes `coeffs` = list of (order, ndarray[no,2]) and `max_steps`)."""
    orders = np.array([float(o) for o, _ in command.coeffs], np.float64)
    coeffs = np.stack([np.asarray(c, np.float64) for _, c in command.coeffs])
    return orders, coeffs, float(command.max_steps)


def build_tables(tasks, dt=0.1, pad_observation_dim=16, pad_action_dim=8, pad_command_dim=16):
    if isinstance(tasks, dict):
        tasks = [tasks]
    if pad_command_dim != pad_observation_dim:
        raise ValueError("pad_command_dim must equal pad_observation_dim on the device path")
    n_task = len(tasks)
    NS = max(int(t["state_dim"]) for t in tasks)
    NA, NO = int(pad_action_dim), int(pad_observation_dim)
    for t in tasks:
        # same asserts as set_task (:51-54)
        assert t["observation_dim"] <= NO, \
            f"Padded observation dimension {NO} is smaller than actual observation dimension {t['observation_dim']}"
        assert t["action_dim"] <= NA, \
            f"Padded action dimension {NA} is smaller than actual action dimension {t['action_dim']}"
    if NS > NS_MAX or NA > NA_MAX or NO > NO_MAX:
        raise ValueError("unsupported dims: state %d (<=%d), action %d (<=%d), observation %d (<=%d)"
                         % (NS, NS_MAX, NA, NA_MAX, NO, NO_MAX))
    NI = max(len(t["initial_states"]) for t in tasks)
    phiT = np.zeros((n_task, NS, NS), np.float32)
    gamT = np.zeros((n_task, NA, NS), np.float32)
    cT = np.zeros((n_task, NS, NO), np.float32)
    xt = np.zeros((n_task, NS), np.float32)
    y0 = np.zeros((n_task, NO), np.float32)
    valid = np.zeros((n_task, NO), np.float32)
    cmd0 = np.zeros((n_task, NO), np.float32)
    four_coef = np.zeros((n_task, KMAX, NO, 2), np.float32)
    four_omega = np.zeros((n_task, KMAX), np.float64)
    four_period = np.ones(n_task, np.float64)
    scal = np.zeros((n_task, 8), np.float32)
    ints = np.zeros((n_task, 4), np.int32)
    init = np.zeros((n_task, NI, NS), np.float32)
    for i, t in enumerate(tasks):
        ns, na, no = int(t["state_dim"]), int(t["action_dim"]), int(t["observation_dim"])
        phi, gam, xdt = build_dynamics_matrices(np.asarray(t["ld_A"], np.float64), np.asarray(t["ld_B"], np.float64),
                                                np.asarray(t["ld_X"], np.float64), dt)
        phiT[i, :ns, :ns] = phi.T
        gamT[i, :na, :ns] = gam.T
        cT[i, :ns, :no] = np.asarray(t["ld_C"], np.float64).T
        xt[i, :ns] = xdt
        y0[i, :no] = np.asarray(t["ld_Y"], np.float64)
        valid[i, :no] = np.asarray(t["target_valid"], np.float64)
        ttype = str(t["target_type"])
        n_terms = 0
        if ttype == "static_target":
            cmd0[i, :no] = np.asarray(t["command"], np.float64)
        elif ttype == "dynamic_target":
            orders, coeffs, period = fourier_terms(t["command"])
            n_terms = len(orders)
            if n_terms > KMAX:
                raise ValueError("command has %d Fourier terms (max %d)" % (n_terms, KMAX))
            four_omega[i, :n_terms] = orders
            four_coef[i, :n_terms, :no, :] = coeffs
            four_period[i] = period
        else:
            raise Exception("Unknown target type: {}".format(ttype))
        scal[i, :6] = [t["action_cost"], t["reward_base"], t["terminate_punish"], t["reward_factor"],
                       float(t["noise_drift"]) * dt, dt]
        ints[i] = [int(t["max_steps"]), int(t["target_delay"]), len(t["initial_states"]), n_terms]
        for k, x0 in enumerate(t["initial_states"]):
            init[i, k, :ns] = np.asarray(x0, np.float64)
    return dict(NS=NS, NA=NA, NO=NO, NI=NI, dt=float(dt), phiT=phiT, gamT=gamT, cT=cT, xt=xt, y0=y0, valid=valid,
                cmd0=cmd0, four_coef=four_coef, four_omega=four_omega, four_period=four_period, scal=scal,
                ints=ints, init=init)
