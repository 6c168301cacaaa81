"""Host-side mirror of the reference's model classes, backed by the HIP kernels.

Mirrors (same class / method names and argument meaning):
  car_dynamics.py:9-258   KinematicBicyclePacejka   (nx = 6, the model on the reference's MPC path)
  dynamics.py:122-173     KinematicBicycleSimplified (nx = 4, the model BASELINE.json's metric names)

The reference builds CasADi symbolic functions; here the same call sites evaluate
the gfx950 kernels (B = 1 for the reference's single-car calls, any B for batched
use).  NumPy in / NumPy out, like the reference.
"""
import numpy as np
import torch

from . import _lib
from .solver import BatchedMPC

# car_dynamics.py:65-88
PARAM_NAMES = ["length", "axis_front", "axis_rear", "front", "rear", "width", "height", "mass",
               "inertia", "max_steer", "max_drive", "bf", "cf", "df", "br", "cr", "dr",
               "cm1", "cm2", "cr0", "cr1", "cr2"]

# main.py:82-86 / dynamics.py:5-42
DEFAULT_PARAMS = np.array([9.7e-2, 4.7e-2, 5e-2, 0.09, 0.07, 8e-2, 5.5e-2, 0.1735, 18.3e-5,
                           0.32, 1.0, 0.268, 2.165, 3.47, 0.242, 2.38, 2.84,
                           0.266, 0.1, 0.1025, 0.1629, 0.0011])


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("the MPC model kernels need a HIP device (no CPU fallback exists)")
    return torch.device("cuda", torch.cuda.current_device())


def _t(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64 if dtype == torch.float64 else np.int32),
                           dtype=dtype, device=_dev())


class _BicycleBase:
    """Shared plumbing: engines are cached per (params, Ts, centerline size, cost)."""
    MODEL_ID = None
    NX = None
    CLIP_INPUTS = 0     # whether the RHS clips d, delta to the actuator limits first

    def __init__(self):
        self.params = DEFAULT_PARAMS.copy()  # numeric stand-in for the reference's symbolic vector
        self.f = None
        self.f_d = None
        self.Ts = 0.05
        self._engines = {}

    # -- engine cache -------------------------------------------------------------------
    def _engine(self, p=None, S=100, N=1, v_ref=1.0, cost_w=None, wrap_mode=_lib.WRAP_FLOOR):
        p = self.params if p is None else np.asarray(p, dtype=np.float64).ravel()
        if p.shape[0] != 22:
            raise ValueError("vehicle parameter vector must have 22 entries (car_dynamics.py:65-88)")
        cw = (0.5, 1, 1, .5, 0.1, 0.01) if cost_w is None else tuple(float(v) for v in cost_w)
        key = (p.tobytes(), float(self.Ts), int(S), int(N), float(v_ref), cw, int(wrap_mode))
        eng = self._engines.get(key)
        if eng is None:
            cfg = _lib.default_config(self.MODEL_ID, int(N), S=int(S), Ts=float(self.Ts), veh=list(p),
                                      v_ref=float(v_ref), cost_w=list(cw), wrap_mode=int(wrap_mode),
                                      clip_inputs=int(self.CLIP_INPUTS))
            eng = BatchedMPC(cfg, _dev())
            self._engines[key] = eng
        return eng

    # -- car_dynamics.py:93-147 ---------------------------------------------------------
    def dynamics(self, Ts=0.05):
        """Returns f_d(y, u, p) -> y+ (4 RK4 steps of Ts/4) and sets self.f (continuous RHS)."""
        self.Ts = float(Ts)

        def f(y, u, p=None):
            y = np.asarray(y, dtype=np.float64).reshape(-1, self.NX)
            u = np.asarray(u, dtype=np.float64).reshape(-1, 2)
            return self._engine(p).rhs(_t(y), _t(u)).cpu().numpy().reshape(-1 if y.shape[0] == 1 else y.shape)

        def f_d(y, u, p=None):
            y = np.asarray(y, dtype=np.float64).reshape(-1, self.NX)
            u = np.asarray(u, dtype=np.float64).reshape(-1, 2)
            out = self._engine(p).rollout(_t(y), _t(u)).cpu().numpy()[:, 0, :]
            return out.reshape(-1) if y.shape[0] == 1 else out

        self.f, self.f_d = f, f_d
        return f_d

    # -- car_dynamics.py:149-157 --------------------------------------------------------
    def input_to_matrix(self, u):
        u = np.asarray(u)
        return u.reshape((2, u.shape[0] // 2), order="F")

    # -- car_dynamics.py:159-166 --------------------------------------------------------
    def simulate(self, N_sim, y_0, u, p=None):
        """mapaccum semantics: returns the (nx, N_sim) matrix of x_1..x_N_sim."""
        y0 = np.asarray(y_0, dtype=np.float64).reshape(1, self.NX)
        u = np.asarray(u, dtype=np.float64)
        U = (u.reshape(2, N_sim, order="F") if u.ndim == 1 else u.reshape(2, N_sim)).T.reshape(1, 2 * N_sim)
        X = self._engine(p).rollout(_t(y0), _t(U)).cpu().numpy()[0]  # [N_sim, nx]
        return X.T

    # -- car_dynamics.py:168-172 --------------------------------------------------------
    def wrap_to_pi(self, angle):
        return np.mod(np.asarray(angle) + np.pi, 2 * np.pi) - np.pi

    # -- car_dynamics.py:174-192 --------------------------------------------------------
    def find_nearest_point(self, size, vehicle_position, centerline):
        """centerline: (size, 2) array.  Returns (nearest, previous, next) rows."""
        cl = np.asarray(centerline, dtype=np.float64).reshape(size, 2)
        pos = np.asarray(vehicle_position, dtype=np.float64).reshape(-1)
        pose = np.array([[pos[0], pos[1], 0.0]])
        _, idx = self._engine(S=size).stage_errors(_t(pose), _t(cl.ravel(order="F")))
        i = int(idx.cpu()[0])
        return cl[i], cl[i - 1 if i > 0 else 0], cl[i + 1]

    # -- car_dynamics.py:194-228 --------------------------------------------------------
    def compute_errors(self, size, vehicle_position, vehicle_heading, centerline_flat):
        """centerline_flat: [x_0..x_{S-1}, y_0..y_{S-1}].  Returns (cte, heading_error, pos_error)."""
        cl = np.asarray(centerline_flat, dtype=np.float64).reshape(-1)
        pos = np.asarray(vehicle_position, dtype=np.float64).reshape(-1)
        pose = np.array([[pos[0], pos[1], float(np.asarray(vehicle_heading).reshape(-1)[0])]])
        err, _ = self._engine(S=size).stage_errors(_t(pose), _t(cl))
        e = err.cpu().numpy()[0]
        return e[0], e[1], e[2]

    # -- car_dynamics.py:230-258 --------------------------------------------------------
    def generate_stage_cost_fun(self, centerline_size, target_v, c=np.array([0.5, 1, 1, .5, 0.1, 0.01])):
        def L_cost(X, u, centerline):
            X = np.asarray(X, dtype=np.float64).reshape(-1, self.NX)
            u = np.asarray(u, dtype=np.float64).reshape(-1, 2)
            cl = np.asarray(centerline, dtype=np.float64).reshape(-1)
            eng = self._engine(S=centerline_size, v_ref=target_v, cost_w=c)
            out = eng.stage_cost(_t(X), _t(u), _t(cl)).cpu().numpy()
            return float(out[0]) if out.shape[0] == 1 else out
        return L_cost


class KinematicBicyclePacejka(_BicycleBase):
    """car_dynamics.py:9 -- state [x, y, phi, vx, vy, omega], input [d, delta].  The CasADi RHS of
    car_dynamics.py:93-129 uses the inputs as they come (no clipping)."""
    MODEL_ID = _lib.MODEL_PACEJKA
    NX = 6
    CLIP_INPUTS = 0


class KinematicBicycleSimplified(_BicycleBase):
    """dynamics.py:122 -- state [x, y, phi, v], input [d, delta].  dynamics.py:163 clips d and delta
    to the actuator limits inside the RHS, so this mirror's engines do too."""
    MODEL_ID = _lib.MODEL_KINEMATIC
    NX = 4
    CLIP_INPUTS = 1
