"""The pose loop cut into the same stages and exchange records as the HIP path (numpy, float64).

TEST INFRASTRUCTURE (see oracle/picp_oracle.py).  The sharded driver of the package
(sharded.run_sharded) only needs an object with state_init / stage1 / stage2 / solve / stopped /
finish; this numpy implementation lets the multi-rank control flow -- partitioning, the two
all-reduces per iteration, the stop protocol -- run on CPU ranks under the gloo backend, and be
compared with the monolithic restatement in picp_oracle.projective_icp.

Record layouts are the device's (csrc/picp_device.h):
  s1 = [sum chi, count, in-image count, 0]                       over chi <= kernel_threshold
  s2 = [H(21), b(6), sum chi, count, 0, 0, 0]  of ONE set: the stage-1 set with chi < mean (mode 0) or
       the whole stage-1 set (mode 1, entered when the solve step finds the mode-0 set smaller than
       min_inliers; that round does not advance the iteration)
"""
from __future__ import annotations

import numpy as np
import torch

from . import picp_oracle as po

IU = np.triu_indices(6)


class NumpyStages:
    def __init__(self, uv, pw, w_T_c0, params: po.IcpParams):
        self.uv = np.asarray(uv, dtype=np.float64).reshape(-1, 2)
        self.pw = np.asarray(pw, dtype=np.float64).reshape(-1, 3)
        self.start = np.array(w_T_c0, dtype=np.float64)
        self.p = params

    def state_init(self):                                  # src/ProjectiveICP.py:157-171
        self.w_T_c = self.start.copy()
        self.c_T_w = np.linalg.inv(self.w_T_c)
        self.thr, self.lam = self.p.kernel_threshold, self.p.damping
        self.error_prev, self.ring = np.inf, np.zeros(10)
        self.mean_slope = self.sigma_slope = 0.0
        self.stuck = self.flicker = self.it = 0
        self.mode = 0
        self.stop = self.p.num_iterations <= 0
        self.hist = {"T": [], "error": [], "num_inliers": [], "kernel_threshold": [], "dumping_factor": [],
                     "n_inside": []}

    def _terms(self):
        inside, e, J = po.errors_and_jacobians(self.uv, self.pw, self.w_T_c, self.p, self.c_T_w)
        with np.errstate(invalid="ignore"):
            chi = (e[:, None, :] @ e[:, :, None]).reshape(-1)     # as picp_oracle.linearize (:293)
            keep = inside & (chi <= self.thr)
        return inside, e, J, chi, keep

    def stage1(self):
        if self.stop:
            return torch.zeros(4, dtype=torch.float64)
        inside, e, J, chi, keep = self._terms()
        self._cache = (e, J, chi, keep)
        return torch.tensor([chi[keep].sum(), keep.sum(), inside.sum(), 0.0], dtype=torch.float64)

    def stage2(self, s1):
        out = np.zeros(32)
        if self.stop or float(s1[1]) == 0.0:
            return torch.from_numpy(out)
        e, J, chi, keep = self._cache
        mean = float(s1[0]) / float(s1[1])
        m = keep if self.mode else (keep & (chi < mean))
        out[0:21] = np.einsum("mri,mrj->ij", J[m], J[m])[IU]
        out[21:27] = np.einsum("mri,mr->i", J[m], e[m])
        out[27], out[28] = chi[m].sum(), m.sum()
        return torch.from_numpy(out)

    def solve(self, s1, s2):                               # :260-273, :200-238
        if self.stop:
            return
        s1, s2, p = s1.numpy(), s2.numpy(), self.p
        n_inside, n1 = s1[2], s1[1]
        if n_inside != 0 and n1 != 0 and not self.mode and s2[28] < p.min_inliers:
            self.mode = 1                                  # :307 failed: re-sum the stage-1 set, same pose
            return
        self.mode = 0
        if n_inside == 0:
            T1, n_inl, error, done = self.w_T_c, 0, np.inf, False
        else:
            done = True
            H, b = np.zeros((6, 6)), np.zeros(6)
            if n1 == 0:
                n_inl, error = 0, np.inf
            else:
                H[IU] = s2[0:21]
                H = H + np.triu(H, 1).T
                b = s2[21:27]
                n_inl = int(s2[28])
                error = s2[27] / n_inl if n_inl > 0 else np.inf
            dx = np.linalg.lstsq(H + np.eye(6) * self.lam, -b, rcond=None)[0]
            T1 = self.w_T_c @ po.v2T(dx)
        if n_inl == n_inside:
            self.thr = p.min_kernel_threshold
        elif n_inl < p.min_inliers:
            if self.thr + 10 <= p.max_kernel_threshold:
                self.thr += 10
        elif self.thr - 10 >= p.min_kernel_threshold:
            self.thr -= 10
        self.it += 1
        if self.it > 1:
            slope = 1e3 if (error == np.inf or self.error_prev == np.inf) else abs(self.error_prev - error)
            self.ring[self.it % 10] = slope
            self.mean_slope, self.sigma_slope = np.mean(self.ring), np.std(self.ring)
        self.stuck = self.stuck + 1 if (done and self.mean_slope < 1e-2 and self.sigma_slope < 1e-2) else 0
        self.flicker = self.flicker + 1 if (done and self.mean_slope > 1 and self.sigma_slope > 1) else 0
        if self.lam / 2 >= p.min_damping and (self.stuck > 10 or (self.stuck == 0 and self.flicker == 0)):
            self.lam /= 2
        if self.lam * 2 <= p.max_damping and self.flicker > 10:
            self.lam *= 2
        if done and (error < 0.01 or self.stuck > 100) and not p.disable_stop:
            self.stop = True
        if self.it >= p.num_iterations:
            self.stop = True
        self.w_T_c, self.c_T_w, self.error_prev = T1, np.linalg.inv(T1), error
        for k, v in (("T", T1), ("error", error), ("num_inliers", n_inl), ("kernel_threshold", self.thr),
                     ("dumping_factor", self.lam), ("n_inside", int(n_inside))):
            self.hist[k].append(v)

    def stopped(self):
        return self.stop

    @property
    def state(self):                                       # where check_replicas puts its tensor
        return torch.zeros(1)

    def state_checksum(self):
        """Checksum of the replicated state's bit patterns (sharded.check_replicas)."""
        vals = np.concatenate([np.asarray(self.w_T_c, dtype=np.float64).reshape(-1), self.ring,
                               [self.thr, self.lam, self.error_prev, self.mean_slope, self.sigma_slope, self.stuck,
                                self.flicker, self.it, self.mode, float(self.stop)]]).astype(np.float64)
        bits = vals.view(np.int64).astype(object)
        return int(sum(int(b) * (2 * i + 1) for i, b in enumerate(bits))) & ((1 << 62) - 1)

    # Whole rounds with the exchange inside, as the library's picp_sharded_rounds does with its own
    # communicator: lets the driver's "native" control flow (sharded._run_native) run on CPU ranks.
    comm_world = 0                                         # set by the test to the number of ranks

    def rounds(self, n):
        import torch.distributed as dist
        for _ in range(int(n)):
            s1 = self.stage1()
            if self.comm_world > 1:
                dist.all_reduce(s1, op=dist.ReduceOp.SUM)
            s2 = self.stage2(s1)
            if self.comm_world > 1:
                dist.all_reduce(s2, op=dist.ReduceOp.SUM)
            self.solve(s1, s2)

    def finish(self):
        best = int(np.argmin(self.hist["error"]))
        self.hist["best_index"] = best
        if self.hist["error"][best] > 5:
            return self.start, False, self.hist
        return self.hist["T"][best], True, self.hist
