"""Host side of the inverse-dynamics producers (include/osot_mi355x.h: osot_id_rows, osot_computed_torque): the model
quantities the reference asks XBot::ModelInterface for (inertia matrix, non-linear term, contact Jacobians;
DynamicFeasibility.cpp:24-36, TorqueLimits.cpp:27-40, InverseDynamics.cpp:67-77) as device tensors, and the ctypes plumbing
that points the producer at the row ranges of a BatchedStack's A_k / C buffers.  Plumbing only: the arithmetic is in
csrc/osot_id.h."""
import ctypes as C

import numpy as np
import torch

from . import abi


def _t(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(device)


class IdModel:
    """B [B][nv][nv], h [B][nv], Jc [B][contacts][3 or 6][nv]; x = [qddot; forces] (InverseDynamics.cpp:12-28)"""

    def __init__(self, Bm, h, Jc, device=0, floating_base=True):
        self.device = torch.device("cuda", device)
        self.Bm, self.h, self.Jc = _t(Bm, self.device), _t(h, self.device), _t(Jc, self.device)
        self.B, self.nv = self.Bm.shape[0], self.Bm.shape[1]
        self.n_contacts, self.cdim = self.Jc.shape[1], self.Jc.shape[2]
        self.n = self.nv + self.n_contacts * self.cdim
        self.floating_base = floating_base
        self._lib = abi.lib()

    def _c(self):
        m = abi.IdModel()
        m.B, m.nv, m.n_contacts, m.contact_dim = self.B, self.nv, self.n_contacts, self.cdim
        m.Bm, m.h, m.Jc = self.Bm.data_ptr(), self.h.data_ptr(), self.Jc.data_ptr()
        m.floating_base = 1 if self.floating_base else 0
        return m

    def write_rows(self, stack, dyn_block=None, tau_block=None, tasks=()):
        """dyn_block / tau_block: indices of the plan's DYN_FEASIBILITY / TORQUE_LIMITS row blocks; tasks: (level, first row,
        J [B][rows][nv]) -- the [J 0] task matrices.  Everything lands in stack.C / stack.A[level] in place."""
        plan, n = stack.plan, stack.plan.n
        assert n == self.n
        cs = plan.nc_stored * n
        pd = lambda j: None if j is None else C.c_void_p(stack.C.data_ptr() + 8 * plan.rows_stored_offset(j) * n)
        nt = len(tasks)
        vp = C.c_void_p
        Jp = (vp * max(nt, 1))(); Jr = (C.c_int * max(nt, 1))(); Ad = (vp * max(nt, 1))(); As = (C.c_longlong * max(nt, 1))()
        keep = []
        for i, (k, row0, J) in enumerate(tasks):
            J = _t(J, self.device); keep.append(J)
            Jp[i], Jr[i] = J.data_ptr(), J.shape[1]
            Ad[i] = stack.A[k].data_ptr() + 8 * row0 * n
            As[i] = plan.ma(k) * n
        m = self._c()
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        abi.check(self._lib.osot_id_rows(C.byref(m), pd(dyn_block), cs, pd(tau_block), cs, nt, Jp, Jr, Ad, As, st), "osot_id_rows")

    def computed_torque(self, x, fb_tol=10e-3):
        """InverseDynamics::computedTorque: (tau [B][nv], ok [B]) for the solved x [B][n]; ok = 0 where a floating-base row
        of tau exceeds fb_tol (the reference's 10e-3, InverseDynamics.cpp:87)"""
        B = x.shape[0]
        x = x.contiguous()
        tau = torch.empty((B, self.nv), dtype=torch.float64, device=self.device)
        ok = torch.empty((B,), dtype=torch.int32, device=self.device)
        m = self._c(); m.B = B
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        abi.check(self._lib.osot_computed_torque(C.byref(m), C.c_void_p(x.data_ptr()), C.c_void_p(tau.data_ptr()),
                                                 C.c_void_p(ok.data_ptr()), fb_tol, st), "osot_computed_torque")
        return tau, ok


def force_gains(J, Bi, Kp, Kd, p0, rows, f_virtual=None, a_ref=None):
    """GainType::Force of acceleration::Cartesian (src/tasks/acceleration/Cartesian.cpp:161-169): Mi = J Bi J' per instance
    (compute_cartesian_inertia_inverse, :517-524), Gp = Mi Kp and Gd = Mi Kd written into the task's leaf array
    p0 [B][2 rows + 2 rows^2] behind the errors (Task.acc_gain_matrices), a_ref [B][rows] += Mi f_virtual.
    J [B][rows][nv], Bi [B][nv][nv] device tensors; Kp, Kd rows x rows (host)."""
    B, r, nv = J.shape
    assert r == rows and p0.shape == (B, 2 * rows + 2 * rows * rows) and p0.is_contiguous() and J.is_contiguous() and Bi.is_contiguous()
    Kp = np.ascontiguousarray(Kp, dtype=np.float64).reshape(rows * rows)
    Kd = np.ascontiguousarray(Kd, dtype=np.float64).reshape(rows * rows)
    vp = C.c_void_p
    st = vp(torch.cuda.current_stream(J.device).cuda_stream)
    abi.check(abi.lib().osot_id_force_gains(B, nv, rows, vp(J.data_ptr()), vp(Bi.data_ptr()), Kp.ctypes.data_as(abi.dp),
                                            Kd.ctypes.data_as(abi.dp), vp(f_virtual.data_ptr()) if f_virtual is not None else None,
                                            vp(p0.data_ptr() + 8 * 2 * rows), 2 * rows + 2 * rows * rows,
                                            vp(a_ref.data_ptr()) if a_ref is not None else None, st), "osot_id_force_gains")
