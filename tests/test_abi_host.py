"""C-ABI checks that need no GPU: the library loads, exports every symbol include/osot_mi355x.h declares,
validates plans and refuses bad arguments with the documented codes."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from opensot_amd import abi, synth
from opensot_amd.plan import Bound, Rows, StackPlan, Task, eps_abs_from_factor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(abi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return abi.lib()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "osot_mi355x.h")).read()
    declared = set(re.findall(r"\b(osot_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(abi.SYMBOLS), declared ^ set(abi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_struct_layouts_match_the_compiled_header(lib):
    """opensot_amd/abi.py mirrors the header's structs by hand: size and the offset of EVERY member against what the library
    was compiled with (osot_abi_layout), and every struct typedef of the header has a mirror"""
    hdr = open(os.path.join(ROOT, "include", "osot_mi355x.h")).read()
    typedefs = set(re.findall(r"\}\s*(osot_[a-z_0-9]+)\s*;", hdr)) - {"osot_task_kind", "osot_bound_kind", "osot_rows_kind"}
    assert typedefs == set(abi.STRUCTS), typedefs ^ set(abi.STRUCTS)
    lib.osot_abi_layout.argtypes = [C.c_char_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int, C.POINTER(C.c_int)]
    for name, cls in abi.STRUCTS.items():
        size, nf = C.c_ulonglong(), C.c_int()
        offs = (C.c_ulonglong * 64)()
        assert lib.osot_abi_layout(name.encode(), C.byref(size), offs, 64, C.byref(nf)) == abi.OK, name
        assert size.value == C.sizeof(cls), (name, size.value, C.sizeof(cls))
        mine = [getattr(cls, f[0]).offset for f in cls._fields_]
        assert nf.value == len(mine), (name, nf.value, len(mine))
        assert list(offs[:nf.value]) == mine, (name, list(offs[:nf.value]), mine)
    assert lib.osot_abi_layout(b"no_such_struct", C.byref(size), None, 0, None) == abi.ERR_INVALID


def test_version_and_error_text(lib):
    assert b"gfx950" in lib.osot_version()
    assert lib.osot_plan_validate(None) == abi.ERR_INVALID
    assert b"null" in lib.osot_last_error()


def test_plan_sizes(lib):
    plan, _ = synth.make_velocity_stack("C4", 1)
    p = plan.to_c()
    assert lib.osot_plan_validate(C.byref(p)) == abi.OK
    m, ma = C.c_int(), C.c_int()
    want = [(3, 3), (24, 24), (32, 0)]
    for k in range(3):
        assert lib.osot_plan_level_rows(C.byref(p), k, C.byref(m), C.byref(ma)) == abi.OK
        assert (m.value, ma.value) == want[k] == (plan.m(k), plan.ma(k))
    assert lib.osot_plan_level_rows(C.byref(p), 3, C.byref(m), C.byref(ma)) == abi.ERR_INVALID
    nc = C.c_int()
    assert lib.osot_plan_constraint_rows(C.byref(p), C.byref(nc)) == abi.OK and nc.value == 16 == plan.nc


def test_plan_validation_errors(lib):
    def code(plan):
        p = plan.to_c() if isinstance(plan, StackPlan) else plan
        return lib.osot_plan_validate(C.byref(p))
    good = StackPlan(n=8, levels=[[Task(abi.TASK_GENERIC, 3)], [Task(abi.TASK_POSTURAL, 8)]])
    assert code(good) == abi.OK
    p = good.to_c(); p.n = 65
    assert code(p) == abi.ERR_INVALID
    p = good.to_c(); p.level[0].task[0].kind = 9
    assert code(p) == abi.ERR_UNSUPPORTED
    p = good.to_c(); p.level[0].task[0].rows = 0
    assert code(p) == abi.ERR_INVALID
    p = good.to_c(); p.level[1].task[0].rows = 9          # Postural [I_rows 0] cannot exceed n rows
    assert code(p) == abi.ERR_INVALID
    p = good.to_c(); p.eps_abs = -1.0
    assert code(p) == abi.ERR_INVALID
    # Postural not last in its level
    p = StackPlan(n=8, levels=[[Task(abi.TASK_GENERIC, 3), Task(abi.TASK_GENERIC, 2)]]).to_c()
    p.level[0].task[0].kind = abi.TASK_POSTURAL; p.level[0].task[0].rows = 8
    assert code(p) == abi.ERR_UNSUPPORTED


def test_backend_argument_checks_without_gpu(lib):
    h = C.c_void_p()
    assert lib.osot_backend_create(0, 0, abi.HST_SEMIDEF, 1.0, C.byref(h)) == abi.ERR_INVALID
    assert lib.osot_backend_create(abi.MAX_QP_VARS + 1, 0, abi.HST_SEMIDEF, 1.0, C.byref(h)) == abi.ERR_INVALID      # (128 since round 6: osot_qp_big.h)
    assert lib.osot_backend_create(65, 0, abi.HST_SEMIDEF, 1.0, C.byref(h)) == abi.OK and lib.osot_backend_destroy(h) == abi.OK
    assert lib.osot_backend_create(3, -1, abi.HST_SEMIDEF, 1.0, C.byref(h)) == abi.ERR_INVALID
    assert lib.osot_backend_create(3, 1, abi.HST_SEMIDEF, -1.0, C.byref(h)) == abi.ERR_INVALID
    assert lib.osot_backend_create(3, 1, abi.HST_SEMIDEF, 1.0, C.byref(h)) == abi.OK
    e = C.c_double()
    assert lib.osot_backend_get_eps_regularisation(h, C.byref(e)) == abi.OK
    assert e.value == pytest.approx(2.221e-13, rel=1e-12)       # TestQPOases.cpp:798-836
    assert lib.osot_backend_set_eps_regularisation(h, -1.0) == abi.ERR_INVALID  # "Negative eps is not allowed!"
    l = np.array([1.0, 0, 0]); u = np.zeros(3)
    dp = abi.dp
    assert lib.osot_backend_update_bounds(h, l.ctypes.data_as(dp), u.ctypes.data_as(dp)) == abi.ERR_INVALID
    assert lib.osot_backend_solve(h) == abi.ERR_INVALID          # solve() before initProblem()
    nv, nc = C.c_int(), C.c_int()
    lib.osot_backend_get_num_variables(h, C.byref(nv)); lib.osot_backend_get_num_constraints(h, C.byref(nc))
    assert (nv.value, nc.value) == (3, 1)
    assert lib.osot_backend_destroy(h) == abi.OK


def test_control_cycle_argument_checks_without_gpu(lib):
    """osot_control_cycle (round 4) refuses null arguments before it touches a device"""
    kb, lb, out, qb = abi.KinBatch(), abi.LeafBatch(), abi.AssembledOut(), abi.QpBatch()
    assert lib.osot_control_cycle(None, None, C.byref(kb), C.byref(lb), C.byref(out), C.byref(qb), None, None) == abi.ERR_INVALID
    # osot_control_rollout (round 5): a rollout has at least one step, several steps need q_integrate, then the same checks
    assert lib.osot_control_rollout(None, None, C.byref(kb), C.byref(lb), C.byref(out), C.byref(qb), None, 0, None, None, None) == abi.ERR_INVALID
    assert b"at least one step" in lib.osot_last_error()
    assert lib.osot_control_rollout(None, None, C.byref(kb), C.byref(lb), C.byref(out), C.byref(qb), None, 4, None, None, None) == abi.ERR_INVALID
    assert b"integrates q" in lib.osot_last_error()
    assert lib.osot_control_rollout(None, None, C.byref(kb), C.byref(lb), C.byref(out), C.byref(qb), None, 1, None, None, None) == abi.ERR_INVALID
    assert b"null argument" in lib.osot_last_error()


def test_eps_factor_convention():
    assert eps_abs_from_factor(1.0) == pytest.approx(2.221e-13)
    assert eps_abs_from_factor(2e2) == pytest.approx(4.442e-11)   # iHQP default (iHQP.h:32)
    assert eps_abs_from_factor(1e6) == pytest.approx(2.221e-7)    # benchmark value (coman_ik.cpp:453)


def test_subtask_plan_validation(lib):
    """SubTask blocks (SubTask.cpp:22-112): rows must equal popcount(row_mask), the mask must stay inside the parent,
    and a Postural sub-task counts as stored rows (a whole Postural block is implicit)"""
    from opensot_amd.plan import StackPlan, Task, subtask, eps_abs_from_factor
    import ctypes as C
    n = 12
    cart = Task(abi.TASK_CARTESIAN, 6, name="c")
    post = Task(abi.TASK_POSTURAL, n, name="p")
    plan = StackPlan(n=n, levels=[[subtask(cart, [0, 1, 2], lam=0.5)], [subtask(post, range(6, n))]], bounds=[], rowblocks=[],
                     eps_abs=eps_abs_from_factor(1e6))
    d = plan.to_c()
    assert lib.osot_plan_validate(C.byref(d)) == abi.OK
    m, ma = C.c_int(), C.c_int()
    assert lib.osot_plan_level_rows(C.byref(d), 1, C.byref(m), C.byref(ma)) == abi.OK and (m.value, ma.value) == (6, 6)
    whole = StackPlan(n=n, levels=[[cart], [post]], bounds=[], rowblocks=[], eps_abs=eps_abs_from_factor(1e6)).to_c()
    assert lib.osot_plan_level_rows(C.byref(whole), 1, C.byref(m), C.byref(ma)) == abi.OK and (m.value, ma.value) == (n, 0)
    d.level[0].task[0].rows = 2                       # popcount(row_mask) = 3
    assert lib.osot_plan_validate(C.byref(d)) == abi.ERR_INVALID
    d.level[0].task[0].rows = 3
    d.level[0].task[0].row_mask = 0b1000011           # bit 6 is beyond a Cartesian task's 6 rows
    assert lib.osot_plan_validate(C.byref(d)) == abi.ERR_INVALID


def test_regularisation_task_plan_validation(lib):
    """AutoStack::setRegularisationTask: identity-Jacobian kinds only (the cost joins the diagonal of every level's H),
    rows in 1..n, no sub-task; anything else is refused, not approximated"""
    from opensot_amd.plan import StackPlan, Task, eps_abs_from_factor
    import ctypes as C
    n = 12
    plan = StackPlan(n=n, levels=[[Task(abi.TASK_GENERIC, 5, name="g")]], bounds=[], rowblocks=[],
                     eps_abs=eps_abs_from_factor(1e6), regularisation=Task(abi.TASK_GENERIC, n, weight=1e-2, name="minvel"))
    d = plan.to_c()
    assert d.has_regularisation == 1 and d.regularisation.rows == n
    assert lib.osot_plan_validate(C.byref(d)) == abi.OK
    d.regularisation.kind = abi.TASK_CARTESIAN                # dense Jacobian: not expressible
    assert lib.osot_plan_validate(C.byref(d)) == abi.ERR_UNSUPPORTED
    d.regularisation.kind = abi.TASK_POSTURAL
    d.regularisation.rows = n + 1
    assert lib.osot_plan_validate(C.byref(d)) == abi.ERR_INVALID
    d.regularisation.rows = n
    d.regularisation.row_mask = 0b11
    assert lib.osot_plan_validate(C.byref(d)) == abi.ERR_UNSUPPORTED
    d.regularisation.row_mask = 0
    d.regularisation.weight = -1.0
    assert lib.osot_plan_validate(C.byref(d)) == abi.ERR_INVALID
    d.has_regularisation = 0                                   # ignored when absent
    assert lib.osot_plan_validate(C.byref(d)) == abi.OK


def test_task_local_rows_plan_validation(lib):
    """row blocks tagged with a level (`task << constraint`): only_level = level + 1, within the plan's levels"""
    from opensot_amd.plan import StackPlan, Task, Rows, eps_abs_from_factor
    import ctypes as C
    plan = StackPlan(n=8, levels=[[Task(abi.TASK_GENERIC, 3, name="a")], [Task(abi.TASK_GENERIC, 2, name="b")]], bounds=[],
                     rowblocks=[Rows(abi.ROWS_GENERIC, 2, name="global"), Rows(abi.ROWS_GENERIC, 3, name="local", level=1)],
                     eps_abs=eps_abs_from_factor(1e6))
    d = plan.to_c()
    assert (d.rowblock[0].only_level, d.rowblock[1].only_level) == (0, 2)
    assert lib.osot_plan_validate(C.byref(d)) == abi.OK
    d.rowblock[1].only_level = 3          # there is no level 2
    assert lib.osot_plan_validate(C.byref(d)) == abi.ERR_INVALID
    d.rowblock[1].only_level = -1
    assert lib.osot_plan_validate(C.byref(d)) == abi.ERR_INVALID


def test_task_to_constraint_rows_plan_validation(lib):
    """constraints::TaskToConstraint (`stack << l_sole`, TaskToConstraint.cpp:25-68) as row kinds: sizes of the
    underlying task, err_ub >= err_lb (the reference throws otherwise, :43)"""
    from opensot_amd.plan import StackPlan, Task, Rows, eps_abs_from_factor
    import ctypes as C
    plan = StackPlan(n=12, levels=[[Task(abi.TASK_GENERIC, 3, name="a")]], bounds=[],
                     rowblocks=[Rows(abi.ROWS_TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Rows(abi.ROWS_TASK_COM, 3, lam=0.2, err_lb=-0.01, err_ub=0.02, name="com_band")],
                     eps_abs=eps_abs_from_factor(1e6))
    d = plan.to_c()
    assert lib.osot_plan_validate(C.byref(d)) == abi.OK
    nc, ns = C.c_int(), C.c_int()
    assert lib.osot_plan_constraint_rows(C.byref(d), C.byref(nc)) == abi.OK and lib.osot_plan_stored_constraint_rows(C.byref(d), C.byref(ns)) == abi.OK
    assert (nc.value, ns.value) == (9, 9)
    d.rowblock[0].rows = 5
    assert lib.osot_plan_validate(C.byref(d)) == abi.ERR_INVALID
    d.rowblock[0].rows = 6
    d.rowblock[1].err_ub[2] = -0.02      # one component of err_ub below err_lb (the band is per row)
    assert lib.osot_plan_validate(C.byref(d)) == abi.ERR_INVALID


def test_unit_row_block_plan_validation(lib):
    """OSOT_ROWS_UNIT_GENERIC: rows e_(first_col + i) without storage; the range must lie inside the variables"""
    from opensot_amd.plan import StackPlan, Task, Rows, eps_abs_from_factor
    import ctypes as C
    plan = StackPlan(n=10, levels=[[Task(abi.TASK_GENERIC, 3, name="a")]], bounds=[],
                     rowblocks=[Rows(abi.ROWS_UNIT_GENERIC, 4, first_col=6, level=0, name="local_box"), Rows(abi.ROWS_GENERIC, 2, name="rows")],
                     eps_abs=eps_abs_from_factor(1e6))
    d = plan.to_c()
    assert lib.osot_plan_validate(C.byref(d)) == abi.OK
    nc, ns = C.c_int(), C.c_int()
    lib.osot_plan_constraint_rows(C.byref(d), C.byref(nc)); lib.osot_plan_stored_constraint_rows(C.byref(d), C.byref(ns))
    assert (nc.value, ns.value) == (6, 2) and plan.nc_stored == 2
    d.rowblock[0].first_col = 7           # 7 + 4 > 10
    assert lib.osot_plan_validate(C.byref(d)) == abi.ERR_INVALID
