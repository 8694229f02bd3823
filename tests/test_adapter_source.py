"""adapters/opensot_backend/MI355XBackEnd.cpp -- the BackEnd subclass + create_instance a maintainer adds to OpenSoT
(src/solvers/QPOasesBackEnd.cpp:14-24, include/OpenSoT/solvers/BackEnd.h:125-171) -- held against the C-ABI it calls:
CPU: every osot_* symbol it uses is declared in include/osot_mi355x.h, listed in opensot_amd.abi.SYMBOLS and exported by the built
library; every pure virtual of BackEnd is overridden; it COMPILES (-fsyntax-only) against the test doubles of tests/adapter_mock (and
against the real headers where an OpenSoT tree with Eigen / Boost exists).  GPU: the compiled plugin, loaded by dlopen + create_instance
like OpenSoT's factory does, answers the reference's robot-free known-answer tests."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from opensot_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "adapters", "opensot_backend", "MI355XBackEnd.cpp")
MOCK = os.path.join(ROOT, "tests", "adapter_mock")


def _code():
    txt = open(SRC).read()
    return re.sub(r"//[^\n]*", "", txt)          # comments out


def test_adapter_calls_only_declared_and_exported_symbols():
    used = sorted(set(re.findall(r"\b(osot_[a-z_]+)\s*\(", _code())))
    assert "osot_backend_create" in used and "osot_backend_solve" in used and len(used) >= 12
    header = open(os.path.join(ROOT, "include", "osot_mi355x.h")).read()
    lib = abi.lib()
    for sym in used:
        assert re.search(r"\b" + sym + r"\s*\(", header), f"{sym} is not declared in include/osot_mi355x.h"
        assert sym in abi.SYMBOLS, f"{sym} is not in opensot_amd.abi.SYMBOLS"
        assert hasattr(lib, sym), f"{sym} is not exported by libosot_mi355x.so"
    # the struct it passes through boost::any exists with the fields the INTEGRATION table names
    assert re.search(r"typedef struct \{ int max_iterations; int last_iterations; int last_status; \} osot_backend_options;", header)


def test_adapter_overrides_every_pure_virtual_and_exports_the_factory_symbols():
    code = _code()
    for sig in (r"bool\s+initProblem\s*\(", r"bool\s+solve\s*\(\s*\)\s*override", r"boost::any\s+getOptions\s*\(\s*\)\s*override",
                r"void\s+setOptions\s*\(\s*const\s+boost::any&", r"double\s+getObjective\s*\(\s*\)\s*override",
                r"bool\s+setEpsRegularisation\s*\(", r"double\s+getEpsRegularisation\s*\("):
        assert re.search(sig, code), sig
    assert re.search(r'extern "C" OpenSoT::solvers::BackEnd\* create_instance\(const int number_of_variables, const int number_of_constraints,\s*'
                     r"OpenSoT::HessianType hessian_type, const double eps_regularisation\)", code)
    assert re.search(r'extern "C" void destroy_instance\(OpenSoT::solvers::BackEnd\* instance\)', code)
    assert "_solution" in code          # BackEnd::getSolution() returns the protected member: the plugin must write it (BackEnd.h:23)
    # INTEGRATION.md quotes the file instead of carrying a copy
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "adapters/opensot_backend/MI355XBackEnd.cpp" in integ and "class MI355XBackEnd : public BackEnd {" not in integ


def test_adapter_compiles():
    """-fsyntax-only against the test doubles (argument types of every osot_backend_* call are checked by the compiler against
    include/osot_mi355x.h), and against a real OpenSoT tree when OPENSOT_INCLUDE / EIGEN_INCLUDE name one"""
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + MOCK, "-I" + os.path.join(ROOT, "include"), SRC]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    real = [os.environ.get("OPENSOT_INCLUDE"), os.environ.get("EIGEN_INCLUDE")]
    if all(real) and all(os.path.isdir(p) for p in real):
        r = subprocess.run(["g++", "-std=c++20", "-fsyntax-only"] + ["-I" + p for p in real] + ["-I" + os.path.join(ROOT, "include"), SRC],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_compiled_adapter_answers_the_reference_known_answers(gpu_device, tmp_path):
    """the plugin built for real (test-double matrix classes in place of Eigen) and loaded like BackEndFactory.cpp:4-17 loads a
    back-end; TestQPOases.cpp:208-254 and :346-412 through initProblem / updateConstraints / updateTask / solve / getSolution"""
    so = str(tmp_path / "libOpenSotBackEndODYS.so")
    exe = str(tmp_path / "run_adapter")
    libdir = os.path.join(ROOT, "opensot_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I" + MOCK, "-I" + os.path.join(ROOT, "include"), SRC, "-o", so,
                           "-L" + libdir, "-losot_mi355x", "-Wl,-rpath," + libdir])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + MOCK, os.path.join(MOCK, "run_adapter.cpp"), "-o", exe, "-ldl"])
    out = subprocess.run([exe, so], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["ok"] == [1, 1, 1, 1]
    np.testing.assert_allclose(r["s1"], [10 / 3.0] * 3, atol=1e-3)
    np.testing.assert_allclose(r["s2"], [10, -10, 10], atol=1e-6)
    np.testing.assert_allclose(r["s3"], [5, -5], atol=1e-14)
    np.testing.assert_allclose(r["s4"], [1, -1], atol=1e-14)
    assert abs(r["f4"] + 1.0) < 1e-12
    assert abs(r["eps"] - 2.221e-13 * 1e4) / (2.221e-13 * 1e4) < 1e-3      # 1e3 * EPS * factor (QPOasesBackEnd.cpp:57, 67)
