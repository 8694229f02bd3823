// adapters/opensot_backend/MI355XBackEnd.cpp -- the OpenSoT back-end plugin over libosot_mi355x.so.
//
// What it mirrors: the factory symbol `create_instance` / `destroy_instance` and the BackEnd subclass every OpenSoT back-end
// is (src/solvers/QPOasesBackEnd.cpp:14-24; include/OpenSoT/solvers/BackEnd.h:23-171).  OpenSoT loads back-ends by NAME:
// BackEndFactory(be, nV, nC, hessianType, eps) -> dlopen("libOpenSotBackEnd<Name>.so") -> create_instance
// (src/solvers/BackEndFactory.cpp:4-17).  solver_back_ends::ODYS has an enumerator and a factory branch
// (include/OpenSoT/solvers/BackEndFactory.h:15, src/solvers/BackEndFactory.cpp:62-68) but no implementation or CMake target in the
// tree, so this file installed as libOpenSotBackEndODYS.so (CMakeLists.txt beside it) needs NO change to OpenSoT's sources: a stack
// selects it with iHQP(stack, eps, solver_back_ends::ODYS) (include/OpenSoT/solvers/iHQP.h:59-108).
//
// It is compiled INSIDE an OpenSoT build tree: the object that crosses create_instance is a C++ class with Eigen members and
// boost::any virtuals, so it needs the host's Eigen / Boost / OpenSoT headers and flags (e.g. EIGEN_DONT_VECTORIZE,
// CMakeLists.txt:54-57).  This repository's image has none of them -- tests/test_adapter_source.py checks this file against
// include/osot_mi355x.h (every osot_* symbol it calls is declared there and exported by the library; every pure virtual of
// BackEnd.h:125-150 is overridden) and compiles it with -fsyntax-only wherever the OpenSoT headers exist.
//
// Eigen's default storage is column-major; the C-ABI takes row-major: hence the RowMajor copies (the qpOASES wrapper does the
// same, src/solvers/QPOasesBackEnd.cpp:128, 257).
#include <OpenSoT/solvers/BackEnd.h>
#include <osot_mi355x.h>
#include <boost/any.hpp>
#include <stdexcept>

namespace OpenSoT { namespace solvers {

class MI355XBackEnd : public BackEnd {
    typedef Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> MatrixRM;
    osot_backend* _be = nullptr;
    MatrixRM _H_rm, _A_rm;
    double _eps;

public:
    MI355XBackEnd(int nv, int nc, HessianType ht, double eps) : BackEnd(nv, nc), _eps(eps) {
        if (osot_backend_create(nv, nc, (int)ht, eps, &_be) != OSOT_OK)
            throw std::runtime_error(osot_last_error());
    }
    ~MI355XBackEnd() override { osot_backend_destroy(_be); }

    bool initProblem(const Eigen::MatrixXd& H, const Eigen::VectorXd& g, const Eigen::MatrixXd& A,
                     const Eigen::VectorXd& lA, const Eigen::VectorXd& uA,
                     const Eigen::VectorXd& l, const Eigen::VectorXd& u) override {
        if (A.rows() != _A.rows()) return false;                       // QPOasesBackEnd.cpp:91-94
        _H = H; _g = g; _A = A; _lA = lA; _uA = uA; _l = l; _u = u;
        _H_rm = _H; _A_rm = _A;
        if (osot_backend_init_problem(_be, _H_rm.data(), _g.data(), _A_rm.data(), _lA.data(), _uA.data(),
                                      _l.size() ? _l.data() : nullptr, _u.size() ? _u.data() : nullptr) != OSOT_OK)
            return false;
        _solution.resize(_g.size());
        return osot_backend_get_solution(_be, _solution.data()) == OSOT_OK;   // BackEnd::getSolution() reads _solution (BackEnd.h:23)
    }

    bool solve() override {
        _H_rm = _H; _A_rm = _A;                                       // updateTask / updateConstraints stored copies (BackEnd.cpp:19-93)
        if (osot_backend_update_task(_be, _H_rm.data(), _g.data()) != OSOT_OK) return false;
        if (osot_backend_update_constraints(_be, _A_rm.data(), _lA.data(), _uA.data(), (int)_A.rows()) != OSOT_OK) return false;
        if (osot_backend_update_bounds(_be, _l.size() ? _l.data() : nullptr, _u.size() ? _u.data() : nullptr) != OSOT_OK) return false;
        if (osot_backend_solve(_be) != OSOT_OK) return false;         // infeasible / iteration cap -> false, like qpOASES' status
        return osot_backend_get_solution(_be, _solution.data()) == OSOT_OK;
    }

    boost::any getOptions() override {
        osot_backend_options o;
        osot_backend_get_options(_be, &o);
        return boost::any(o);
    }
    // (the qpOASES back-end carries qpOASES::Options here, QPOasesBackEnd.cpp:309-318; this one its own POD: the iteration cap, the
    //  counterpart of nWSR)
    void setOptions(const boost::any& a) override {
        if (a.type() == typeid(osot_backend_options)) {
            const osot_backend_options o = boost::any_cast<osot_backend_options>(a);
            osot_backend_set_options(_be, &o);
        }
    }
    double getObjective() override { double f = 0; osot_backend_get_objective(_be, &f); return f; }
    bool setEpsRegularisation(const double eps) override { return osot_backend_set_eps_regularisation(_be, eps) == OSOT_OK; }
    double getEpsRegularisation() override { double e = 0; osot_backend_get_eps_regularisation(_be, &e); return e; }
};

}}  // namespace OpenSoT::solvers

extern "C" OpenSoT::solvers::BackEnd* create_instance(const int number_of_variables, const int number_of_constraints,
                                                     OpenSoT::HessianType hessian_type, const double eps_regularisation) {
    return new OpenSoT::solvers::MI355XBackEnd(number_of_variables, number_of_constraints, hessian_type, eps_regularisation);
}
extern "C" void destroy_instance(OpenSoT::solvers::BackEnd* instance) { delete instance; }
