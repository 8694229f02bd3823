// oracle/ref_qpoases_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A plain C-ABI around the reference's vendored qpOASES 3.1, compiled IN PLACE from
// /root/reference/external/qpOASES-ext/src/*.cpp by oracle/Makefile into oracle/_ref/.
// No reference source is copied into this repository; this file is our own code and
// only *calls* qpOASES::SQProblem the way the reference's qpOASES back-end does:
//
//   * option set ............ src/solvers/QPOasesBackEnd.cpp:51-76  (setToMPC, PL_NONE,
//                             enableRegularisation=FALSE, epsRegularisation*=factor,
//                             numRegularisationSteps=0, numRefinementSteps=1)
//   * init (cold) ........... src/solvers/QPOasesBackEnd.cpp:86-168
//   * per-cycle solve ....... src/solvers/QPOasesBackEnd.cpp:248-307 (clamp +-1e20, H_ii+=eps,
//                             hotstart nWSR=13200 -> warm init -> cold init fallback chain)
//   * +-inf clamp ........... src/solvers/QPOasesBackEnd.cpp:339-356
//   * re-allocation on row-count change ... src/solvers/QPOasesBackEnd.cpp:229-244
//
// Matrices cross this ABI ROW-MAJOR (qpOASES' native layout, QPOasesBackEnd.cpp:128,257);
// H is symmetric so its layout is irrelevant.
#include <qpOASES.hpp>
#include <vector>
#include <cstring>
#include <memory>
#include <cstdio>

namespace {

struct RefQP {
    int nV, nC;
    int hessian_type;
    double eps_abs;           // absolute epsilon added to diag(H) (= 1e3*EPS*factor)
    bool has_bounds;
    std::unique_ptr<qpOASES::SQProblem> prob;
    qpOASES::Options opt;
    qpOASES::Bounds bounds;
    qpOASES::Constraints constraints;
    std::vector<double> H, g, A, lA, uA, l, u, x, y;
    int last_nwsr;
    int n_fallback_warm, n_fallback_cold;
};

// qpOASES routes messages through ONE global handler whose visibility every QProblemB constructor /
// setPrintLevel call rewrites; with several host threads that races and INFO lines leak to stdout.
// Sending the handler's output to /dev/null keeps the batch driver quiet (and its timing honest).
void silence_messages() {
    static FILE* devnull = fopen("/dev/null", "w");
    if (devnull) qpOASES::getGlobalMessageHandler()->setOutputFile(devnull);
}

void make_problem(RefQP* q) {
    silence_messages();
    q->prob.reset(new qpOASES::SQProblem(q->nV, q->nC, (qpOASES::HessianType)q->hessian_type));
    q->prob->setOptions(q->opt);
}

void clamp_infty(RefQP* q) {
    for (size_t i = 0; i < q->lA.size(); ++i) {
        if (q->lA[i] < -qpOASES::INFTY) q->lA[i] = -qpOASES::INFTY;
        if (q->uA[i] >  qpOASES::INFTY) q->uA[i] =  qpOASES::INFTY;
    }
    for (size_t i = 0; i < q->l.size(); ++i) {
        if (q->l[i] < -qpOASES::INFTY) q->l[i] = -qpOASES::INFTY;
        if (q->u[i] >  qpOASES::INFTY) q->u[i] =  qpOASES::INFTY;
    }
}

void add_eps(RefQP* q) {
    for (int i = 0; i < q->nV; ++i) q->H[(size_t)i * q->nV + i] += q->eps_abs;
}

int cold_init(RefQP* q) {
    // mirrors QPOasesBackEnd::initProblem: H_ii += eps, clamp, init(), read back.
    add_eps(q);
    clamp_infty(q);
    int nWSR = 13200;
    qpOASES::returnValue val = q->prob->init(q->H.data(), q->g.data(),
                                             q->nC ? q->A.data() : 0,
                                             q->has_bounds ? q->l.data() : 0,
                                             q->has_bounds ? q->u.data() : 0,
                                             q->nC ? q->lA.data() : 0,
                                             q->nC ? q->uA.data() : 0, nWSR, 0);
    q->last_nwsr = nWSR;
    if (qpOASES::getSimpleStatus(val) < 0) return 0;
    if (q->prob->getPrimalSolution(q->x.data()) != qpOASES::SUCCESSFUL_RETURN) return 0;
    q->prob->getDualSolution(q->y.data());
    q->prob->getBounds(q->bounds);
    q->prob->getConstraints(q->constraints);
    return 1;
}

}  // namespace

extern "C" {

// eps_factor has the meaning of BackEndFactory's eps_regularisation argument
// (src/solvers/BackEndFactory.cpp:4-17): absolute eps = 1e3 * EPS * factor.
void* refqp_create(int nV, int nC, int hessian_type, double eps_factor,
                   double termination_tolerance_override /* <=0: keep OpenSoT's option set */) {
    RefQP* q = new RefQP();
    q->nV = nV; q->nC = nC; q->hessian_type = hessian_type;
    q->has_bounds = false;
    q->last_nwsr = 0; q->n_fallback_warm = 0; q->n_fallback_cold = 0;
    qpOASES::Options opt;
    opt.setToMPC();
    opt.printLevel = qpOASES::PL_NONE;
    opt.enableRegularisation = qpOASES::BT_FALSE;
    opt.epsRegularisation *= eps_factor;
    opt.numRegularisationSteps = 0;
    opt.numRefinementSteps = 1;
    if (termination_tolerance_override > 0) opt.terminationTolerance = termination_tolerance_override;
    opt.ensureConsistency();
    q->opt = opt;
    q->eps_abs = opt.epsRegularisation;
    q->H.assign((size_t)nV * nV, 0.0); q->g.assign(nV, 0.0);
    q->A.assign((size_t)nC * nV, 0.0); q->lA.assign(nC, 0.0); q->uA.assign(nC, 0.0);
    q->x.assign(nV, 0.0); q->y.assign(nV + nC, 0.0);
    make_problem(q);
    return q;
}

void refqp_destroy(void* h) { delete (RefQP*)h; }

double refqp_eps_abs(void* h) { return ((RefQP*)h)->eps_abs; }
int refqp_last_nwsr(void* h) { return ((RefQP*)h)->last_nwsr; }
int refqp_fallbacks(void* h, int which) { RefQP* q = (RefQP*)h; return which ? q->n_fallback_cold : q->n_fallback_warm; }

void refqp_update_task(void* h, const double* H, const double* g) {
    RefQP* q = (RefQP*)h;
    std::memcpy(q->H.data(), H, sizeof(double) * q->nV * q->nV);
    std::memcpy(q->g.data(), g, sizeof(double) * q->nV);
}

// returns 1 on success; if the row count changed the SQProblem is rebuilt and cold-initialised
// (QPOasesBackEnd.cpp:229-244), which needs the task/bounds already stored.
int refqp_update_constraints(void* h, const double* A, const double* lA, const double* uA, int nC) {
    RefQP* q = (RefQP*)h;
    bool resized = (nC != q->nC);
    q->nC = nC;
    q->A.assign(A, A + (size_t)nC * q->nV);
    q->lA.assign(lA, lA + nC); q->uA.assign(uA, uA + nC);
    if (resized) {
        q->y.assign(q->nV + nC, 0.0);
        make_problem(q);
        return cold_init(q);
    }
    return 1;
}

void refqp_update_bounds(void* h, const double* l, const double* u) {
    RefQP* q = (RefQP*)h;
    if (!l || !u) { q->has_bounds = false; q->l.clear(); q->u.clear(); return; }
    q->has_bounds = true;
    q->l.assign(l, l + q->nV); q->u.assign(u, u + q->nV);
}

int refqp_init(void* h, const double* H, const double* g, const double* A, const double* lA,
               const double* uA, const double* l, const double* u) {
    RefQP* q = (RefQP*)h;
    refqp_update_task(h, H, g);
    if (q->nC) {
        q->A.assign(A, A + (size_t)q->nC * q->nV);
        q->lA.assign(lA, lA + q->nC); q->uA.assign(uA, uA + q->nC);
    }
    refqp_update_bounds(h, l, u);
    return cold_init(q);
}

// per-cycle solve: QPOasesBackEnd::solve(). The caller must have pushed a fresh H via
// refqp_update_task (eps is added to the stored H here, exactly once per call, as the
// reference does on its _H copy).
int refqp_solve(void* h) {
    RefQP* q = (RefQP*)h;
    int nWSR = 13200;
    clamp_infty(q);
    add_eps(q);
    const double* A = q->nC ? q->A.data() : 0;
    const double* l = q->has_bounds ? q->l.data() : 0;
    const double* u = q->has_bounds ? q->u.data() : 0;
    const double* lA = q->nC ? q->lA.data() : 0;
    const double* uA = q->nC ? q->uA.data() : 0;
    qpOASES::returnValue val = q->prob->hotstart(q->H.data(), q->g.data(), A, l, u, lA, uA, nWSR, 0);
    q->last_nwsr = nWSR;
    if (val != qpOASES::SUCCESSFUL_RETURN) {
        q->n_fallback_warm++;
        nWSR = 13200;
        val = q->prob->init(q->H.data(), q->g.data(), A, l, u, lA, uA, nWSR, 0,
                            q->x.data(), q->y.data(), &q->bounds, &q->constraints);
        q->last_nwsr = nWSR;
        if (val != qpOASES::SUCCESSFUL_RETURN) {
            q->n_fallback_cold++;
            return cold_init(q);   // note: adds eps a second time, like the reference
        }
    }
    qpOASES::returnValue ok = q->prob->getPrimalSolution(q->x.data());
    q->prob->getDualSolution(q->y.data());
    q->prob->getBounds(q->bounds);
    q->prob->getConstraints(q->constraints);
    if (qpOASES::getSimpleStatus(ok) < 0) return cold_init(q);
    return 1;
}

void refqp_get_solution(void* h, double* x) {
    RefQP* q = (RefQP*)h;
    std::memcpy(x, q->x.data(), sizeof(double) * q->nV);
}
void refqp_get_dual(void* h, double* y) {
    RefQP* q = (RefQP*)h;
    std::memcpy(y, q->y.data(), sizeof(double) * (q->nV + q->nC));
}
double refqp_objective(void* h) { return ((RefQP*)h)->prob->getObjVal(); }

}  // extern "C"
