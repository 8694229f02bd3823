// tests/adapter_mock/run_adapter.cpp -- TEST HARNESS: loads the compiled adapter the way OpenSoT's factory does
// (dlopen + "create_instance", src/solvers/BackEndFactory.cpp:4-17) and drives it in iHQP's call order
// (updateTask -> updateConstraints -> updateBounds -> solve -> getSolution, src/solvers/iHQP.cpp:279, 335-349) on the reference's
// robot-free known-answer problems (tests/solvers/TestQPOases.cpp:208-254, 346-412).  Prints one JSON line.
#include <OpenSoT/solvers/BackEnd.h>
#include <dlfcn.h>
#include <cstdio>
using OpenSoT::solvers::BackEnd;
typedef BackEnd* (*create_t)(const int, const int, OpenSoT::HessianType, const double);
typedef void (*destroy_t)(BackEnd*);

static Eigen::MatrixXd mat(int r, int c, const double* v) { Eigen::MatrixXd M(r, c); for (int i = 0; i < r; ++i) for (int j = 0; j < c; ++j) M(i, j) = v[i * c + j]; return M; }
static Eigen::VectorXd vec(int n, const double* v) { Eigen::VectorXd x(n); for (int i = 0; i < n; ++i) x(i) = v[i]; return x; }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { std::fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    create_t create = (create_t)dlsym(h, "create_instance");
    destroy_t destroy = (destroy_t)dlsym(h, "destroy_instance");
    if (!create || !destroy) return 4;
    // TestQPOases.cpp:208-254: [1 1 1] x = 10 in the box +-10 -> (3.333, 3.333, 3.333); then the row x0 + x2 = 20 -> (10, -10, 10)
    BackEnd* qp = create(3, 1, OpenSoT::HST_SEMIDEF, 1e4);
    const double H3[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1}, g3[3] = {-10, -10, -10}, A0[3] = {0, 0, 0}, z1[1] = {0}, lo[3] = {-10, -10, -10}, up[3] = {10, 10, 10};
    bool ok1 = qp->initProblem(mat(3, 3, H3), vec(3, g3), mat(1, 3, A0), vec(1, z1), vec(1, z1), vec(3, lo), vec(3, up));
    ok1 = ok1 && qp->solve();
    double s1[3]; for (int i = 0; i < 3; ++i) s1[i] = qp->getSolution()(i);
    const double A1[3] = {1, 0, 1}, b20[1] = {20};
    bool ok2 = qp->updateConstraints(mat(1, 3, A1), vec(1, b20), vec(1, b20)) && qp->solve();
    double s2[3]; for (int i = 0; i < 3; ++i) s2[i] = qp->getSolution()(i);
    const double eps = qp->getEpsRegularisation();
    destroy(qp);
    // TestQPOases.cpp:346-412: H = I, g = (-5, 5) -> x = -g; then g = (-1, 1) through updateTask
    qp = create(2, 2, OpenSoT::HST_IDENTITY, 1e-9);
    const double I2[4] = {1, 0, 0, 1}, g2[2] = {-5, 5}, Z2[4] = {0, 0, 0, 0}, m10[2] = {-10, -10}, p10[2] = {10, 10};
    bool ok3 = qp->initProblem(mat(2, 2, I2), vec(2, g2), mat(2, 2, Z2), vec(2, m10), vec(2, p10), vec(2, m10), vec(2, p10)) && qp->solve();
    double s3[2] = {qp->getSolution()(0), qp->getSolution()(1)};
    const double g2b[2] = {-1, 1};
    bool ok4 = qp->updateTask(mat(2, 2, I2), vec(2, g2b)) && qp->solve();
    double s4[2] = {qp->getSolution()(0), qp->getSolution()(1)};
    const double f4 = qp->getObjective();
    delete qp;      // (the factory's shared_ptr destroys through the virtual destructor: BackEndFactory.cpp:9-10)
    std::printf("{\"ok\": [%d, %d, %d, %d], \"s1\": [%.17g, %.17g, %.17g], \"s2\": [%.17g, %.17g, %.17g], \"s3\": [%.17g, %.17g], "
                "\"s4\": [%.17g, %.17g], \"f4\": %.17g, \"eps\": %.17g}\n",
                (int)ok1, (int)ok2, (int)ok3, (int)ok4, s1[0], s1[1], s1[2], s2[0], s2[1], s2[2], s3[0], s3[1], s4[0], s4[1], f4, eps);
    return 0;
}
