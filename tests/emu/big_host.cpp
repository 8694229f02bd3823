// tests/emu/big_host.cpp -- TEST INFRASTRUCTURE: the wide-QP solver of opensot_amd/csrc/osot_qp_big.h compiled for the host with a
// team of one thread (every parallel section becomes a plain loop), so that its algorithm can be checked against the oracle where no
// GPU is present.  Not part of the product: libosot_mi355x.so runs the same source as a 256-thread workgroup.
#define OSOT_BIG_HOST 1
#include <vector>
#include "osot_qp_big.h"

namespace {
struct TeamHost { int tid = 0, nt = 1; void sync() const {} };
}

extern "C" __attribute__((visibility("default")))
int osot_big_host_solve(int n, int nc, const double* H, const double* g, const double* A, const double* lA, const double* uA,
                        const double* l, const double* u, double eps_abs, int max_iter, double* x, int* status, int* iters) {
    using namespace osot::big;
    if (n < 1 || n > kMaxVars || nc < 0 || nc > kMaxRows) return -1;
    std::vector<double> work(2 * (size_t)n * n);
    std::vector<char> sh(shared_bytes(n, nc) + 16);
    Args a;
    a.n = n; a.nc = nc; a.max_iter = max_iter > 0 ? max_iter : 20 * (n + nc) + 100; a.eps = eps_abs;
    a.H = H; a.g = g; a.A = A; a.lA = lA; a.uA = uA; a.l = l; a.u = u;
    a.x = x; a.status = status; a.iters = iters;
    a.Lw = work.data(); a.J = work.data() + (size_t)n * n;
    const Shared s = carve(sh.data(), n, nc);
    solve(TeamHost{}, a, s);
    return 0;
}
