// asan_sparse_driver.cpp -- TEST INFRASTRUCTURE ONLY (tests/test_sparse_symbolic.py): runs the host-side symbolic
// analysis of the product (csrc/sparse_symbolic.h) and the C++ oracle of the multifrontal Cholesky over one matrix read
// from a file, at two panel budgets, so that the test can build it with -fsanitize=address,undefined.
// File layout: int32 m, n, nnz; int32 colptr[n+1]; int32 rowind[nnz]; double val[nnz].
#include <cstdio>
#include <vector>

#include "sparse_chol_oracle.cpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    int hdr[3];
    if (fread(hdr, sizeof(int), 3, f) != 3) return 4;
    const int m = hdr[0], n = hdr[1], nnz = hdr[2];
    std::vector<int> cp((size_t)n + 1), ri((size_t)nnz);
    std::vector<double> cv((size_t)nnz);
    if (fread(cp.data(), sizeof(int), cp.size(), f) != cp.size()) return 4;
    if (fread(ri.data(), sizeof(int), ri.size(), f) != ri.size()) return 4;
    if (fread(cv.data(), sizeof(double), cv.size(), f) != cv.size()) return 4;
    fclose(f);
    std::vector<double> d((size_t)n), rhs((size_t)m), z((size_t)m), stats(8);
    for (int j = 0; j < n; ++j) d[j] = 0.5 + (double)((j * 2654435761u) % 1000u) / 500.0;
    for (int i = 0; i < m; ++i) rhs[i] = 1.0 + (double)(i % 7);
    std::vector<int> perm((size_t)m), parent((size_t)m), cnt((size_t)m);
    const int budgets[2][2] = {{32, 7680}, {4, 96}};
    for (int t = 0; t < 2; ++t) {
        int nfix = 0;
        int rc = spchol_oracle(m, n, cp.data(), ri.data(), cv.data(), d.data(), rhs.data(), 1e-30, 1e64, 0.0, budgets[t][0],
                               budgets[t][1], perm.data(), nullptr, z.data(), &nfix, stats.data());
        // rc 3 with the tiny budget: a front does not fit 96 doubles even as a one-column panel -- a refusal, not a fault
        if (rc != 0 && !(t == 1 && rc == 3)) { printf("spchol_oracle rc %d (budget %d)\n", rc, t); return 10 + rc; }
        rc = spsym_structures(m, n, cp.data(), ri.data(), budgets[t][0], budgets[t][1], perm.data(), parent.data(), cnt.data());
        if (rc != 0 && !(t == 1 && rc == 3)) { printf("spsym_structures rc %d (budget %d)\n", rc, t); return 20 + rc; }
    }
    printf("ok %d x %d panels %.0f height %.0f\n", m, n, stats[0], stats[1]);
    return 0;
}
