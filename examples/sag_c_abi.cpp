// sag_c_abi.cpp -- a host that is not Python and not PyTorch: it owns its device memory through the HIP
// runtime, builds the neighbor-group partition with the C ABI (include/gnna.h), runs the three
// aggregation entry points and checks them against a plain CPU loop.
//
//   hipcc -O2 -I include examples/sag_c_abi.cpp -L gnnadvisor_osdi21_amd/csrc -lgnna \
//         -Wl,-rpath,$PWD/gnnadvisor_osdi21_amd/csrc -o /tmp/sag_c_abi && /tmp/sag_c_abi
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gnna.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define GNNA_OK_OR_DIE(x) do { int rc_ = (x); if (rc_ != GNNA_OK) { std::printf("libgnna error %d: %s (%s:%d)\n", rc_, gnna_last_error(), __FILE__, __LINE__); return 3; } } while (0)

template <typename T>
static T *to_device(const std::vector<T> &v)
{
    T *d = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d), v.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
    if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main()
{
    const int64_t n = 5000;
    const int dim = 41, part_size = 8;
    // a seeded edge list; the C ABI builds the CSR (duplicates merged, columns sorted) and the degree norms
    std::vector<int32_t> src, dst;
    uint64_t s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
    for (int64_t e = 0; e < 60000; e++) {
        const int32_t u = (int32_t)(rnd() % n), v = (int32_t)(rnd() % (e % 7 == 0 ? 16 : n));  // a few hub columns
        src.push_back(u); dst.push_back(v); src.push_back(v); dst.push_back(u);
    }
    std::vector<int32_t> rp(n + 1), ci(src.size());
    const int64_t nnz = gnna_csr_from_edges_i32(src.data(), dst.data(), (int64_t)src.size(), n, rp.data(), ci.data());
    if (nnz < 0) { std::printf("csr_from_edges: %s\n", gnna_last_error()); return 3; }
    ci.resize(nnz);
    std::vector<float> deg(n);
    GNNA_OK_OR_DIE(gnna_degrees_f32(rp.data(), n, deg.data()));
    const int64_t P = gnna_count_parts(part_size, rp.data(), n);
    std::vector<int32_t> pp(P + 1), p2n(P);
    GNNA_OK_OR_DIE(gnna_build_part_i32(part_size, rp.data(), n, pp.data(), p2n.data(), P));

    std::vector<float> X((size_t)n * dim);
    for (auto &x : X) x = (float)((int)(rnd() % 2001) - 1000) / 1000.f;

    float *dX = to_device(X), *dDeg = to_device(deg), *dY = nullptr;
    int32_t *dRp = to_device(rp), *dCi = to_device(ci), *dPp = to_device(pp), *dP2n = to_device(p2n);
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&dY), X.size() * sizeof(float)));
    if (!dX || !dDeg || !dRp || !dCi || !dPp || !dP2n) { std::printf("allocation failed\n"); return 2; }
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    std::vector<float> Y(X.size());
    double worst = 0.0;
    for (int mode = 0; mode < 3; mode++) {
        const float eps = 0.5f;
        if (mode == 0)
            GNNA_OK_OR_DIE(gnna_sag_f32(dX, dRp, dCi, dDeg, dPp, dP2n, dY, n, dim, P, part_size, 32, 4, stream));
        else if (mode == 1)
            GNNA_OK_OR_DIE(gnna_agg_gcn_f32(dX, dRp, dCi, dDeg, dPp, dP2n, dY, n, dim, P, part_size, 32, 4, stream));
        else
            GNNA_OK_OR_DIE(gnna_agg_gin_f32(dX, dRp, dCi, eps, dPp, dP2n, dY, n, dim, P, part_size, 32, 4, stream));
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipMemcpy(Y.data(), dY, Y.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; i++) {
            for (int d = 0; d < dim; d++) {
                double ref = 0.0, scale = 0.0;
                for (int32_t e = rp[i]; e < rp[i + 1]; e++) {
                    const double c = mode == 1 ? (double)deg[i] * (double)deg[ci[e]] : (mode == 2 ? (double)eps : 1.0);
                    ref += c * X[(size_t)ci[e] * dim + d];
                    scale += std::fabs(c * X[(size_t)ci[e] * dim + d]);
                }
                const double err = std::fabs((double)Y[(size_t)i * dim + d] - ref) / (scale > 1.0 ? scale : 1.0);
                if (err > worst) worst = err;
            }
        }
    }
    {
        // the general entry (0.4.0): rows written with the leading dimension the library prefers for this gather, the
        // result clamped at zero in the same call -- relu(A X) into a buffer whose rows are 2 * dim floats apart
        const int64_t ld_in = gnna_preferred_ld(dim, n, nnz), ld_out = 2 * dim;
        std::vector<float> Xl((size_t)n * ld_in, -7.f), Yl((size_t)n * ld_out, -9.f);
        for (int64_t i = 0; i < n; i++)
            for (int d = 0; d < dim; d++) Xl[(size_t)i * ld_in + d] = X[(size_t)i * dim + d];
        float *dXl = to_device(Xl), *dYl = to_device(Yl);
        GNNA_OK_OR_DIE(gnna_agg_ld_f32(0, dXl, ld_in, n, dCi, nullptr, nullptr, 1.f, dPp, dP2n, dYl, ld_out, n, dim, P, part_size,
                                       GNNA_EPILOGUE_RELU, stream));
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipMemcpy(Yl.data(), dYl, Yl.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; i++) {
            for (int d = 0; d < dim; d++) {
                double ref = 0.0, scale = 0.0;
                for (int32_t e = rp[i]; e < rp[i + 1]; e++) {
                    ref += X[(size_t)ci[e] * dim + d];
                    scale += std::fabs(X[(size_t)ci[e] * dim + d]);
                }
                ref = ref > 0.0 ? ref : 0.0;
                const double err = std::fabs((double)Yl[(size_t)i * ld_out + d] - ref) / (scale > 1.0 ? scale : 1.0);
                if (err > worst) worst = err;
            }
            for (int d = dim; d < ld_out; d++)
                if (Yl[(size_t)i * ld_out + d] != -9.f) { std::printf("gnna_agg_ld_f32 wrote between the output rows\n"); return 1; }
        }
    }
    std::printf("libgnna %d: n=%lld nnz=%lld groups=%lld dim=%d  max err / scale = %.3e\n", gnna_version(),
                (long long)n, (long long)nnz, (long long)P, dim, worst);
    if (!(worst <= 1e-4)) { std::printf("MISMATCH\n"); return 1; }
    std::printf("OK\n");
    return 0;
}
