/*
 * gnna_oracle.c -- CPU restatement of GNNAdvisor's neighbor-group aggregation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (gnnadvisor_osdi21_amd/) never links, imports or calls it.
 *
 * Each function restates one piece of the reference (paths relative to
 * /root/reference/GNNAdvisor/GNNConv/):
 *
 *   oracle_count_parts / oracle_build_part_ref   GNNAdvisor.cpp:210-251 (build_part),
 *                                                bug-for-bug incl. float32 storage and
 *                                                the missing-sentinel case
 *   oracle_build_part                            same partition, int32 storage, sentinel
 *                                                always written (the product's contract)
 *   oracle_sag_groups                            GNNAdvisor_kernel.cu:186-259 (SAG kernel)
 *   oracle_gcn_groups                            GNNAdvisor_kernel.cu:324-415 / 478-552
 *   oracle_gin_groups                            GNNAdvisor_kernel.cu:620-689 / 749-814
 *   oracle_csr_*_f64                             independent fp64 CSR formulas (the
 *                                                "reference CPU SpMM" of unitest.py:33-40)
 *   oracle_csr_sag_omp                           row-parallel fp32 CSR SpMM used as the
 *                                                multi-core CPU baseline in bench.py
 *
 * Pinning: oracle_build_part_ref is checked bit-for-bit against outputs of the
 * reference's own build_part (compiled from /root/reference by oracle/build_ref.py into
 * oracle/_ref/, and committed as tests/golden/build_part_*.json); the aggregation
 * functions are checked against the reference's own known-answer test (X = ones =>
 * row-nnz counts, unitest.py:27,54-63) and the worked example of SURVEY appendix A.
 *
 * Compile with -ffp-contract=off: the reference rounds coefficient*feature and the
 * accumulate separately (__fmaf_rn(a, b, 0) followed by +=, .cu:355,389,405).
 */
#include <omp.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* ------------------------------------------------------------------ build_part */

/* GNNAdvisor.cpp:219-227 -- pass 1: numParts = sum_i ceil(deg_i / partSize). */
int64_t oracle_count_parts(int partSize, const int32_t *indptr, int64_t num_nodes)
{
    int64_t numParts = 0;
    for (int64_t i = 0; i < num_nodes; i++) {
        int degree = indptr[i + 1] - indptr[i];
        int thisNumParts = (degree % partSize == 0) ? degree / partSize : degree / partSize + 1;
        numParts += thisNumParts;
    }
    return numParts;
}

/* GNNAdvisor.cpp:229-249 -- pass 2, bug-for-bug.  Outputs are float32 (torch::zeros
 * default dtype, .cpp:229-230), pre-zeroed by the caller like torch::zeros does.
 * The closing sentinel partPtr[P] is only written when the LAST node has a part whose
 * end equals indptr[N] (.cpp:246-247); otherwise it stays 0. */
void oracle_build_part_ref(int partSize, const int32_t *indptr, int64_t num_nodes,
                           float *partPtr /* [P+1], zeroed */, float *part2Node /* [P], zeroed */)
{
    int64_t part_counter = 0;
    for (int64_t i = 0; i < num_nodes; i++) {
        int degree = indptr[i + 1] - indptr[i];
        int thisNumParts = (degree % partSize == 0) ? degree / partSize : degree / partSize + 1;
        for (int pid = 0; pid < thisNumParts; pid++) {
            int partBeg = indptr[i] + pid * partSize;
            int partEnd = partBeg + partSize < indptr[i + 1] ? partBeg + partSize : indptr[i + 1];
            partPtr[part_counter] = (float)partBeg;
            part2Node[part_counter++] = (float)i;
            if (i == num_nodes - 1 && partEnd == indptr[i + 1])
                partPtr[part_counter] = (float)partEnd;
        }
    }
}

/* Same partition with the product's contract: exact int32 storage, sentinel
 * partPtr[P] = indptr[N] always.  Differs from oracle_build_part_ref only (a) in the
 * sentinel when the last node has degree 0 and (b) where float32 cannot hold the
 * offset (> 2^24).  Both divergences are documented in DESIGN.md. */
void oracle_build_part(int partSize, const int32_t *indptr, int64_t num_nodes,
                       int32_t *partPtr /* [P+1] */, int32_t *part2Node /* [P] */)
{
    int64_t p = 0;
    for (int64_t i = 0; i < num_nodes; i++) {
        for (int beg = indptr[i]; beg < indptr[i + 1]; beg += partSize) {
            partPtr[p] = beg;
            part2Node[p++] = (int32_t)i;
        }
    }
    partPtr[p] = indptr[num_nodes];
}

/* ------------------------------------------------------------------ aggregation */

/* GNNAdvisor_kernel.cu:186-259.  One "warp" per neighbor-group: a float32 partial row,
 * zeroed at the first neighbor, += input[nid][d] per neighbor in CSR order (.cu:231-250),
 * then added into output[srcId] (.cu:253-257; atomic order is unspecified on the GPU,
 * here groups are flushed in index order).  output must be zeroed by the caller
 * (zeros_like, .cu:121).  An empty group adds nothing (the reference would flush
 * uninitialised shared memory for it -- never produced by a correct partition). */
void oracle_sag_groups(const float *input, const int32_t *column_index,
                       const int32_t *part_pointers, const int32_t *part2Node,
                       int64_t num_parts, int dim, float *output, float *scratch /* [dim] */)
{
    for (int64_t w = 0; w < num_parts; w++) {
        int srcId = part2Node[w];
        int partBeg = part_pointers[w], partEnd = part_pointers[w + 1];
        if (partEnd <= partBeg) continue;
        for (int d = 0; d < dim; d++) scratch[d] = 0.0f;
        for (int n = partBeg; n < partEnd; n++) {
            const float *row = input + (size_t)column_index[n] * dim;
            for (int d = 0; d < dim; d++) scratch[d] += row[d];
        }
        float *out = output + (size_t)srcId * dim;
        for (int d = 0; d < dim; d++) out[d] += scratch[d];
    }
}

/* GNNAdvisor_kernel.cu:324-415 (forward) and 478-552 (backward): identical arithmetic.
 * coef = round(degrees[src] * degrees[nid])       (.cu:389  __fmaf_rn(a, b, 0))
 * partial[d] += round(coef * input[nid][d])       (.cu:405) */
void oracle_gcn_groups(const float *input, const int32_t *column_index, const float *degrees,
                       const int32_t *part_pointers, const int32_t *part2Node,
                       int64_t num_parts, int dim, float *output, float *scratch)
{
    for (int64_t w = 0; w < num_parts; w++) {
        int srcId = part2Node[w];
        int partBeg = part_pointers[w], partEnd = part_pointers[w + 1];
        if (partEnd <= partBeg) continue;
        float src_norm = degrees[srcId];
        for (int d = 0; d < dim; d++) scratch[d] = 0.0f;
        for (int n = partBeg; n < partEnd; n++) {
            int nid = column_index[n];
            float coef = src_norm * degrees[nid];
            const float *row = input + (size_t)nid * dim;
            for (int d = 0; d < dim; d++) {
                float t = coef * row[d];
                scratch[d] += t;
            }
        }
        float *out = output + (size_t)srcId * dim;
        for (int d = 0; d < dim; d++) out[d] += scratch[d];
    }
}

/* GNNAdvisor_kernel.cu:620-689 (forward) and 749-814 (backward): unweighted partial,
 * scaled by epsilon at the flush (.cu:686, 811). */
void oracle_gin_groups(const float *input, const int32_t *column_index, float epsilon,
                       const int32_t *part_pointers, const int32_t *part2Node,
                       int64_t num_parts, int dim, float *output, float *scratch)
{
    for (int64_t w = 0; w < num_parts; w++) {
        int srcId = part2Node[w];
        int partBeg = part_pointers[w], partEnd = part_pointers[w + 1];
        if (partEnd <= partBeg) continue;
        for (int d = 0; d < dim; d++) scratch[d] = 0.0f;
        for (int n = partBeg; n < partEnd; n++) {
            const float *row = input + (size_t)column_index[n] * dim;
            for (int d = 0; d < dim; d++) scratch[d] += row[d];
        }
        float *out = output + (size_t)srcId * dim;
        for (int d = 0; d < dim; d++) {
            float t = epsilon * scratch[d];
            out[d] += t;
        }
    }
}

/* ------------------------------------------------------------------ independent fp64 CSR formulas */

/* mode 0: Y = A X ; mode 1: Y_i = sum_j deg_i*deg_j*X_j ; mode 2: Y = eps * A X.
 * (the CPU-side reference of unitest.py:33-40 is index_add over the edge list, which
 * for a deduplicated CSR is exactly mode 0.) */
void oracle_csr_f64(int mode, const float *input, const int32_t *row_pointers,
                    const int32_t *column_index, const float *degrees, double epsilon,
                    int64_t num_nodes, int dim, double *output)
{
    for (int64_t i = 0; i < num_nodes; i++) {
        double *out = output + (size_t)i * dim;
        for (int d = 0; d < dim; d++) out[d] = 0.0;
        for (int n = row_pointers[i]; n < row_pointers[i + 1]; n++) {
            int nid = column_index[n];
            double c = 1.0;
            if (mode == 1) c = (double)degrees[i] * (double)degrees[nid];
            const float *row = input + (size_t)nid * dim;
            for (int d = 0; d < dim; d++) out[d] += c * (double)row[d];
        }
        if (mode == 2)
            for (int d = 0; d < dim; d++) out[d] *= epsilon;
    }
}

/* Row-parallel fp32 CSR SpMM (Y = A X) -- the multi-core CPU baseline timed by
 * bench.py (cpu_baseline.kind = "port").  rows [row_beg, row_end).  Written the way a CPU
 * SpMM would be: 64-float register-blocked accumulator per output row (the compiler keeps it
 * in vector registers), next source row prefetched, dynamic row scheduling for skewed degrees.
 * Per output element the neighbours are still summed in CSR order, so results equal the
 * plain loop bit for bit. */
__attribute__((optimize("O3"), target("avx2")))
void oracle_csr_sag_omp(const float *restrict input, const int32_t *restrict row_pointers,
                        const int32_t *restrict column_index, int64_t row_beg, int64_t row_end,
                        int dim, float *restrict output)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = row_beg; i < row_end; i++) {
        const int32_t beg = row_pointers[i], end = row_pointers[i + 1];
        for (int d0 = 0; d0 < dim; d0 += 64) {
            const int w = dim - d0 < 64 ? dim - d0 : 64;
            float acc[64];
            for (int d = 0; d < 64; d++) acc[d] = 0.0f;
            if (w == 64) {
                for (int32_t n = beg; n < end; n++) {
                    const float *restrict row = input + (size_t)column_index[n] * dim + d0;
                    if (n + 4 < end) __builtin_prefetch(input + (size_t)column_index[n + 4] * dim + d0);
                    for (int d = 0; d < 64; d++) acc[d] += row[d];
                }
            } else {
                for (int32_t n = beg; n < end; n++) {
                    const float *restrict row = input + (size_t)column_index[n] * dim + d0;
                    for (int d = 0; d < w; d++) acc[d] += row[d];
                }
            }
            float *restrict out = output + (size_t)i * dim + d0;
            for (int d = 0; d < w; d++) out[d] = acc[d];
        }
    }
}

/* First-touch copy for the timed CPU baseline: `dst` is a fresh (never written) allocation; copying
 * it with the same static thread layout that later reads it spreads its pages over the NUMA nodes of
 * the threads instead of leaving them all on the node of the one thread that filled the array. */
void oracle_first_touch_copy(float *restrict dst, const float *restrict src, int64_t n)
{
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (n + 1023) / 1024; b++) {
        const int64_t lo = b * 1024, hi = lo + 1024 < n ? lo + 1024 : n;
        if (src) for (int64_t i = lo; i < hi; i++) dst[i] = src[i];
        else for (int64_t i = lo; i < hi; i++) dst[i] = 0.0f;
    }
}

int oracle_num_threads(void)
{
    int n = 1;
#pragma omp parallel
    {
#pragma omp master
        n = omp_get_num_threads();
    }
    return n;
}

/* Single-thread neighbor-group SAG over a slice of groups [g_beg, g_end): the scalar
 * port of the reference algorithm, timed as cpu_baseline (cores = 1). */
void oracle_sag_groups_slice(const float *input, const int32_t *column_index,
                             const int32_t *part_pointers, const int32_t *part2Node,
                             int64_t g_beg, int64_t g_end, int dim, float *output, float *scratch)
{
    oracle_sag_groups(input, column_index, part_pointers + g_beg, part2Node + g_beg,
                      g_end - g_beg, dim, output, scratch);
}
