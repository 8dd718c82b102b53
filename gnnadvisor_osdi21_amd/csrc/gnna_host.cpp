// gnna_host.cpp -- host half of libgnna.so: error reporting, scheduling knobs and the
// neighbor-group partitioner that replaces build_part (reference:
// GNNAdvisor/GNNConv/GNNAdvisor.cpp:210-251).
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "gnna.h"
#include "gnna_internal.h"

namespace {

thread_local char t_error[512] = "";

// Defaults chosen by measurement on MI355X (DESIGN.md "Tuning"); override with
// gnna_set_tuning() or the GNNA_TUNE environment variable
// ("G=16,U=8,BPC=0,XCD=1,TRUST=0").
const gnna_tuning kDefaultTuning = {/*groups_per_chunk=*/16, /*loads_in_flight=*/8,
                                    /*blocks_per_cu=*/0, /*xcd_remap=*/1, /*trust_canonical=*/0};
gnna_tuning g_tuning = kDefaultTuning;
std::mutex g_tuning_mutex;
std::once_flag g_env_once;

void apply_env()
{
    const char *s = std::getenv("GNNA_TUNE");
    if (!s) return;
    char buf[256];
    std::strncpy(buf, s, sizeof(buf) - 1);
    buf[sizeof(buf) - 1] = 0;
    for (char *tok = std::strtok(buf, ",; "); tok; tok = std::strtok(nullptr, ",; ")) {
        char *eq = std::strchr(tok, '=');
        if (!eq) continue;
        *eq = 0;
        int v = std::atoi(eq + 1);
        if (!std::strcmp(tok, "G")) g_tuning.groups_per_chunk = v;
        else if (!std::strcmp(tok, "U")) g_tuning.loads_in_flight = v;
        else if (!std::strcmp(tok, "BPC")) g_tuning.blocks_per_cu = v;
        else if (!std::strcmp(tok, "XCD")) g_tuning.xcd_remap = v;
        else if (!std::strcmp(tok, "TRUST")) g_tuning.trust_canonical = v;
    }
}

}  // namespace

namespace gnna {
int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(t_error, sizeof(t_error), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace gnna

extern "C" {
#pragma GCC visibility push(default)

int gnna_version(void) { return GNNA_VERSION; }

const char *gnna_last_error(void) { return t_error; }

void gnna_set_tuning(const gnna_tuning *t)
{
    std::call_once(g_env_once, apply_env);
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    if (!t) { g_tuning = kDefaultTuning; return; }
    if (t->groups_per_chunk > 0) g_tuning.groups_per_chunk = t->groups_per_chunk;
    if (t->loads_in_flight > 0) g_tuning.loads_in_flight = t->loads_in_flight;
    if (t->blocks_per_cu >= 0) g_tuning.blocks_per_cu = t->blocks_per_cu;
    if (t->xcd_remap >= 0) g_tuning.xcd_remap = t->xcd_remap;
    if (t->trust_canonical >= 0) g_tuning.trust_canonical = t->trust_canonical;
}

void gnna_get_tuning(gnna_tuning *t)
{
    std::call_once(g_env_once, apply_env);
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    if (t) *t = g_tuning;
}

// Pass 1 of build_part (GNNAdvisor.cpp:219-227): P = sum_i ceil(deg_i / partSize).
int64_t gnna_count_parts(int partSize, const int32_t *indptr, int64_t num_nodes)
{
    if (partSize <= 0) return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "partSize must be positive (got %d)", partSize);
    if (num_nodes < 0 || (num_nodes > 0 && !indptr))
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "bad indptr / num_nodes");
    int64_t parts = 0;
    const int64_t ps = partSize;
    for (int64_t i = 0; i < num_nodes; i++) {
        const int64_t deg = (int64_t)indptr[i + 1] - (int64_t)indptr[i];
        if (deg < 0) return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "indptr decreases at row %lld", (long long)i);
        parts += (deg + ps - 1) / ps;
    }
    return parts;
}

// Pass 2 (GNNAdvisor.cpp:233-249).  Writes plain int32 (the reference stores the offsets
// in float32 tensors, which is inexact beyond 2^24 edges) and always closes the array
// with partPtr[P] = indptr[N] (the reference leaves it 0 when the last row is empty).
int gnna_build_part_i32(int partSize, const int32_t *indptr, int64_t num_nodes,
                        int32_t *partPtr, int32_t *part2Node, int64_t num_parts)
{
    const int64_t expect = gnna_count_parts(partSize, indptr, num_nodes);
    if (expect < 0) return (int)expect;
    if (expect != num_parts)
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "num_parts=%lld but the CSR has %lld groups at partSize=%d",
                          (long long)num_parts, (long long)expect, partSize);
    if (!partPtr || (num_parts > 0 && !part2Node))
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "null output pointer");
    int64_t p = 0;
    for (int64_t i = 0; i < num_nodes; i++) {
        const int32_t end = indptr[i + 1];
        for (int64_t beg = indptr[i]; beg < end; beg += partSize) {
            partPtr[p] = (int32_t)beg;
            part2Node[p++] = (int32_t)i;
        }
    }
    partPtr[p] = num_nodes > 0 ? indptr[num_nodes] : 0;
    return GNNA_OK;
}

#pragma GCC visibility pop
}  // extern "C"
