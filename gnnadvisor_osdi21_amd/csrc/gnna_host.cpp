// gnna_host.cpp -- host half of libgnna.so: error reporting, scheduling knobs and the
// neighbor-group partitioner that replaces build_part (reference:
// GNNAdvisor/GNNConv/GNNAdvisor.cpp:210-251).
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cmath>
#include <mutex>
#include <numeric>
#include <sched.h>
#include <string>
#include <thread>
#include <vector>

#include "gnna.h"
#include "gnna_internal.h"

namespace {

thread_local char t_error[512] = "";

// Defaults chosen by measurement on MI355X (DESIGN.md "Tuning"); override with
// gnna_set_tuning() or the GNNA_TUNE environment variable
// ("G=16,U=4,BPC=0,XCD=1,TRUST=0,PHASES=0").
const gnna_tuning kDefaultTuning = {/*struct_size=*/(int)sizeof(gnna_tuning), /*groups_per_chunk=*/16, /*loads_in_flight=*/4,
                                    /*blocks_per_cu=*/0, /*xcd_remap=*/1, /*trust_canonical=*/0,
                                    /*column_phases=*/0, /*avg_degree=*/0, /*nonlocal_ids=*/0,
                                    /*gcn_prescale=*/0, /*pad_rows=*/0, /*zero_fill=*/0,
                                    /*sweep=*/0, /*sweep_slack=*/0, /*deterministic=*/0, /*pack_ids=*/0,
                                    /*ids_check_every=*/64, /*wide_blocks=*/0};
gnna_tuning g_tuning = kDefaultTuning;
std::mutex g_tuning_mutex;
std::once_flag g_env_once;

// per-graph hints (gnna_set_graph_hints): a small table keyed by the graph's column_index pointer
struct GraphHint {
    const void *key = nullptr;
    int avg_degree = 0;
    int nonlocal_ids = 0;
    uint64_t stamp = 0;
    // measured schedules (gnna_set_graph_phases): column phases for up to 8 feature widths
    int sched_dim[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int sched_phases[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
constexpr int kMaxGraphHints = 64;
GraphHint g_hints[kMaxGraphHints];
uint64_t g_hint_clock = 0;

void apply_env()
{
    if (const char *c = std::getenv("GNNA_DEBUG_FULL_CHECKSUM"))
        if (std::atoi(c) != 0) g_tuning.ids_check_every = 1;
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
        else if (!std::strcmp(tok, "PHASES")) g_tuning.column_phases = v;
        else if (!std::strcmp(tok, "AVGDEG")) g_tuning.avg_degree = v;
        else if (!std::strcmp(tok, "NONLOCAL")) g_tuning.nonlocal_ids = v;
        else if (!std::strcmp(tok, "PRESCALE")) g_tuning.gcn_prescale = v;
        else if (!std::strcmp(tok, "PAD")) g_tuning.pad_rows = v;
        else if (!std::strcmp(tok, "ZERO")) g_tuning.zero_fill = v;
        else if (!std::strcmp(tok, "SWEEP")) g_tuning.sweep = v;
        else if (!std::strcmp(tok, "CHECK")) g_tuning.ids_check_every = v;
        else if (!std::strcmp(tok, "SLACK")) g_tuning.sweep_slack = v;
        else if (!std::strcmp(tok, "DET")) g_tuning.deterministic = v;
        else if (!std::strcmp(tok, "PACK")) g_tuning.pack_ids = v;
        else if (!std::strcmp(tok, "BLOCKS")) g_tuning.wide_blocks = v;
    }
}

}  // namespace

namespace gnna {
void apply_graph_hints(const void *column_index, int dim, gnna_tuning *tune)
{
    if (!column_index) return;
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    for (auto &h : g_hints) {
        if (h.key == column_index) {
            if (h.avg_degree > 0) {
                tune->avg_degree = h.avg_degree;
                tune->nonlocal_ids = h.nonlocal_ids;
            }
            if (tune->column_phases == 0) {  // an explicit process-wide setting wins over a measured schedule
                for (int i = 0; i < 8; i++)
                    if (h.sched_dim[i] == dim && dim > 0) tune->column_phases = h.sched_phases[i];
            }
            h.stamp = ++g_hint_clock;
            return;
        }
    }
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(t_error, sizeof(t_error), fmt, ap);
    va_end(ap);
    return code;
}

// quota of one cgroup v2 directory ("<quota|max> <period>" in cpu.max) in CPUs; 0 = no limit / unreadable
static double cgroup2_quota(const std::string &dir)
{
    double quota = 0.0;
    if (FILE *f = std::fopen((dir + "/cpu.max").c_str(), "r")) {
        char q[64] = {0};
        double period = 0.0;
        if (std::fscanf(f, "%63s %lf", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) quota = std::atof(q) / period;
        std::fclose(f);
    }
    return quota;
}

// CPUs the host passes may use: the smaller of the CPUs the calling thread is allowed to run on (sched_getaffinity: cpusets,
// taskset) and the tightest CPU quota on the way from the process's own cgroup (/proc/self/cgroup) up to the root -- the root alone
// is what a process inside a cgroup namespace sees, the nested path what one outside of it does.  A quota below one CPU means one thread.
static int cgroup_cpu_quota()      // tightest CPU quota on the way up from this process's cgroup, rounded down (>= 1); 0 = none
{
    double quota = 0.0;
    auto tighten = [&](double q) { if (q > 0 && (quota == 0.0 || q < quota)) quota = q; };
    std::string own;                                                           // cgroup v2 line: "0::/path"
    if (FILE *f = std::fopen("/proc/self/cgroup", "r")) {
        char line[1024];
        while (std::fgets(line, sizeof(line), f))
            if (std::strncmp(line, "0::", 3) == 0) { own = line + 3; while (!own.empty() && (own.back() == '\n' || own.back() == '/')) own.pop_back(); }
        std::fclose(f);
    }
    tighten(cgroup2_quota("/sys/fs/cgroup"));
    for (std::string path = own; !path.empty() && path[0] == '/'; path.erase(path.find_last_of('/')))
        tighten(cgroup2_quota("/sys/fs/cgroup" + path));
    if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
        double q = 0.0, period = 0.0;
        if (std::fscanf(fq, "%lf", &q) != 1) q = 0.0;
        std::fclose(fq);
        if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(fp, "%lf", &period) != 1) period = 0.0;
            std::fclose(fp);
        }
        if (q > 0 && period > 0) tighten(q / period);
    }
    return quota > 0.0 ? std::max(1, (int)quota) : 0;
}

int host_thread_budget(int cap)
{
    static const int quota = cgroup_cpu_quota();
    unsigned hw = std::thread::hardware_concurrency();
    int n = (int)(hw ? hw : 1u);
    // the CALLING thread's affinity, at every call: the threads spawned below inherit it (a process started under taskset; a
    // main thread an OpenMP runtime bound to one core -- sixteen workers there are slower than two)
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min(n, c); }
    if (quota > 0) n = std::min(n, quota);
    if (const char *e = std::getenv("GNNA_HOST_THREADS")) { const int v = std::atoi(e); if (v > 0) n = v; }   // read every time
    return std::max(1, std::min(n, cap));
}
}  // namespace gnna

namespace {
template <typename F>
void parallel_rows(int64_t n, F &&fn)
{
    int64_t nt = std::max<int64_t>(1, std::min<int64_t>(gnna::host_thread_budget(64), n / 4096 + 1));
    if (nt == 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    const int64_t step = (n + nt - 1) / nt;
    for (int64_t t = 0; t < nt; t++) {
        const int64_t lo = t * step, hi = std::min(n, lo + step);
        if (lo >= hi) break;
        th.emplace_back([&fn, lo, hi] { fn(lo, hi); });
    }
    for (auto &t : th) t.join();
}
}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int gnna_version(void) { return GNNA_VERSION; }

#ifndef GNNA_SOURCE_HASH
#define GNNA_SOURCE_HASH "unhashed"        /* (built by hand, not by gnnadvisor_osdi21_amd/build.py) */
#endif
const char *gnna_build_id(void) { return "0.6.0+" GNNA_SOURCE_HASH; }

const char *gnna_last_error(void) { return t_error; }

int gnna_host_threads(void) { return gnna::host_thread_budget(64); }

int gnna_set_tuning(const gnna_tuning *t)
{
    std::call_once(g_env_once, apply_env);
    if (t && t->struct_size != (int)sizeof(gnna_tuning))
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "gnna_set_tuning: struct_size %d, this library's gnna_tuning has %d bytes "
                          "(caller built against another gnna.h)", t->struct_size, (int)sizeof(gnna_tuning));
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    if (!t) { g_tuning = kDefaultTuning; apply_env(); return GNNA_OK; }  // defaults = built-in values + GNNA_TUNE
    if (t->groups_per_chunk > 0) g_tuning.groups_per_chunk = t->groups_per_chunk;
    if (t->loads_in_flight > 0) g_tuning.loads_in_flight = t->loads_in_flight;
    if (t->blocks_per_cu >= 0) g_tuning.blocks_per_cu = t->blocks_per_cu;
    if (t->xcd_remap >= 0) g_tuning.xcd_remap = t->xcd_remap;
    if (t->trust_canonical >= 0) g_tuning.trust_canonical = t->trust_canonical;
    if (t->column_phases >= 0) g_tuning.column_phases = t->column_phases;
    if (t->avg_degree >= 0) g_tuning.avg_degree = t->avg_degree;
    if (t->nonlocal_ids >= 0) g_tuning.nonlocal_ids = t->nonlocal_ids;
    if (t->gcn_prescale >= 0) g_tuning.gcn_prescale = t->gcn_prescale;
    if (t->pad_rows >= 0) g_tuning.pad_rows = t->pad_rows;
    if (t->zero_fill >= 0) g_tuning.zero_fill = t->zero_fill;
    if (t->sweep >= 0) g_tuning.sweep = t->sweep;
    if (t->sweep_slack >= 0) g_tuning.sweep_slack = t->sweep_slack;
    if (t->deterministic >= 0) g_tuning.deterministic = t->deterministic;
    if (t->pack_ids >= 0) g_tuning.pack_ids = t->pack_ids;
    if (t->ids_check_every > 0) g_tuning.ids_check_every = t->ids_check_every;
    if (t->wide_blocks >= 0) g_tuning.wide_blocks = t->wide_blocks;
    return GNNA_OK;
}

int gnna_set_graph_hints(const int32_t *column_index, int avg_degree, int nonlocal_ids)
{
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    if (!column_index) {  // forget everything
        for (auto &h : g_hints) h = GraphHint();
        return GNNA_OK;
    }
    GraphHint *slot = nullptr;
    for (auto &h : g_hints)
        if (h.key == column_index) slot = &h;
    if (avg_degree <= 0) {  // forget this graph
        if (slot) *slot = GraphHint();
        return GNNA_OK;
    }
    if (!slot) {  // free entry, else the least recently used one
        slot = &g_hints[0];
        for (auto &h : g_hints) {
            if (!h.key) { slot = &h; break; }
            if (h.stamp < slot->stamp) slot = &h;
        }
    }
    if (slot->key != column_index) *slot = GraphHint();
    slot->key = column_index;
    slot->avg_degree = avg_degree;
    slot->nonlocal_ids = nonlocal_ids ? 1 : 0;
    slot->stamp = ++g_hint_clock;
    return GNNA_OK;
}

int gnna_set_graph_phases(const int32_t *column_index, int dim, int column_phases)
{
    if (!column_index || dim <= 0 || column_phases < 0 || column_phases > gnna::kMaxSlices)
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "gnna_set_graph_phases: bad argument (dim=%d phases=%d)", dim,
                          column_phases);
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    GraphHint *slot = nullptr;
    for (auto &h : g_hints)
        if (h.key == column_index) slot = &h;
    if (!slot) {
        slot = &g_hints[0];
        for (auto &h : g_hints) {
            if (!h.key) { slot = &h; break; }
            if (h.stamp < slot->stamp) slot = &h;
        }
        *slot = GraphHint();
        slot->key = column_index;
    }
    slot->stamp = ++g_hint_clock;
    int at = -1;
    for (int i = 0; i < 8; i++)
        if (slot->sched_dim[i] == dim) at = i;
    for (int i = 0; i < 8 && at < 0; i++)
        if (slot->sched_dim[i] == 0) at = i;
    if (at < 0) at = dim % 8;  // table full: overwrite some entry
    slot->sched_dim[at] = column_phases > 0 ? dim : 0;
    slot->sched_phases[at] = column_phases;
    return GNNA_OK;
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

// ---- graph inputs --------------------------------------------------------------------------


int64_t gnna_csr_from_edges_i32(const int32_t *src, const int32_t *dst, int64_t num_edges, int64_t num_nodes,
                                int32_t *row_pointers, int32_t *column_index)
{
    if (num_edges < 0 || num_nodes < 0 || !row_pointers || (num_edges > 0 && (!src || !dst || !column_index)))
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "bad edge list arguments");
    if (num_edges > 0x7fffffffLL) return gnna::fail(GNNA_ERR_UNSUPPORTED, "more than 2^31-1 edges: shard the graph");
    // counting sort by source row, the edge list cut into slabs that are counted and scattered in parallel (atomic counts and
    // cursors into ONE array: the order inside a row does not matter, every row is sorted below)
    std::vector<int64_t> start((size_t)num_nodes + 1, 0);
    {
        std::atomic<int64_t> bad{-1};
        parallel_rows(num_edges, [&](int64_t lo, int64_t hi) {
            for (int64_t e = lo; e < hi; e++) {
                const int32_t s = src[e], d = dst[e];
                if (s < 0 || s >= num_nodes || d < 0 || d >= num_nodes) {
                    int64_t none = -1;
                    (void)bad.compare_exchange_strong(none, e);      // (any offending edge will do for the message)
                    return;
                }
                __atomic_fetch_add(&start[(size_t)s + 1], (int64_t)1, __ATOMIC_RELAXED);
            }
        });
        const int64_t e = bad.load();
        if (e >= 0)
            return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "edge %lld (%d -> %d) outside [0, %lld)", (long long)e, src[e], dst[e],
                              (long long)num_nodes);
    }
    for (int64_t i = 0; i < num_nodes; i++) start[(size_t)i + 1] += start[(size_t)i];
    std::vector<int32_t> bucket((size_t)num_edges);
    {
        std::vector<int64_t> cursor(start.begin(), start.end() - 1);
        parallel_rows(num_edges, [&](int64_t lo, int64_t hi) {
            for (int64_t e = lo; e < hi; e++)
                bucket[(size_t)__atomic_fetch_add(&cursor[(size_t)src[e]], (int64_t)1, __ATOMIC_RELAXED)] = dst[e];
        });
    }
    // per-row sort + unique (parallel over row ranges)
    std::vector<int32_t> uniq((size_t)num_nodes, 0);
    parallel_rows(num_nodes, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            int32_t *b = bucket.data() + start[(size_t)i], *e = bucket.data() + start[(size_t)i + 1];
            std::sort(b, e);
            uniq[(size_t)i] = (int32_t)(std::unique(b, e) - b);
        }
    });
    int64_t nnz = 0;
    row_pointers[0] = 0;
    for (int64_t i = 0; i < num_nodes; i++) {
        nnz += uniq[(size_t)i];
        row_pointers[i + 1] = (int32_t)nnz;
    }
    parallel_rows(num_nodes, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++)
            std::copy_n(bucket.data() + start[(size_t)i], uniq[(size_t)i], column_index + row_pointers[i]);
    });
    return nnz;
}

// ---- sharded ingestion: 64-bit edge counts, one destination-row range per rank ---------------------------
int gnna_row_counts_i64(const int32_t *rows, int64_t num_edges, int64_t num_nodes, int64_t *counts)
{
    if (num_edges < 0 || num_nodes < 0 || (num_nodes > 0 && !counts) || (num_edges > 0 && !rows))
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "bad row count arguments");
    // slabs of the edge list counted in parallel with atomic increments into the ONE counts array (an edge list of > 2^31
    // entries is fine): no per-thread histogram -- 16 of them were 14 GB of transient memory at papers100M's 111 M rows
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(gnna::host_thread_budget(32), num_edges / (1 << 22) + 1));
    std::vector<int64_t> bad((size_t)nt, -1);
    std::vector<std::thread> th;
    const int64_t step = (num_edges + nt - 1) / nt;
    for (int64_t t = 0; t < nt; t++) {
        th.emplace_back([&, t] {
            const int64_t lo = t * step, hi = std::min(num_edges, lo + step);
            for (int64_t e = lo; e < hi; e++) {
                const int32_t r = rows[e];
                if (r < 0 || r >= num_nodes) { bad[(size_t)t] = e; return; }
                if (nt == 1) counts[r]++;
                else __atomic_fetch_add(&counts[r], (int64_t)1, __ATOMIC_RELAXED);
            }
        });
    }
    for (auto &t : th) t.join();
    for (int64_t t = 0; t < nt; t++)
        if (bad[(size_t)t] >= 0)
            return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "edge %lld: row %d outside [0, %lld)", (long long)bad[(size_t)t],
                              rows[bad[(size_t)t]], (long long)num_nodes);
    return GNNA_OK;
}

int gnna_row_splits_i64(const int64_t *counts, int64_t num_nodes, int world, int64_t *bounds, int64_t *row_pointers)
{
    if (num_nodes < 0 || world < 1 || !bounds || (num_nodes > 0 && !counts))
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "bad row split arguments");
    // global 64-bit row pointers (papers100M symmetrised has > 2^31 edges) ...
    std::vector<int64_t> own;
    int64_t *rp = row_pointers;
    if (!rp) { own.resize((size_t)num_nodes + 1); rp = own.data(); }
    rp[0] = 0;
    for (int64_t i = 0; i < num_nodes; i++) {
        if (counts[i] < 0) return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "negative count at row %lld", (long long)i);
        rp[i + 1] = rp[i] + counts[i];
    }
    // ... cut into `world` contiguous blocks of about equal edge count: block b ends behind the first row whose
    // running count reaches b / world of the total (the rule of dist.balanced_row_splits)
    const int64_t nnz = rp[num_nodes];
    bounds[0] = 0;
    for (int b = 1; b < world; b++) {
        const long double target = (long double)nnz * b / world;
        const int64_t *it = std::lower_bound(rp + 1, rp + num_nodes + 1, target,
                                             [](int64_t v, long double t) { return (long double)v < t; });
        int64_t cut = (int64_t)(it - (rp + 1)) + 1;
        cut = std::min<int64_t>(std::max<int64_t>(cut, bounds[b - 1]), num_nodes);
        bounds[b] = cut;
    }
    bounds[world] = num_nodes;
    return GNNA_OK;
}

int64_t gnna_csr_from_edges_range_i32(const int32_t *src, const int32_t *dst, int64_t num_edges, int64_t num_nodes,
                                      int64_t row_lo, int64_t row_hi, int32_t *row_pointers, int32_t *column_index,
                                      int64_t capacity)
{
    if (num_edges < 0 || num_nodes < 0 || row_lo < 0 || row_hi < row_lo || row_hi > num_nodes || !row_pointers ||
        capacity < 0 || (num_edges > 0 && (!src || !dst)) || (capacity > 0 && !column_index))
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "bad edge range arguments");
    const int64_t rows = row_hi - row_lo;
    // pass 1: the shard's rows (64-bit counts: the edge LIST may exceed 2^31 entries, the shard may not)
    std::vector<int64_t> start((size_t)rows + 1, 0);
    for (int64_t e = 0; e < num_edges; e++) {
        const int32_t s = src[e], d = dst[e];
        if (s < 0 || s >= num_nodes || d < 0 || d >= num_nodes)
            return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "edge %lld (%d -> %d) outside [0, %lld)", (long long)e, s, d,
                              (long long)num_nodes);
        if (s >= row_lo && s < row_hi) start[(size_t)(s - row_lo) + 1]++;
    }
    for (int64_t i = 0; i < rows; i++) start[(size_t)i + 1] += start[(size_t)i];
    const int64_t mine = start[(size_t)rows];
    if (mine > 0x7fffffffLL)
        return gnna::fail(GNNA_ERR_UNSUPPORTED, "%lld edges in rows [%lld, %lld): more than 2^31-1 per shard -- use more ranks",
                          (long long)mine, (long long)row_lo, (long long)row_hi);
    // pass 2: bucket by row, then per-row sort + unique (scipy coo->csr semantics, dataset.py:108-118)
    std::vector<int32_t> bucket((size_t)mine);
    {
        std::vector<int64_t> cursor(start.begin(), start.end() - 1);
        for (int64_t e = 0; e < num_edges; e++) {
            const int32_t s = src[e];
            if (s >= row_lo && s < row_hi) bucket[(size_t)cursor[(size_t)(s - row_lo)]++] = dst[e];
        }
    }
    std::vector<int32_t> uniq((size_t)rows, 0);
    parallel_rows(rows, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            int32_t *b = bucket.data() + start[(size_t)i], *e = bucket.data() + start[(size_t)i + 1];
            std::sort(b, e);
            uniq[(size_t)i] = (int32_t)(std::unique(b, e) - b);
        }
    });
    int64_t nnz = 0;
    for (int64_t i = 0; i < rows; i++) nnz += uniq[(size_t)i];
    if (nnz > capacity)
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "column_index holds %lld entries, the shard has %lld", (long long)capacity,
                          (long long)nnz);
    nnz = 0;
    row_pointers[0] = 0;
    for (int64_t i = 0; i < rows; i++) {
        std::copy_n(bucket.data() + start[(size_t)i], uniq[(size_t)i], column_index + nnz);
        nnz += uniq[(size_t)i];
        row_pointers[i + 1] = (int32_t)nnz;
    }
    return nnz;
}

int gnna_degrees_f32(const int32_t *row_pointers, int64_t num_nodes, float *degrees)
{
    if (num_nodes < 0 || (num_nodes > 0 && (!row_pointers || !degrees)))
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "bad degrees arguments");
    for (int64_t i = 0; i < num_nodes; i++) {
        const int32_t d = row_pointers[i + 1] - row_pointers[i];
        degrees[i] = std::sqrt((float)(d > 0 ? d : 1));
    }
    return GNNA_OK;
}

int gnna_edge_span(const int32_t *src, const int32_t *dst, int64_t num_edges, double *avg_edge_span)
{
    if (num_edges < 0 || !avg_edge_span || (num_edges > 0 && (!src || !dst)))
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "bad edge span arguments");
    long double acc = 0;
    for (int64_t e = 0; e < num_edges; e++) acc += std::llabs((long long)src[e] - (long long)dst[e]);
    *avg_edge_span = num_edges ? (double)(acc / num_edges) : 0.0;
    return GNNA_OK;
}

int gnna_reorder_rcm_i32(const int32_t *src, const int32_t *dst, int64_t num_edges, int64_t num_nodes,
                         int32_t *new_id)
{
    if (num_edges < 0 || num_nodes < 0 || (num_nodes > 0 && !new_id) || (num_edges > 0 && (!src || !dst)))
        return gnna::fail(GNNA_ERR_INVALID_ARGUMENT, "bad reorder arguments");
    if (num_nodes > 0x7fffffffLL || 2 * num_edges > 0x7fffffffLL)
        return gnna::fail(GNNA_ERR_UNSUPPORTED, "graph too large for int32 reorder");
    // symmetrised adjacency via the CSR builder on (src|dst, dst|src)
    std::vector<int32_t> s2((size_t)2 * num_edges), d2((size_t)2 * num_edges);
    std::copy_n(src, num_edges, s2.begin()); std::copy_n(dst, num_edges, s2.begin() + num_edges);
    std::copy_n(dst, num_edges, d2.begin()); std::copy_n(src, num_edges, d2.begin() + num_edges);
    std::vector<int32_t> rp((size_t)num_nodes + 1), ci((size_t)2 * num_edges);
    const int64_t nnz = gnna_csr_from_edges_i32(s2.data(), d2.data(), 2 * num_edges, num_nodes, rp.data(), ci.data());
    if (nnz < 0) return (int)nnz;
    auto deg = [&](int32_t v) { return rp[(size_t)v + 1] - rp[(size_t)v]; };
    // seeds: nodes by (degree, id); BFS visiting neighbours by (degree, id) = Cuthill-McKee
    std::vector<int32_t> seeds((size_t)num_nodes);
    std::iota(seeds.begin(), seeds.end(), 0);
    std::stable_sort(seeds.begin(), seeds.end(), [&](int32_t a, int32_t b) { return deg(a) < deg(b); });
    std::vector<int32_t> order;
    order.reserve((size_t)num_nodes);
    std::vector<char> seen((size_t)num_nodes, 0);
    std::vector<int32_t> nb;
    for (int32_t seed : seeds) {
        if (seen[(size_t)seed]) continue;
        seen[(size_t)seed] = 1;
        size_t head = order.size();
        order.push_back(seed);
        while (head < order.size()) {
            const int32_t v = order[head++];
            nb.clear();
            for (int32_t k = rp[(size_t)v]; k < rp[(size_t)v + 1]; k++) {
                const int32_t u = ci[(size_t)k];
                if (!seen[(size_t)u]) { seen[(size_t)u] = 1; nb.push_back(u); }
            }
            std::stable_sort(nb.begin(), nb.end(), [&](int32_t a, int32_t b) { return deg(a) < deg(b); });
            order.insert(order.end(), nb.begin(), nb.end());
        }
    }
    // reverse (RCM): the last visited node gets id 0
    for (int64_t k = 0; k < num_nodes; k++) new_id[(size_t)order[(size_t)(num_nodes - 1 - k)]] = (int32_t)k;
    return GNNA_OK;
}

#pragma GCC visibility pop
}  // extern "C"
