// Host-side temporal-CSR builder + error plumbing (no GPU code in this file).
//
// Replaces get_neighbor_sampler + NeighborSampler.__init__ (reference utils/utils.py:283-302,
// :73-110): the reference appends (neighbor, edge id, time) tuples to two Python lists per
// interaction and then sorts every list by time with the stable `sorted`.  Here: one counting
// pass for the row sizes, one placement pass in edge-list order (src entry before dst entry,
// like utils/utils.py:299-300), and a per-row stable sort only for rows that are not already
// chronological (the published datasets are, utils/utils.py:99).
#include <algorithm>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "common.h"

namespace dygnn {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

}  // namespace dygnn

extern "C" const char* dygnn_last_error(void) { return dygnn::g_last_error.c_str(); }

extern "C" int dygnn_abi_version(void) { return 14; }

extern "C" int dygnn_csr_build_host(int64_t num_edges, const int64_t* src, const int64_t* dst, const int64_t* eid,
                                    const double* ts, int64_t num_nodes, int64_t* indptr, int32_t* nbr_out,
                                    int32_t* eid_out, double* ts_out) {
    DYGNN_REQUIRE(num_edges >= 0 && num_nodes >= 1, "csr_build: bad sizes (edges=%lld nodes=%lld)",
                  (long long)num_edges, (long long)num_nodes);
    DYGNN_REQUIRE(src && dst && eid && ts && indptr && (num_edges == 0 || (nbr_out && eid_out && ts_out)),
                  "csr_build: null pointer");
    std::vector<int64_t> fill(num_nodes + 1, 0);
    for (int64_t e = 0; e < num_edges; ++e) {
        const int64_t s = src[e], d = dst[e];
        // node ids index the adjacency directly (utils/utils.py:299-300): out-of-range ids are the
        // reference's IndexError.
        DYGNN_REQUIRE(s >= 0 && s < num_nodes && d >= 0 && d < num_nodes,
                      "csr_build: node id out of range at interaction %lld (src=%lld dst=%lld, rows=%lld)",
                      (long long)e, (long long)s, (long long)d, (long long)num_nodes);
        DYGNN_REQUIRE(eid[e] >= 0 && eid[e] <= INT32_MAX && num_nodes <= INT32_MAX,
                      "csr_build: edge / node id does not fit the 32-bit CSR payload");
        ++fill[s + 1];
        ++fill[d + 1];
    }
    indptr[0] = 0;
    for (int64_t n = 0; n < num_nodes; ++n) indptr[n + 1] = indptr[n] + fill[n + 1];
    for (int64_t n = 0; n < num_nodes; ++n) fill[n] = indptr[n];
    for (int64_t e = 0; e < num_edges; ++e) {
        int64_t p = fill[src[e]]++;
        nbr_out[p] = (int32_t)dst[e]; eid_out[p] = (int32_t)eid[e]; ts_out[p] = ts[e];
        p = fill[dst[e]]++;
        nbr_out[p] = (int32_t)src[e]; eid_out[p] = (int32_t)eid[e]; ts_out[p] = ts[e];
    }
    std::vector<int64_t> perm;
    std::vector<int32_t> tn, te;
    std::vector<double> tt;
    for (int64_t n = 0; n < num_nodes; ++n) {
        const int64_t a = indptr[n], b = indptr[n + 1];
        if (b - a < 2 || std::is_sorted(ts_out + a, ts_out + b)) continue;
        const int64_t m = b - a;
        perm.resize(m);
        std::iota(perm.begin(), perm.end(), 0);
        std::stable_sort(perm.begin(), perm.end(),
                         [&](int64_t x, int64_t y) { return ts_out[a + x] < ts_out[a + y]; });
        tn.assign(nbr_out + a, nbr_out + b);
        te.assign(eid_out + a, eid_out + b);
        tt.assign(ts_out + a, ts_out + b);
        for (int64_t i = 0; i < m; ++i) {
            nbr_out[a + i] = tn[perm[i]]; eid_out[a + i] = te[perm[i]]; ts_out[a + i] = tt[perm[i]];
        }
    }
    return DYGNN_OK;
}

// ---- `uniform` neighbour sampling: the reference's draws, replayed on the host ------------------------------------------
// utils/utils.py:186-188: `self.random_state.choice(a=len(history), size=k)` per query row, on numpy's LEGACY RandomState =
// MT19937 + randint(0, n): every draw takes 32-bit outputs masked to the smallest 2^b - 1 >= n - 1 until one is <= n - 1 (rejection);
// a row with one past interaction consumes nothing.  The caller (NeighborSampler._draw_host) moves the generator's key / position out of
// and back into its numpy RandomState around the call, so the stream continues exactly where numpy's would: one library call per batch
// instead of one Python call per row (the rows of a TGAT level: 10^3 .. 10^5 per batch).
namespace {
struct Mt { uint32_t* key; int pos; };
inline void mt_refill(Mt& m) {
    constexpr int N = 624, M = 397;
    constexpr uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
    uint32_t* k = m.key;
    int i = 0;
    for (; i < N - M; ++i) { const uint32_t y = (k[i] & UP) | (k[i + 1] & LO); k[i] = k[i + M] ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
    for (; i < N - 1; ++i) { const uint32_t y = (k[i] & UP) | (k[i + 1] & LO); k[i] = k[i + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
    const uint32_t y = (k[N - 1] & UP) | (k[0] & LO);
    k[N - 1] = k[M - 1] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
    m.pos = 0;
}
inline uint32_t mt_next(Mt& m) {
    if (m.pos >= 624) mt_refill(m);
    uint32_t y = m.key[m.pos++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
}  // namespace

extern "C" int dygnn_mt19937_choice_rows_host(uint32_t* key, int32_t* pos, const int32_t* hist_len, int64_t n, int32_t k, int32_t* sampled) {
    DYGNN_REQUIRE(key && pos && hist_len && sampled && n >= 0 && k > 0, "mt19937_choice_rows: bad arguments");
    DYGNN_REQUIRE(*pos >= 0 && *pos <= 624, "mt19937_choice_rows: generator position %d outside [0, 624]", (int)*pos);
    Mt m{key, *pos};
    for (int64_t r = 0; r < n; ++r) {
        int32_t* o = sampled + r * k;
        const int32_t cnt = hist_len[r];
        if (cnt <= 1) {                       // no history: the row is skipped (utils/utils.py:178); one entry: randint(0, 1) draws nothing
            for (int32_t j = 0; j < k; ++j) o[j] = 0;
            continue;
        }
        const uint32_t rng = (uint32_t)cnt - 1u;
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        for (int32_t j = 0; j < k; ++j) {
            uint32_t v;
            do { v = mt_next(m) & mask; } while (v > rng);
            o[j] = (int32_t)v;
        }
    }
    *pos = m.pos;
    return DYGNN_OK;
}
