// comm_layout.hpp -- displacement arithmetic of the row-sharded gatherv (host only, no HIP: unit-tested by
// tests/cpp/test_gather_layout.cpp).  Rank p contributes rows[p] rows and nnz[p] entries; its rows land at
// r_off[p], its entries at n_off[p] of the concatenated CSR, and its local row offsets are rebased by n_off[p].
#pragma once
#include <cstdint>
#include <vector>

namespace speck {

struct GatherLayout {
    std::vector<uint64_t> rows, nnz;      // per rank
    std::vector<uint64_t> r_off, n_off;   // exclusive prefixes, nranks + 1 entries
    uint64_t total_rows = 0, total_nnz = 0;
};

// false when the concatenation does not fit the u32 row offsets of the dCSR layout
inline bool gather_layout(const uint64_t* rows, const uint64_t* nnz, int nranks, GatherLayout* out)
{
    out->rows.assign(rows, rows + nranks);
    out->nnz.assign(nnz, nnz + nranks);
    out->r_off.assign(size_t(nranks) + 1, 0);
    out->n_off.assign(size_t(nranks) + 1, 0);
    for (int p = 0; p < nranks; ++p) {
        out->r_off[p + 1] = out->r_off[p] + rows[p];
        out->n_off[p + 1] = out->n_off[p] + nnz[p];
    }
    out->total_rows = out->r_off[nranks];
    out->total_nnz = out->n_off[nranks];
    return out->total_nnz <= 0xFFFFFFFFull && out->total_rows <= 0xFFFFFFFFull;
}

// the rank whose row range holds global row r (empty ranges own nothing); what the rebase kernel computes
inline int owner_of_row(const GatherLayout& l, uint64_t r)
{
    int lo = 0, hi = (int)l.rows.size() - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (l.r_off[mid] <= r) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// host restatement of the exchange on plain arrays (the oracle of the unit test and of the gloo tests):
// concatenates shards with LOCAL row offsets into one CSR with global offsets
inline void concat_shards(const GatherLayout& l, const std::vector<std::vector<uint32_t>>& ro_local,
                          const std::vector<std::vector<uint32_t>>& col, std::vector<uint32_t>* ro_out,
                          std::vector<uint32_t>* col_out)
{
    ro_out->assign(l.total_rows + 1, 0);
    col_out->assign(l.total_nnz, 0);
    for (size_t p = 0; p < l.rows.size(); ++p) {
        for (uint64_t i = 0; i < l.rows[p]; ++i) (*ro_out)[l.r_off[p] + i] = ro_local[p][i] + (uint32_t)l.n_off[p];
        for (uint64_t i = 0; i < l.nnz[p]; ++i) (*col_out)[l.n_off[p] + i] = col[p][i];
    }
    (*ro_out)[l.total_rows] = (uint32_t)l.total_nnz;
}

}  // namespace speck
