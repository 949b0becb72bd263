// Unit test of the displacement arithmetic of the row-sharded gatherv (speck_amd/csrc/comm_layout.hpp): plain C++,
// no GPU.  Build + run: g++ -std=c++17 -I speck_amd/csrc tests/cpp/test_gather_layout.cpp -o /tmp/t && /tmp/t
#include <cassert>
#include <cstdio>
#include <random>

#include "comm_layout.hpp"

using namespace speck;

static void check(const std::vector<uint64_t>& rows, uint64_t seed)
{
    const int n = (int)rows.size();
    std::mt19937_64 rng(seed);
    // shards with LOCAL offsets and recognisable column ids
    std::vector<std::vector<uint32_t>> ro(n), col(n);
    std::vector<uint64_t> nnz(n);
    std::vector<uint32_t> want_ro{0}, want_col;
    for (int p = 0; p < n; ++p) {
        uint32_t run = 0;
        for (uint64_t i = 0; i < rows[p]; ++i) {
            ro[p].push_back(run);
            const uint32_t len = (uint32_t)(rng() % 4);  // empty rows included
            for (uint32_t k = 0; k < len; ++k) {
                col[p].push_back((uint32_t)(p * 1000003u + run + k));
                want_col.push_back(col[p].back());
            }
            run += len;
            want_ro.push_back((uint32_t)want_col.size());
        }
        nnz[p] = run;
    }
    GatherLayout l;
    assert(gather_layout(rows.data(), nnz.data(), n, &l));
    assert(l.total_rows + 1 == want_ro.size() && l.total_nnz == want_col.size());
    for (int p = 0; p < n; ++p) {
        assert(l.r_off[p + 1] - l.r_off[p] == rows[p] && l.n_off[p + 1] - l.n_off[p] == nnz[p]);
        for (uint64_t r = l.r_off[p]; r < l.r_off[p + 1]; ++r) assert(owner_of_row(l, r) == p);
    }
    std::vector<uint32_t> got_ro, got_col;
    concat_shards(l, ro, col, &got_ro, &got_col);
    assert(got_ro == want_ro && got_col == want_col);
    // the rebase as the device kernel does it: received local offsets + n_off[owner]
    std::vector<uint32_t> dev(l.total_rows + 1);
    for (int p = 0; p < n; ++p)
        for (uint64_t i = 0; i < rows[p]; ++i) dev[l.r_off[p] + i] = ro[p][i];
    for (uint64_t r = 0; r < l.total_rows; ++r) dev[r] += (uint32_t)l.n_off[owner_of_row(l, r)];
    dev[l.total_rows] = (uint32_t)l.total_nnz;
    assert(dev == want_ro);
}

int main()
{
    check({5}, 1);
    check({3, 4}, 2);
    check({0, 7, 0, 2}, 3);            // ranks without rows
    check({10, 0, 0, 0, 0, 0, 0, 1}, 4);
    check({1000, 999, 1001, 1, 0, 500, 250, 249}, 5);
    check({0, 0, 0}, 6);               // nothing at all
    // overflow of the u32 offsets is reported, not wrapped
    GatherLayout l;
    const uint64_t rows2[2] = {1, 1}, big[2] = {0xFFFFFFFFull, 1};
    assert(!gather_layout(rows2, big, 2, &l));
    const uint64_t ok[2] = {0xFFFFFFFEull, 1};
    assert(gather_layout(rows2, ok, 2, &l) && l.total_nnz == 0xFFFFFFFFull);
    std::puts("gather layout ok");
    return 0;
}
