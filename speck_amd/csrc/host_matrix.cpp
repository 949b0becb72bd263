// host_matrix.cpp -- host-side CSR container, synthetic SuiteSparse stand-ins and the
// on-disk formats of the reference driver.  Pure CPU code inside libspeck_amd.so.
//   * CSR<T> container            -- reference include/CSR.h:57-65
//   * MatrixMarket reader         -- reference source/COO.cpp:53-164 (coordinate only;
//     real/integer/double/pattern/complex(real part); general/symmetric/Hermitian mirrored
//     without dedup; 1-based -> 0-based)
//   * COO -> CSR                  -- reference source/CSR.cpp:173-212 (sort by (row, col))
//   * .hicsr read/write           -- reference source/CSR.cpp:27-73, 88-137 (80-byte header,
//     16-byte State<T>, then data, col_ids, row_offsets)
//   * "<path>d_.hicsr" cache rule -- reference source/DataLoader.cpp:9-58
// Synthetic generators follow SURVEY.md section 8d (no SuiteSparse files offline).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <functional>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/speck_c_api.h"

struct speck_host_csr {
    uint64_t rows = 0, cols = 0;
    std::vector<uint32_t> row_offsets;  // rows + 1
    std::vector<uint32_t> col_ids;
    std::vector<double> data;
};

namespace {

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double unit() { return double(next() >> 11) * 0x1.0p-53; }  // [0,1)
    uint64_t below(uint64_t n) { return next() % n; }
};

double draw_value(SplitMix64& g, bool signed_values)
{
    double v = 0.5 + g.unit();
    if (signed_values && (g.next() & 1)) v = -v;
    return v;
}

// Appends one row: sorts + dedups the candidate columns, then draws the values in
// ascending column order.
void emit_row(speck_host_csr& m, std::vector<uint32_t>& cand, SplitMix64& g, bool signed_values)
{
    std::sort(cand.begin(), cand.end());
    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
    for (uint32_t c : cand) {
        m.col_ids.push_back(c);
        m.data.push_back(draw_value(g, signed_values));
    }
    m.row_offsets.push_back((uint32_t)m.col_ids.size());
    cand.clear();
}

// config #1: the exact sequence of draws matters -- known answers
// nnzA=199976, P=3994059, nnzC=3911793 for n=10000, seed=42.
void gen_uniform(speck_host_csr& m, uint32_t n, uint64_t seed, bool signed_values)
{
    SplitMix64 g(seed);
    std::vector<uint8_t> used(n, 0);
    std::vector<uint32_t> cand;
    m.rows = m.cols = n;
    m.row_offsets.assign(1, 0);
    for (uint32_t r = 0; r < n; ++r) {
        uint32_t k = 10 + (uint32_t)(g.next() % 21);
        if (k > n) k = n;
        while (cand.size() < k) {
            uint32_t c = (uint32_t)(g.next() % n);
            if (!used[c]) {
                used[c] = 1;
                cand.push_back(c);
            }
        }
        std::sort(cand.begin(), cand.end());
        for (uint32_t c : cand) {
            used[c] = 0;
            m.col_ids.push_back(c);
            m.data.push_back(draw_value(g, signed_values));
        }
        m.row_offsets.push_back((uint32_t)m.col_ids.size());
        cand.clear();
    }
}

// Discrete power law on [1, cap] with P(k) ~ k^-alpha; alpha solved so that the mean is `mean`.
struct PowerLawLen {
    std::vector<double> cdf;
    PowerLawLen(uint32_t cap, double mean)
    {
        double lo = 0.5, hi = 6.0;
        for (int it = 0; it < 60; ++it) {
            double a = 0.5 * (lo + hi), z = 0, mu = 0;
            for (uint32_t k = 1; k <= cap; ++k) {
                double p = std::pow((double)k, -a);
                z += p;
                mu += p * k;
            }
            if (mu / z > mean) lo = a; else hi = a;
        }
        double a = 0.5 * (lo + hi), z = 0;
        cdf.resize(cap);
        for (uint32_t k = 1; k <= cap; ++k) {
            z += std::pow((double)k, -a);
            cdf[k - 1] = z;
        }
        for (auto& c : cdf) c /= z;
    }
    uint32_t draw(SplitMix64& g) const
    {
        double u = g.unit();
        return (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()) + 1;
    }
};

// Community model shared by the scircuit-, webbase- and mac_econ-like stand-ins.  Row lengths are
// drawn first (k in [kmin, cap], P(k) ~ k^-alpha with alpha solved for the wanted mean), then every
// entry of row r picks its column
//   * with p_local inside r's own COMMUNITY (`block` consecutive ids), position = block * u^skew:
//     skew > 1 concentrates the picks of all rows of a community on its first ids (the pages every
//     page of a host links to) -- the products of a row then pile up on few columns, which is what
//     gives A*A its compression P / nnz(C);
//   * with p_pref one of the `hubs` LONGEST rows, rank = hubs * u^hub_skew (hub rows are hub columns,
//     as in circuit and web graphs, and everybody links to the same few of them): raises
//     P / nnz(A) above the mean row length, and the long rows that hubs reference overlap;
//   * else uniformly; a pick that lands on a row longer than `short_cut` is redrawn once with
//     probability p_short (references biased to short rows: lowers P / nnz(A), mac_econ).
// `symmetric`: every entry is mirrored and the diagonal is present (structurally symmetric circuit
// matrices): P = sum of squared row lengths, and every row of A*A meets itself and each neighbour
// at least twice.
// The parameter sets below were fitted with scripts/calibrate_standins.py against SURVEY.md's table
// (n, nnz(A), longest row, P, nnz(C)); tests/test_host.py asserts the fit.
struct CommunityParams {
    double mean_len;
    uint32_t kmin, cap;
    uint32_t block;
    double p_local, skew, p_pref, p_short;
    uint32_t hubs;
    double hub_skew;
    uint32_t short_cut;
    int symmetric;
    uint32_t skew_from;  // rows shorter than this pick uniformly (inside the community and among the hubs'
                         //   complement): leaf pages do not carry the product count
    double density;      // > 0: the local picks of a non-leaf row fall into the first max(tmpl, len / density)
    uint32_t tmpl;       //   ids of its community instead (windows nested at the community's start: the
                         //   hubs of a community are dense there and overlap, and everybody links to them)
    int sort_block;  // the rows of a community are ordered by descending length: its first ids (the ones
                     //   the skewed local picks prefer) are its hubs
};

// `key=value,key=value` overrides from the environment (calibration runs only)
void apply_overrides(CommunityParams& p)
{
    const char* e = std::getenv("SPECK_GEN_PARAMS");
    if (!e) return;
    std::stringstream ss(e);
    for (std::string kv; std::getline(ss, kv, ',');) {
        const size_t eq = kv.find('=');
        if (eq == std::string::npos) continue;
        const std::string k = kv.substr(0, eq);
        const double v = std::atof(kv.c_str() + eq + 1);
        if (k == "mean") p.mean_len = v;
        else if (k == "kmin") p.kmin = (uint32_t)v;
        else if (k == "cap") p.cap = (uint32_t)v;
        else if (k == "block") p.block = (uint32_t)v;
        else if (k == "p_local") p.p_local = v;
        else if (k == "skew") p.skew = v;
        else if (k == "p_pref") p.p_pref = v;
        else if (k == "p_short") p.p_short = v;
        else if (k == "hubs") p.hubs = (uint32_t)v;
        else if (k == "hub_skew") p.hub_skew = v;
        else if (k == "short_cut") p.short_cut = (uint32_t)v;
        else if (k == "symmetric") p.symmetric = (int)v;
        else if (k == "sort_block") p.sort_block = (int)v;
        else if (k == "skew_from") p.skew_from = (uint32_t)v;
        else if (k == "density") p.density = v;
        else if (k == "tmpl") p.tmpl = (uint32_t)v;
    }
}

void gen_community(speck_host_csr& m, uint32_t n, uint64_t seed, bool signed_values, CommunityParams p)
{
    apply_overrides(p);
    SplitMix64 g(seed);
    p.cap = std::max(p.cap, p.kmin);
    PowerLawLen pl(p.cap - p.kmin + 1, std::max(1.0, p.mean_len - (p.kmin - 1)));
    std::vector<uint32_t> len(n);
    for (uint32_t r = 0; r < n; ++r) len[r] = std::min<uint32_t>(pl.draw(g) + p.kmin - 1, n);
    const uint32_t block = std::max(1u, std::min(p.block, n));
    if (p.sort_block)
        for (uint32_t b0 = 0; b0 < n; b0 += block)
            std::sort(len.begin() + b0, len.begin() + std::min(n, b0 + block), std::greater<uint32_t>());
    // the `hubs` longest rows, longest first (ties: lower id first)
    const uint32_t n_hubs = std::max(1u, std::min(p.hubs, n));
    std::vector<uint32_t> hub(n);
    for (uint32_t r = 0; r < n; ++r) hub[r] = r;
    std::partial_sort(hub.begin(), hub.begin() + n_hubs, hub.end(), [&](uint32_t a, uint32_t b) {
        return len[a] != len[b] ? len[a] > len[b] : a < b;
    });
    uint64_t total_len = 0;
    for (uint32_t r = 0; r < n; ++r) total_len += len[r];
    std::vector<uint64_t> edges;  // row << 32 | col
    edges.reserve((size_t)(total_len * (p.symmetric ? 2 : 1)) + n);
    for (uint32_t r = 0; r < n; ++r) {
        // a long row spreads over as many consecutive communities as it needs to keep its entries
        uint32_t wide = block;
        while (wide < n && 2.0 * p.p_local * len[r] > wide) wide *= 2;
        const uint32_t b0 = r / wide * wide, bn = std::min(wide, n - b0);
        const bool leaf = len[r] < p.skew_from;
        for (uint32_t j = 0; j < len[r]; ++j) {
            const double u = g.unit();
            uint32_t c;
            if (u < p.p_local) {
                const double v = g.unit();
                if (p.density > 0 && !leaf) {
                    const uint32_t o0 = r / block * block;
                    const uint64_t win = std::max<uint64_t>(p.tmpl, (uint64_t)(len[r] / p.density) + 1);
                    c = (uint32_t)std::min<uint64_t>(n - 1, o0 + (uint64_t)(win * v));
                } else
                    c = b0 + std::min(bn - 1, (uint32_t)(bn * (p.skew == 1.0 || leaf ? v : std::pow(v, p.skew))));
            } else if (!leaf && u < p.p_local + p.p_pref) {
                c = hub[std::min(n_hubs - 1, (uint32_t)(n_hubs * std::pow(g.unit(), p.hub_skew)))];
            } else {
                c = (uint32_t)g.below(n);
                if (len[c] > p.short_cut && g.unit() < p.p_short) c = (uint32_t)g.below(n);
            }
            edges.push_back(uint64_t(r) << 32 | c);
            if (p.symmetric) edges.push_back(uint64_t(c) << 32 | r);
        }
        if (p.symmetric) edges.push_back(uint64_t(r) << 32 | r);
    }
    std::sort(edges.begin(), edges.end());
    edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
    m.rows = m.cols = n;
    m.row_offsets.assign(n + 1, 0);
    m.col_ids.resize(edges.size());
    m.data.resize(edges.size());
    for (size_t i = 0; i < edges.size(); ++i) {  // values in row-major, ascending-column order
        m.col_ids[i] = (uint32_t)edges[i];
        m.data[i] = draw_value(g, signed_values);
        ++m.row_offsets[(edges[i] >> 32) + 1];
    }
    for (uint32_t r = 0; r < n; ++r) m.row_offsets[r + 1] += m.row_offsets[r];
}

// cant-like: 3x3-block FEM on a (nx, ny, nz) node grid, 24 of the 27 neighbours
// (3 of the 8 cube corners are dropped) -> ~64 nnz/row, banded.
void gen_cant(speck_host_csr& m, uint32_t n, uint64_t seed, bool signed_values)
{
    SplitMix64 g(seed);
    const uint32_t nodes = n / 3;
    uint32_t nx = 14, ny = 12;  // x fastest; half bandwidth ~ 3*(nx*ny + nx + 1) ~ 550
    int drop = 3;               // fitted like the community parameters (scripts/calibrate_standins.py)
    if (const char* e = std::getenv("SPECK_GEN_PARAMS")) {
        int a = 0, b = 0, c = 0;
        if (std::sscanf(e, "nx=%d,ny=%d,drop=%d", &a, &b, &c) == 3) nx = a, ny = b, drop = c;
    }
    m.rows = m.cols = n;
    m.row_offsets.assign(1, 0);
    std::vector<uint32_t> cand;
    for (uint32_t r = 0; r < n; ++r) {
        const uint32_t node = std::min(r / 3, nodes ? nodes - 1 : 0);
        const int64_t x = node % nx, y = (node / nx) % ny, z = node / (nx * ny);
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (dz != 0 && dy != 0 && dx != 0) {  // the 8 cube corners: `drop` of them are missing
                        const int corner = (dz > 0) * 4 + (dy > 0) * 2 + (dx > 0);
                        static const int order[8] = {0, 7, 3, 4, 1, 6, 2, 5};
                        bool skip = false;
                        for (int q = 0; q < drop; ++q) skip |= order[q] == corner;
                        if (skip) continue;
                    }
                    const int64_t xx = x + dx, yy = y + dy, zz = z + dz;
                    if (xx < 0 || yy < 0 || zz < 0 || xx >= nx || yy >= ny) continue;
                    const int64_t nb = (zz * ny + yy) * nx + xx;
                    if (nb >= (int64_t)nodes) continue;
                    for (int d = 0; d < 3; ++d) cand.push_back((uint32_t)(nb * 3 + d));
                }
        if (r >= nodes * 3) cand.push_back(r);  // leftover rows when n % 3 != 0
        emit_row(m, cand, g, signed_values);
    }
}

// nlpkkt-like: 27-point stencil on a g^3 grid, A*A = 125-point stencil.
void gen_stencil27(speck_host_csr& m, uint32_t gdim, uint64_t seed, bool signed_values)
{
    SplitMix64 g(seed);
    const uint64_t n = (uint64_t)gdim * gdim * gdim;
    m.rows = m.cols = n;
    m.row_offsets.assign(1, 0);
    m.col_ids.reserve(n * 27);
    m.data.reserve(n * 27);
    for (uint64_t r = 0; r < n; ++r) {
        const int64_t x = r % gdim, y = (r / gdim) % gdim, z = r / ((uint64_t)gdim * gdim);
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int64_t xx = x + dx, yy = y + dy, zz = z + dz;
                    if (xx < 0 || yy < 0 || zz < 0 || xx >= gdim || yy >= gdim || zz >= gdim) continue;
                    m.col_ids.push_back((uint32_t)((zz * gdim + yy) * gdim + xx));
                    m.data.push_back(draw_value(g, signed_values));
                }
        m.row_offsets.push_back((uint32_t)m.col_ids.size());
    }
}

// ---- .hicsr -------------------------------------------------------------------
#pragma pack(push, 1)
struct HicsrHeader {  // reference CSRIOHeader, source/CSR.cpp:27-73 (natural alignment: 9 + 7 pad)
    char magic[9];
    char pad[7];
    uint64_t typesize, compresseddir, indexsize, fixedoffset, offsetsize, num_rows, num_columns,
        num_non_zeroes;
};
struct HicsrState {  // reference State<double>, source/CSR.cpp:15-25
    double scaling;
    uint8_t transpose;
    char pad[7];
};
#pragma pack(pop)
static_assert(sizeof(HicsrHeader) == 80, "header is 80 bytes in the reference");
static_assert(sizeof(HicsrState) == 16, "State<double> is 16 bytes in the reference");
const char kMagic[9] = {'H', 'i', 1, 'C', 'o', 'm', 'p', 's', 'd'};

int load_hicsr(const char* path, speck_host_csr& m)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return SPECK_ERR_IO;
    HicsrHeader h;
    HicsrState st;
    f.read(reinterpret_cast<char*>(&h), sizeof(h));
    if (!f.good() || std::memcmp(h.magic, kMagic, 9) != 0) return SPECK_ERR_IO;
    f.read(reinterpret_cast<char*>(&st), sizeof(st));
    if (!f.good() || h.typesize != sizeof(double) || h.indexsize != sizeof(uint32_t) ||
        h.offsetsize != sizeof(uint32_t))
        return SPECK_ERR_IO;
    // a stale or corrupt cache must not turn into a huge allocation or, later, into device reads
    // out of bounds: the payload the header announces has to be exactly what the file holds
    if (h.num_non_zeroes > 0xFFFFFFFFull || h.num_rows > 0xFFFFFFFEull || h.num_columns > 0xFFFFFFFFull)
        return SPECK_ERR_IO;
    const uint64_t payload = h.num_non_zeroes * 12 + (h.num_rows + 1) * 4;
    f.seekg(0, std::ios::end);
    const uint64_t fsize = (uint64_t)f.tellg();
    if (fsize != sizeof(h) + sizeof(st) + payload) return SPECK_ERR_IO;
    f.seekg(sizeof(h) + sizeof(st), std::ios::beg);
    m.rows = h.num_rows;
    m.cols = h.num_columns;
    m.data.resize(h.num_non_zeroes);
    m.col_ids.resize(h.num_non_zeroes);
    m.row_offsets.resize(h.num_rows + 1);
    f.read(reinterpret_cast<char*>(m.data.data()), m.data.size() * sizeof(double));
    f.read(reinterpret_cast<char*>(m.col_ids.data()), m.col_ids.size() * sizeof(uint32_t));
    f.read(reinterpret_cast<char*>(m.row_offsets.data()), m.row_offsets.size() * sizeof(uint32_t));
    if (!f.good()) return SPECK_ERR_IO;
    if (m.row_offsets[0] != 0 || m.row_offsets[m.rows] != h.num_non_zeroes) return SPECK_ERR_IO;
    for (uint64_t r = 0; r < m.rows; ++r)
        if (m.row_offsets[r] > m.row_offsets[r + 1]) return SPECK_ERR_IO;
    for (uint32_t c : m.col_ids)
        if (c >= m.cols) return SPECK_ERR_IO;
    return SPECK_OK;
}

int store_hicsr(const speck_host_csr& m, const char* path)
{
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) return SPECK_ERR_IO;
    HicsrHeader h{};
    std::memcpy(h.magic, kMagic, 9);
    h.typesize = sizeof(double);
    h.compresseddir = 0;
    h.indexsize = sizeof(uint32_t);
    h.fixedoffset = 0;
    h.offsetsize = sizeof(uint32_t);
    h.num_rows = m.rows;
    h.num_columns = m.cols;
    h.num_non_zeroes = m.col_ids.size();
    HicsrState st{};
    st.scaling = 1.0;
    st.transpose = 0;
    f.write(reinterpret_cast<const char*>(&h), sizeof(h));
    f.write(reinterpret_cast<const char*>(&st), sizeof(st));
    f.write(reinterpret_cast<const char*>(m.data.data()), m.data.size() * sizeof(double));
    f.write(reinterpret_cast<const char*>(m.col_ids.data()), m.col_ids.size() * sizeof(uint32_t));
    f.write(reinterpret_cast<const char*>(m.row_offsets.data()), m.row_offsets.size() * sizeof(uint32_t));
    return f.good() ? SPECK_OK : SPECK_ERR_IO;
}

// ---- MatrixMarket ---------------------------------------------------------------
// Semantics of the reference's loadMTX + convert(COO -> CSR) (source/COO.cpp:53-164, source/CSR.cpp:173-212):
// coordinate format only; real / integer / double / pattern (value 1) / complex (real part); general, symmetric
// and Hermitian -- the latter two mirror every off-diagonal entry WITHOUT deduplication; 1-based -> 0-based;
// rows sorted by column, entries with the same (row, column) keep their file order.
// Built for SuiteSparse-sized files (nlpkkt160: 115 M lines, 230 M mirrored entries): the file is mapped, cut
// into chunks at line boundaries and parsed by all host threads with std::from_chars -- twice: the first pass
// only counts the entries of every (chunk, row), the second writes each entry straight to its place in the CSR
// arrays (offset of the row + entries of earlier chunks in that row: file order inside a row is kept without
// sorting a 16-byte-per-entry COO copy).  Peak memory = the CSR itself + 4 B x rows x chunks of cursors.
// Rows that are not already ascending (most files are written column-major, so they are) are sorted in place.
struct MappedFile {
    const char* p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~MappedFile()
    {
        if (p && n) munmap(const_cast<char*>(p), n);
        if (fd >= 0) close(fd);
    }
};

inline const char* skip_blank(const char* p, const char* e)
{
    while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    return p;
}
inline const char* line_end(const char* p, const char* e)
{
    const void* q = std::memchr(p, '\n', size_t(e - p));
    return q ? static_cast<const char*>(q) : e;
}
// unsigned integer token (what `istream >> uint32_t` accepts for the files in question: an optional '+', digits)
inline bool parse_u64(const char*& p, const char* e, uint64_t& v)
{
    p = skip_blank(p, e);
    if (p < e && *p == '+') ++p;
    auto r = std::from_chars(p, e, v);
    if (r.ec != std::errc()) return false;
    p = r.ptr;
    return true;
}
inline bool parse_f64(const char*& p, const char* e, double& v)
{
    p = skip_blank(p, e);
    if (p < e && *p == '+') ++p;
    auto r = std::from_chars(p, e, v);
    if (r.ec == std::errc::result_out_of_range) {  // denormal / overflow: strtod's answer, as iostreams give it
        std::string tok(p, r.ptr);
        v = std::strtod(tok.c_str(), nullptr);
    } else if (r.ec != std::errc())
        return false;
    p = r.ptr;
    return true;
}

// one data line: 0 = blank / comment, 1 = entry, -1 = malformed
inline int parse_entry(const char* p, const char* e, bool pattern, bool need_value, uint64_t rows, uint64_t cols,
                       uint32_t& r, uint32_t& c, double& d)
{
    p = skip_blank(p, e);
    if (p == e || *p == '%') return 0;
    uint64_t r64, c64;
    if (!parse_u64(p, e, r64) || !parse_u64(p, e, c64)) return -1;
    if (r64 == 0 || c64 == 0 || r64 > rows || c64 > cols) return -1;
    r = (uint32_t)(r64 - 1);
    c = (uint32_t)(c64 - 1);
    d = 1.0;
    // (a complex file: the real part, the rest of the line is ignored; the counting pass skips the value)
    if (!pattern && need_value && !parse_f64(p, e, d)) return -1;
    return 1;
}

int load_mtx(const char* path, speck_host_csr& m)
{
    MappedFile mf;
    mf.fd = open(path, O_RDONLY);
    if (mf.fd < 0) return SPECK_ERR_IO;
    struct stat st;
    if (fstat(mf.fd, &st) != 0 || st.st_size <= 0) return SPECK_ERR_IO;
    mf.n = (size_t)st.st_size;
    void* map = mmap(nullptr, mf.n, PROT_READ, MAP_PRIVATE, mf.fd, 0);
    if (map == MAP_FAILED) {
        mf.n = 0;
        return SPECK_ERR_IO;
    }
    mf.p = static_cast<const char*>(map);
    (void)madvise(map, mf.n, MADV_SEQUENTIAL);
    const char* p = mf.p;
    const char* const end = mf.p + mf.n;

    // banner
    const char* le = line_end(p, end);
    std::string line(p, le);
    if (line.compare(0, 32, "%%MatrixMarket matrix coordinate") != 0) return SPECK_ERR_IO;
    std::istringstream hs(line);
    std::vector<std::string> tok;
    for (std::string t; hs >> t;) tok.push_back(t);
    if (tok.size() < 5) return SPECK_ERR_IO;
    bool pattern = false, mirror = false;
    if (tok[3] == "pattern") pattern = true;
    else if (tok[3] == "complex") { /* real part only, as the reference's `liness >> d` does */ }
    else if (tok[3] != "real" && tok[3] != "integer" && tok[3] != "double") return SPECK_ERR_IO;
    if (tok[4] == "general") mirror = false;
    else if (tok[4] == "symmetric" || tok[4] == "Hermitian") mirror = true;
    else return SPECK_ERR_IO;
    p = le < end ? le + 1 : end;

    // size line: the first line that is not a comment
    uint64_t rows = 0, cols = 0, nnz = 0;
    bool have_size = false;
    while (p < end) {
        le = line_end(p, end);
        if (*p != '%') {
            const char* q = p;
            if (!parse_u64(q, le, rows) || !parse_u64(q, le, cols) || !parse_u64(q, le, nnz)) return SPECK_ERR_IO;
            have_size = true;
            p = le < end ? le + 1 : end;
            break;
        }
        p = le < end ? le + 1 : end;
    }
    if (!have_size) return SPECK_ERR_IO;
    if (rows > 0xFFFFFFFEull || cols > 0xFFFFFFFFull || nnz > 0x7FFFFFFFull) return SPECK_ERR_IO;

    // chunks of whole lines
    const size_t body = size_t(end - p);
    unsigned hw = std::thread::hardware_concurrency();
    size_t nchunks = std::max<size_t>(1, std::min<size_t>({hw ? hw : 1u, 32u, body / (1u << 20) + 1}));
    // the cursors cost 4 B x rows x chunks: keep them below ~half of the CSR they help to build
    while (nchunks > 1 && nchunks * rows * 4 > 6 * (mirror ? 2 * nnz : nnz) + (64u << 20)) --nchunks;
    std::vector<const char*> cut(nchunks + 1);
    cut[0] = p;
    cut[nchunks] = end;
    for (size_t k = 1; k < nchunks; ++k) {
        const char* q = p + body / nchunks * k;
        q = line_end(q, end);
        cut[k] = q < end ? q + 1 : end;
    }
    for (size_t k = 1; k <= nchunks; ++k)
        if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];

    // pass 1: entries per (chunk, row)
    std::vector<std::vector<uint32_t>> cnt(nchunks);
    std::vector<int> bad(nchunks, 0);
    auto run = [&](auto&& fn) {
        std::vector<std::thread> th;
        for (size_t k = 1; k < nchunks; ++k) th.emplace_back(fn, k);
        fn(size_t(0));
        for (auto& t : th) t.join();
    };
    run([&](size_t k) {
        auto& c = cnt[k];
        c.assign(rows, 0);
        const char* q = cut[k];
        const char* const qe = cut[k + 1];
        while (q < qe) {
            const char* l = line_end(q, qe);
            uint32_t r, cc;
            double d;
            const int got = parse_entry(q, l, pattern, false, rows, cols, r, cc, d);
            if (got < 0) {
                bad[k] = 1;
                return;
            }
            if (got > 0) {
                ++c[r];
                if (mirror && r != cc) {
                    if (cc >= rows) {  // the mirrored entry needs a row: symmetric files are square
                        bad[k] = 1;
                        return;
                    }
                    ++c[cc];
                }
            }
            q = l < qe ? l + 1 : qe;
        }
    });
    for (int b : bad)
        if (b) return SPECK_ERR_IO;

    // row offsets; cnt[k][r] becomes the cursor of chunk k in row r
    m.rows = rows;
    m.cols = cols;
    m.row_offsets.assign(rows + 1, 0);
    uint64_t total = 0;
    for (uint64_t r = 0; r < rows; ++r) {
        m.row_offsets[r] = (uint32_t)total;
        for (size_t k = 0; k < nchunks; ++k) {
            const uint32_t c = cnt[k][r];
            cnt[k][r] = (uint32_t)total;
            total += c;
        }
        if (total > 0xFFFFFFFFull) return SPECK_ERR_IO;
    }
    m.row_offsets[rows] = (uint32_t)total;
    m.col_ids.resize(total);
    m.data.resize(total);

    // pass 2: every entry to its place (file order inside a row)
    run([&](size_t k) {
        auto& cur = cnt[k];
        const char* q = cut[k];
        const char* const qe = cut[k + 1];
        while (q < qe) {
            const char* l = line_end(q, qe);
            uint32_t r, cc;
            double d;
            const int got = parse_entry(q, l, pattern, true, rows, cols, r, cc, d);
            if (got < 0) {
                bad[k] = 1;
                return;
            }
            if (got > 0) {
                uint32_t at = cur[r]++;
                m.col_ids[at] = cc;
                m.data[at] = d;
                if (mirror && r != cc) {  // no dedup (reference COO.cpp:153-159)
                    at = cur[cc]++;
                    m.col_ids[at] = r;
                    m.data[at] = d;
                }
            }
            q = l < qe ? l + 1 : qe;
        }
    });
    cnt.clear();
    cnt.shrink_to_fit();
    for (int b : bad)
        if (b) return SPECK_ERR_IO;

    // rows ascending by column; equal columns keep their file order (the reference std::sorts by (row, column): the
    // order of exact duplicates is unspecified there)
    const size_t nsort = std::max<size_t>(1, std::min<size_t>(hw ? hw : 1u, 32u));
    std::vector<std::thread> th;
    auto sort_rows = [&](size_t t) {
        std::vector<std::pair<uint32_t, double>> tmp;
        const uint64_t r0 = rows * t / nsort, r1 = rows * (t + 1) / nsort;
        for (uint64_t r = r0; r < r1; ++r) {
            const uint32_t a = m.row_offsets[r], b = m.row_offsets[r + 1];
            bool sorted = true;
            for (uint32_t i = a + 1; i < b && sorted; ++i) sorted = m.col_ids[i - 1] <= m.col_ids[i];
            if (sorted) continue;
            tmp.resize(b - a);
            for (uint32_t i = a; i < b; ++i) tmp[i - a] = {m.col_ids[i], m.data[i]};
            std::stable_sort(tmp.begin(), tmp.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
            for (uint32_t i = a; i < b; ++i) {
                m.col_ids[i] = tmp[i - a].first;
                m.data[i] = tmp[i - a].second;
            }
        }
    };
    for (size_t t = 1; t < nsort; ++t) th.emplace_back(sort_rows, t);
    sort_rows(0);
    for (auto& t : th) t.join();
    return SPECK_OK;
}

}  // namespace

extern "C" {

int speck_gen_matrix(const char* kind, double scale, uint64_t seed, int signed_values,
                     speck_host_csr** out)
{
    if (!kind || !out || !(scale > 0)) return SPECK_ERR_INVALID;
    auto* m = new speck_host_csr();
    const std::string k(kind);
    const bool sg = signed_values != 0;
    auto scaled = [&](double n0) { return (uint32_t)std::max(1.0, std::floor(n0 * scale + 0.5)); };
    if (k == "uniform") gen_uniform(*m, scaled(10000), seed, sg);
    else if (k == "scircuit" || k == "webbase" || k == "mac_econ") {
        // fitted against SURVEY.md 8 (scripts/calibrate_standins.py; asserted by tests/test_host.py)
        CommunityParams p{};
        uint32_t n0;
        if (k == "scircuit") {  // structurally symmetric, diagonal present, tight communities
            n0 = 170998;
            p.mean_len = 2.6934, p.kmin = 2, p.cap = 471, p.block = 27, p.p_local = 0.98882, p.skew = 1.93942;
            p.p_pref = 0.010936, p.hubs = 256, p.hub_skew = 1.0, p.symmetric = 1;
        } else if (k == "webbase") {  // power-law rows, 2/3 leaf pages, dense overlapping hubs per site
            n0 = 1000005;
            p.mean_len = 3.5334, p.kmin = 1, p.cap = 5446, p.block = 2048, p.p_local = 0.7959, p.skew = 1.0;
            p.p_pref = 0.0058, p.hubs = 4096, p.hub_skew = 1.0, p.skew_from = 8, p.density = 0.7888, p.tmpl = 4;
            p.sort_block = 1;
        } else {  // mac_econ: short rows (max ~44), references biased to short rows
            n0 = 206500;
            p.mean_len = 6.5418, p.kmin = 3, p.cap = 48, p.block = 256, p.p_local = 0.8078, p.skew = 2.9005;
            p.p_short = 0.2971, p.short_cut = 6, p.hubs = 1, p.hub_skew = 1.0;
        }
        gen_community(*m, scaled(n0), seed, sg, p);
    }
    else if (k == "cant") gen_cant(*m, scaled(62451), seed, sg);
    else if (k == "nlpkkt") gen_stencil27(*m, (uint32_t)std::max(2.0, std::floor(203.0 * std::cbrt(scale) + 0.5)), seed, sg);
    else {
        delete m;
        return SPECK_ERR_INVALID;
    }
    *out = m;
    return SPECK_OK;
}

int speck_load_mtx(const char* path, speck_host_csr** out)
{
    if (!path || !out) return SPECK_ERR_INVALID;
    auto* m = new speck_host_csr();
    int rc;
    try {
        rc = load_mtx(path, *m);
    } catch (const std::bad_alloc&) {
        rc = SPECK_ERR_OOM;
    } catch (...) {
        rc = SPECK_ERR_IO;
    }
    if (rc != SPECK_OK) {
        delete m;
        return rc;
    }
    *out = m;
    return SPECK_OK;
}

int speck_load_hicsr(const char* path, speck_host_csr** out)
{
    if (!path || !out) return SPECK_ERR_INVALID;
    auto* m = new speck_host_csr();
    int rc;
    try {
        rc = load_hicsr(path, *m);
    } catch (const std::bad_alloc&) {
        rc = SPECK_ERR_OOM;
    } catch (...) {
        rc = SPECK_ERR_IO;
    }
    if (rc != SPECK_OK) {
        delete m;
        return rc;
    }
    *out = m;
    return SPECK_OK;
}

// MatrixMarket writer (the reference has none; used to stage test inputs and to export stand-ins): coordinate
// real, `general`, or `symmetric` with the LOWER triangle only (what SuiteSparse ships; the caller vouches for the
// symmetry -- entries above the diagonal are simply not written).  Values with 17 significant digits: exact.
int speck_store_mtx(const speck_host_csr* m, const char* path, int symmetric_lower)
{
    if (!m || !path) return SPECK_ERR_INVALID;
    FILE* f = std::fopen(path, "w");
    if (!f) return SPECK_ERR_IO;
    std::vector<char> buf(1 << 22);
    std::setvbuf(f, buf.data(), _IOFBF, buf.size());
    uint64_t n = 0;
    if (symmetric_lower) {
        for (uint64_t r = 0; r < m->rows; ++r)
            for (uint32_t e = m->row_offsets[r]; e < m->row_offsets[r + 1]; ++e) n += m->col_ids[e] <= r;
    } else
        n = m->col_ids.size();
    std::fprintf(f, "%%%%MatrixMarket matrix coordinate real %s\n%llu %llu %llu\n", symmetric_lower ? "symmetric" : "general",
                 (unsigned long long)m->rows, (unsigned long long)m->cols, (unsigned long long)n);
    char line[96];
    for (uint64_t r = 0; r < m->rows; ++r)
        for (uint32_t e = m->row_offsets[r]; e < m->row_offsets[r + 1]; ++e) {
            if (symmetric_lower && m->col_ids[e] > r) continue;
            char* q = line;
            q = std::to_chars(q, line + 32, r + 1).ptr;
            *q++ = ' ';
            q = std::to_chars(q, line + 64, (uint64_t)m->col_ids[e] + 1).ptr;
            *q++ = ' ';
            q += std::snprintf(q, 30, "%.17g", m->data[e]);
            *q++ = '\n';
            std::fwrite(line, 1, size_t(q - line), f);
        }
    const bool ok = std::ferror(f) == 0;
    return (std::fclose(f) == 0 && ok) ? SPECK_OK : SPECK_ERR_IO;
}

int speck_store_hicsr(const speck_host_csr* m, const char* path)
{
    if (!m || !path) return SPECK_ERR_INVALID;
    return store_hicsr(*m, path);
}

int speck_load_matrix(const char* path, int write_cache, speck_host_csr** out)
{
    if (!path || !out) return SPECK_ERR_INVALID;
    const std::string cache = std::string(path) + "d_" + ".hicsr";  // DataLoader.cpp:9-26
    if (speck_load_hicsr(cache.c_str(), out) == SPECK_OK) return SPECK_OK;
    int rc = speck_load_mtx(path, out);
    if (rc != SPECK_OK) return rc;
    if (write_cache) (void)store_hicsr(**out, cache.c_str());  // failure to cache is not fatal
    return SPECK_OK;
}

int speck_host_csr_dims(const speck_host_csr* m, uint64_t* rows, uint64_t* cols, uint64_t* nnz)
{
    if (!m) return SPECK_ERR_INVALID;
    if (rows) *rows = m->rows;
    if (cols) *cols = m->cols;
    if (nnz) *nnz = m->col_ids.size();
    return SPECK_OK;
}

int speck_host_csr_copy(const speck_host_csr* m, uint32_t* row_offsets, uint32_t* col_ids, double* data)
{
    if (!m) return SPECK_ERR_INVALID;
    if (row_offsets) std::memcpy(row_offsets, m->row_offsets.data(), m->row_offsets.size() * 4);
    if (col_ids) std::memcpy(col_ids, m->col_ids.data(), m->col_ids.size() * 4);
    if (data) std::memcpy(data, m->data.data(), m->data.size() * 8);
    return SPECK_OK;
}

int speck_host_csr_from_arrays(uint64_t rows, uint64_t cols, uint64_t nnz, const uint32_t* row_offsets,
                               const uint32_t* col_ids, const double* data, speck_host_csr** out)
{
    if (!out || !row_offsets || (nnz && (!col_ids || !data))) return SPECK_ERR_INVALID;
    auto* m = new speck_host_csr();
    m->rows = rows;
    m->cols = cols;
    m->row_offsets.assign(row_offsets, row_offsets + rows + 1);
    m->col_ids.assign(col_ids, col_ids + nnz);
    m->data.assign(data, data + nnz);
    *out = m;
    return SPECK_OK;
}

int speck_host_csr_free(speck_host_csr* m)
{
    delete m;
    return SPECK_OK;
}

}  // extern "C"
