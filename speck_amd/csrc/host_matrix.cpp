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
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
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

uint32_t clamp_col(int64_t c, uint32_t n)
{
    if (c < 0) c = -c;
    if (c >= (int64_t)n) c = 2 * (int64_t)n - 2 - c;
    if (c < 0) c = 0;
    return (uint32_t)c;
}

// Row lengths first (so that "preferential" column picks can be proportional to the
// length of the target row: hub rows are hub columns, as in circuit / web graphs).
void gen_powerlaw(speck_host_csr& m, uint32_t n, uint64_t seed, bool signed_values, double mean_len,
                  uint32_t cap, double p_local, int64_t local_halfwidth, double p_pref)
{
    SplitMix64 g(seed);
    PowerLawLen pl(cap, mean_len);
    std::vector<uint32_t> len(n);
    std::vector<uint64_t> pref(n + 1, 0);
    for (uint32_t r = 0; r < n; ++r) {
        len[r] = std::min<uint32_t>(pl.draw(g), n);
        pref[r + 1] = pref[r] + len[r];
    }
    m.rows = m.cols = n;
    m.row_offsets.assign(1, 0);
    std::vector<uint32_t> cand;
    for (uint32_t r = 0; r < n; ++r) {
        for (uint32_t j = 0; j < len[r]; ++j) {
            double u = g.unit();
            uint32_t c;
            if (u < p_local) {
                const int64_t hw = local_halfwidth + 2 * (int64_t)len[r];  // long rows spread wider
                c = clamp_col((int64_t)r + (int64_t)g.below(2 * hw + 1) - hw, n);
            } else if (u < p_local + p_pref) {
                uint64_t t = g.below(pref[n]);
                c = (uint32_t)(std::upper_bound(pref.begin(), pref.end(), t) - pref.begin()) - 1;
            } else {
                c = (uint32_t)g.below(n);
            }
            cand.push_back(c);
        }
        emit_row(m, cand, g, signed_values);
    }
}

// mac_econ_fwd500-like: short rows (mean 6.2, max 44), 80 % in a +-band, 20 % anywhere.
void gen_mac_econ(speck_host_csr& m, uint32_t n, uint64_t seed, bool signed_values)
{
    SplitMix64 g(seed);
    m.rows = m.cols = n;
    m.row_offsets.assign(1, 0);
    std::vector<uint32_t> cand;
    for (uint32_t r = 0; r < n; ++r) {
        // 2 + geometric-ish tail, capped at 44, mean ~6.2
        uint32_t k = 2;
        while (k < 44 && g.unit() < 0.808) ++k;
        for (uint32_t j = 0; j < k; ++j) {
            if (g.unit() < 0.8)
                cand.push_back(clamp_col((int64_t)r + (int64_t)g.below(1025) - 512, n));
            else
                cand.push_back((uint32_t)g.below(n));
        }
        emit_row(m, cand, g, signed_values);
    }
}

// cant-like: 3x3-block FEM on a (nx, ny, nz) node grid, 23 of the 27 neighbours
// (4 of the 8 cube corners are dropped) -> ~64 nnz/row, banded.
void gen_cant(speck_host_csr& m, uint32_t n, uint64_t seed, bool signed_values)
{
    SplitMix64 g(seed);
    const uint32_t nodes = n / 3;
    const uint32_t nx = 14, ny = 14;  // x fastest; half bandwidth ~ 3*(nx*ny + nx + 1) ~ 630
    m.rows = m.cols = n;
    m.row_offsets.assign(1, 0);
    std::vector<uint32_t> cand;
    for (uint32_t r = 0; r < n; ++r) {
        const uint32_t node = std::min(r / 3, nodes ? nodes - 1 : 0);
        const int64_t x = node % nx, y = (node / nx) % ny, z = node / (nx * ny);
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (dz != 0 && dy != 0 && dx != 0 && (dx == dy)) continue;  // drop 4 corners
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
    if (!f.good() || h.typesize != sizeof(double)) return SPECK_ERR_IO;
    m.rows = h.num_rows;
    m.cols = h.num_columns;
    m.data.resize(h.num_non_zeroes);
    m.col_ids.resize(h.num_non_zeroes);
    m.row_offsets.resize(h.num_rows + 1);
    f.read(reinterpret_cast<char*>(m.data.data()), m.data.size() * sizeof(double));
    f.read(reinterpret_cast<char*>(m.col_ids.data()), m.col_ids.size() * sizeof(uint32_t));
    f.read(reinterpret_cast<char*>(m.row_offsets.data()), m.row_offsets.size() * sizeof(uint32_t));
    return f.good() ? SPECK_OK : SPECK_ERR_IO;
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
int load_mtx(const char* path, speck_host_csr& m)
{
    std::ifstream f(path);
    if (!f.is_open()) return SPECK_ERR_IO;
    std::string line;
    if (!std::getline(f, line)) return SPECK_ERR_IO;
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

    uint64_t rows = 0, cols = 0, nnz = 0;
    bool have_size = false;
    while (std::getline(f, line)) {
        if (!line.empty() && line[0] == '%') continue;
        std::istringstream ls(line);
        ls >> rows >> cols >> nnz;
        if (ls.fail()) return SPECK_ERR_IO;
        have_size = true;
        break;
    }
    if (!have_size) return SPECK_ERR_IO;
    struct Entry {
        uint32_t r, c;
        double v;
    };
    std::vector<Entry> e;
    e.reserve(mirror ? nnz * 2 : nnz);
    while (std::getline(f, line)) {
        if (!line.empty() && line[0] == '%') continue;
        size_t p = 0;
        while (p < line.size() && std::isspace((unsigned char)line[p])) ++p;
        if (p == line.size()) continue;
        std::istringstream ls(line);
        uint32_t r, c;
        double d = 1.0;
        ls >> r >> c;
        if (!pattern) ls >> d;
        if (ls.fail()) return SPECK_ERR_IO;
        if (r > rows || c > cols || r == 0 || c == 0) return SPECK_ERR_IO;
        e.push_back({r - 1, c - 1, d});
        if (mirror && r != c) e.push_back({c - 1, r - 1, d});  // no dedup (reference COO.cpp:153-159)
    }
    // reference: std::sort by (r, c) -- the order of exact duplicates is unspecified there;
    // stable_sort picks the file order.
    std::stable_sort(e.begin(), e.end(), [](const Entry& a, const Entry& b) {
        return a.r != b.r ? a.r < b.r : a.c < b.c;
    });
    m.rows = rows;
    m.cols = cols;
    m.row_offsets.assign(rows + 1, 0);
    m.col_ids.resize(e.size());
    m.data.resize(e.size());
    for (size_t i = 0; i < e.size(); ++i) {
        m.col_ids[i] = e[i].c;
        m.data[i] = e[i].v;
        ++m.row_offsets[e[i].r + 1];
    }
    for (uint64_t r = 0; r < rows; ++r) m.row_offsets[r + 1] += m.row_offsets[r];
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
    else if (k == "scircuit") gen_powerlaw(*m, scaled(170998), seed, sg, 6.0, 353, 0.80, 6, 0.06);
    else if (k == "webbase") gen_powerlaw(*m, scaled(1000005), seed, sg, 3.2, 4700, 0.70, 16, 0.09);
    else if (k == "mac_econ") gen_mac_econ(*m, scaled(206500), seed, sg);
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
    int rc = load_mtx(path, *m);
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
    int rc = load_hicsr(path, *m);
    if (rc != SPECK_OK) {
        delete m;
        return rc;
    }
    *out = m;
    return SPECK_OK;
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
