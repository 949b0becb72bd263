// convert(dCSR&, const dCSR&, padding) -- reference source/dCSR.cpp:81-89 -- is device to device: K conversions (of a
// matrix and of a row-range VIEW of it, with padding) issue no device-to-host copy; the ONE download at the end, which
// checks the contents, is the only one (tests/test_gpu_driver.py runs this under rocprofv3 --memory-copy-trace).
// Also: the call without the timings argument the reference's default `Timings()` permits (include/Multiply.h:19).
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tests/cpp/convert_d2d.cpp \
//       -Lspeck_amd -lspeck_amd -L/opt/rocm/lib -lamdhip64
#include <cstdio>
#include <cstdlib>

#include "CSR.h"
#include "Multiply.h"

int main(int argc, char** argv)
{
    const int K = argc > 1 ? std::atoi(argv[1]) : 5;
    const unsigned rows = 1000, per = 7, cols = 5000, pad = 3;
    CSR<double> a;
    a.alloc(rows, cols, rows * per);
    for (unsigned r = 0; r <= rows; ++r) a.row_offsets[r] = r * per;
    for (unsigned r = 0; r < rows; ++r)
        for (unsigned j = 0; j < per; ++j) {
            a.col_ids[r * per + j] = (r * 13 + j * 601) % (cols / per) + j * (cols / per);  // ascending inside the row
            a.data[r * per + j] = 0.25 * (r % 17) + j;
        }
    dCSR<double> dA;
    convert(dA, a, pad);  // host -> device, padded
    if (dA.rows != rows || dA.nnz != rows * per || !dA.row_offsets) return 1;
    dCSR<double> dB;
    for (int k = 0; k < K; ++k) {
        const void* before = dB.data;
        convert(dB, dA, pad);  // device -> device
        if (dB.rows != dA.rows || dB.nnz != dA.nnz || dB.cols != dA.cols) return 2;
        if (dB.data == dA.data || dB.col_ids == dA.col_ids || dB.row_offsets == dA.row_offsets) return 3;  // its own buffers
        (void)before;
    }
    // a row-range view of dA (absolute offsets, as the row shards of the multi-GPU path): the copy starts at 0
    const unsigned r0 = 100, r1 = 350;
    dCSR<double> dV;
    {
        dCSRNoDealloc<double> view(dA);
        speck_dcsr v{r1 - r0, view.cols, size_t(r1 - r0) * per, view.data, view.row_offsets + r0, view.col_ids};
        speck_dcsr d = dV.raw();
        if (speck_dcsr_copy(&d, &v, sizeof(double), 0) != SPECK_OK) return 4;
        dV.adopt(d);
    }
    std::printf("conversions done\n");
    CSR<double> b, v;
    convert(b, dB, 0);  // the only device -> host copies of this program (three per matrix)
    convert(v, dV, 0);
    for (unsigned r = 0; r <= rows; ++r)
        if (b.row_offsets[r] != a.row_offsets[r]) return 5;
    for (unsigned i = 0; i < rows * per; ++i)
        if (b.col_ids[i] != a.col_ids[i] || b.data[i] != a.data[i]) return 6;
    for (unsigned r = 0; r <= r1 - r0; ++r)
        if (v.row_offsets[r] != r * per) return 7;
    for (unsigned i = 0; i < (r1 - r0) * per; ++i)
        if (v.col_ids[i] != a.col_ids[r0 * per + i] || v.data[i] != a.data[r0 * per + i]) return 8;
    // MultiplyspECKImplementation without the timings argument
    auto config = spECK::spECKConfig::initialize(0);
    dCSR<double> dAt, dC;
    {
        speck_dcsr s = dA.raw(), t = dAt.raw();
        if (speck_transpose_f64(config.handle, &s, &t) != SPECK_OK) return 9;
        dAt.adopt(t);
    }
    spECK::MultiplyspECKImplementation<double, 4, 1024, spECK_DYNAMIC_MEM_PER_BLOCK, spECK_STATIC_MEM_PER_BLOCK>(dA, dAt, dC, config);
    if (dC.rows != rows || dC.nnz == 0) return 10;
    config.cleanup();
    std::printf("convert d2d ok\n");
    return 0;
}
