// oracle/ref_formats_driver.cpp -- thin command-line driver around the REFERENCE's own
// host-side C++ (source/CSR.cpp, source/COO.cpp, compiled where they lie under
// /root/reference by oracle/Makefile; outputs go to oracle/_ref/ only).
// TEST INFRASTRUCTURE ONLY: used to pin the on-disk formats (MatrixMarket
// reader semantics, COO->CSR ordering, .hicsr bytes) that speck_amd's loader
// must reproduce.  No reference source is copied into this repository.
//
//   ref_formats mtx2hicsr <in.mtx> <out.hicsr>   loadMTX -> convert -> storeCSR
//   ref_formats dump      <in.hicsr>             loadCSR -> text dump on stdout
#include "CSR.h"
#include "COO.h"
#include <cstdio>
#include <cstring>
#include <exception>
#include <iostream>

int main(int argc, char** argv)
{
    try {
        if (argc == 4 && !std::strcmp(argv[1], "mtx2hicsr")) {
            COO<double> coo = loadMTX<double>(argv[2]);
            CSR<double> csr;
            // the reference's convert() prints nnz on stdout; keep stdout clean for callers
            std::streambuf* old = std::cout.rdbuf(std::cerr.rdbuf());
            convert(csr, coo);
            std::cout.rdbuf(old);
            storeCSR(csr, argv[3]);
            std::printf("%zu %zu %zu\n", csr.rows, csr.cols, csr.nnz);
            return 0;
        }
        if (argc == 3 && !std::strcmp(argv[1], "dump")) {
            CSR<double> csr = loadCSR<double>(argv[2]);
            std::printf("%zu %zu %zu\n", csr.rows, csr.cols, csr.nnz);
            for (size_t i = 0; i <= csr.rows; ++i) std::printf("%u ", csr.row_offsets[i]);
            std::printf("\n");
            for (size_t i = 0; i < csr.nnz; ++i) std::printf("%u ", csr.col_ids[i]);
            std::printf("\n");
            for (size_t i = 0; i < csr.nnz; ++i) std::printf("%.17g ", csr.data[i]);
            std::printf("\n");
            return 0;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
    std::fprintf(stderr, "usage: ref_formats mtx2hicsr in.mtx out.hicsr | dump in.hicsr\n");
    return 1;
}
