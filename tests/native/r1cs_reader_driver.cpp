// tests/native/r1cs_reader_driver.cpp — the C++ `.r1cs` reader of include/dock_gpu.hpp (dock_gpu::circom::R1CSFile) on the files given on the command
// line: one JSON line per file with the header fields and an FNV-1a hash of each CSR matrix (rowptr, cols, vals), or the reader's error message.
// tests/test_r1cs_reader_cpp.py compares the lines with crypto_amd/r1cs_file.py's view of the same files (the reference's own fixtures).  No device call.
#include <cstdio>
#include <fstream>
#include <iterator>
#include "../../include/dock_gpu.hpp"
static uint64_t fnv(uint64_t h, const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ULL; } return h; }
static uint64_t hash(const dock_gpu::circom::Csr &m) {
    uint64_t h = 0xcbf29ce484222325ULL;
    h = fnv(h, m.rowptr.data(), m.rowptr.size() * 8); h = fnv(h, m.cols.data(), m.cols.size() * 4); h = fnv(h, m.vals.data(), m.vals.size() * 8);
    return h;
}
int main(int argc, char **argv) {
    for (int k = 1; k < argc; k++) {
        std::ifstream in(argv[k], std::ios::binary);
        std::vector<uint8_t> d((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        try {
            const auto f = dock_gpu::circom::R1CSFile::parse(d.data(), d.size());
            std::printf("{\"n_wires\": %u, \"n_pub_out\": %u, \"n_pub_in\": %u, \"n_prv_in\": %u, \"n_constraints\": %u, \"n_labels\": %llu, \"num_inputs\": %zu, \"bls12_381\": %s, "
                        "\"nnz\": [%zu, %zu, %zu], \"hash\": [\"%016llx\", \"%016llx\", \"%016llx\"], \"last_label\": %llu}\n",
                        f.n_wires, f.n_pub_out, f.n_pub_in, f.n_prv_in, f.n_constraints, (unsigned long long)f.n_labels, f.num_inputs(), f.is_bls12_381() ? "true" : "false",
                        f.a.cols.size(), f.b.cols.size(), f.c.cols.size(), (unsigned long long)hash(f.a), (unsigned long long)hash(f.b), (unsigned long long)hash(f.c),
                        (unsigned long long)(f.wire_mapping.empty() ? 0 : f.wire_mapping.back()));
        } catch (const std::exception &e) { std::printf("{\"error\": \"%s\"}\n", e.what()); }
    }
    return 0;
}
