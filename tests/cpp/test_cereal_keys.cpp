// test_cereal_keys.cpp -- keys and ciphertexts travel through include/cufhe_amd_cereal.hpp into the engine:
//   raw key words (argv[1], argv[2]) -> SaveEvalKey -> LoadEvalKey -> cufhe::Initialize -> cufhe::Nand on the ciphertexts of
//   a std::vector<TLWE<lvl0param>> archive (argv[3], argv[4]) -> SaveTLWEVector (argv[5]).
// The caller (tests/test_gpu_parity.py) compares the output archive with the oracle's words.  This keeps the file-format
// code on a path that reaches the GPU; it does not claim parity with TFHEpp-produced files (none exists here).
#include <cstdio>
#include <fstream>
#include <vector>

#include "../../include/cufhe_amd.hpp"
#include "../../include/cufhe_amd_cereal.hpp"

using namespace cufhe;
using namespace cufhe::cereal_io;
using P = TFHEpp::lvl0param;

static std::vector<uint32_t> raw(const char* path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    std::vector<uint32_t> v((size_t)f.tellg() / 4);
    f.seekg(0);
    f.read((char*)v.data(), (std::streamsize)(v.size() * 4));
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 7) return 2;
    const KeyShape shape{630, 1024, 1, 3, 8, 2};
    const std::string ek_path = argv[6];
    {
        const std::vector<uint32_t> bk = raw(argv[1]), ksk = raw(argv[2]);
        std::vector<uint8_t> header(53);                      // a header of a size the reader has to find by itself
        for (size_t i = 0; i < header.size(); i++) header[i] = (uint8_t)(7 + 3 * i);
        SaveEvalKey(ek_path, shape, bk, ksk, header);
    }
    std::vector<uint32_t> bk, ksk;
    const EvalKeyFound found = LoadEvalKey(ek_path, shape, bk, ksk);
    std::printf("header %llu members %zu bk_member %d ksk_member %d\n", (unsigned long long)found.header_bytes, found.member_bytes.size(),
                found.bk_member, found.ksk_member);
    SetGPUNum(1);
    Initialize(bk.data(), bk.size(), ksk.data(), ksk.size());
    std::vector<uint32_t> a, b;
    size_t n;
    {
        std::ifstream fa(argv[3], std::ios::binary), fb(argv[4], std::ios::binary);
        PortableBinaryReader ra(fa), rb(fb);
        n = LoadTLWEVector(ra, a, P::n + 1);
        if (LoadTLWEVector(rb, b, P::n + 1) != n) return 3;
    }
    std::vector<Ctxt<P>> ca(n), cb(n), co(n);
    Stream st;
    st.Create();
    for (size_t i = 0; i < n; i++) {
        std::copy(a.begin() + i * (P::n + 1), a.begin() + (i + 1) * (P::n + 1), ca[i].tlwehost.begin());
        std::copy(b.begin() + i * (P::n + 1), b.begin() + (i + 1) * (P::n + 1), cb[i].tlwehost.begin());
        Nand(co[i], ca[i], cb[i], st);
    }
    Synchronize();
    std::vector<uint32_t> out(n * (P::n + 1));
    for (size_t i = 0; i < n; i++) std::copy(co[i].tlwehost.begin(), co[i].tlwehost.end(), out.begin() + i * (P::n + 1));
    {
        std::ofstream fo(argv[5], std::ios::binary);
        PortableBinaryWriter w(fo);
        SaveTLWEVector(w, out, P::n + 1);
    }
    st.Destroy();
    CleanUp();
    std::printf("ok %zu\n", n);
    return 0;
}
