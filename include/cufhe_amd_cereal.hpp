// cufhe_amd_cereal.hpp -- reading (and writing) TFHEpp key / ciphertext files.
//
// The reference's manual says "alternatively, write / read key files" (README.md:53) and links cereal
// (src/CMakeLists.txt:17); TFHEpp serialises with cereal::PortableBinary{Output,Input}Archive.  Neither
// TFHEpp nor cereal is in the reference tree, so this header restates the archive ENCODING
// (public cereal format) and the CONTAINER shapes TFHEpp's types have as far as the reference shows them:
//     TLWE<P>                = std::array<T, k n + 1>                 include/cufhe_gpu.cuh:118
//     BootstrappingKey<P>    = std::array<TRGSW<targetP>, domainP::n> src/bootstrap_gpu.cu:111-138 (T[n][(k+1)l][k+1][N])
//     KeySwitchingKey<P>     = T[k N][t][2^(basebit-1)][n + 1]        include/keyswitch_gpu.cuh:123-126
//     EvalKey                = optional (std::unique_ptr) members     src/cufhe_gates_gpu.cu:45-46 (getbk / getiksk)
//
// STATUS: UNVERIFIED AGAINST A TFHEpp-PRODUCED FILE.  No such file exists in this environment and none can
// be made (TFHEpp is absent); what tests/test_cereal.py verifies is the archive encoding against
// hand-assembled byte strings and round trips through the writer below.  Do not read parity with TFHEpp
// into it.  The EvalKey member ORDER differs between TFHEpp versions; LoadEvalKey therefore does not assume
// one: it looks for the unique way to read the file as "header, then optional members whose payload sizes
// are the sizes TFHEpp's key types have for this parameter set", and fails loudly if there is none or more
// than one.
//
// cereal portable binary encoding (cereal/archives/portable_binary.hpp, cereal/types/{array,vector,memory}.hpp):
//     stream  := uint8 is_little_endian, then values
//     arithmetic T            -> sizeof(T) raw bytes (byte-swapped when the flag differs from the host)
//     std::array<arith, N>    -> N * sizeof raw bytes, no length
//     std::array<other, N>    -> the N elements in order
//     std::vector<arith>      -> uint64 count, then raw bytes
//     std::vector<other>      -> uint64 count, then the elements
//     std::unique_ptr<T>      -> uint8 valid; if valid == 1 the T
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

namespace cufhe {
namespace cereal_io {

inline bool host_is_little_endian()
{
    const uint16_t x = 1;
    uint8_t b;
    std::memcpy(&b, &x, 1);
    return b == 1;
}

class PortableBinaryReader {
   public:
    explicit PortableBinaryReader(std::istream& in) : in_(in)
    {
        uint8_t flag = 0;
        in_.read((char*)&flag, 1);
        if (!in_ || flag > 1) throw std::runtime_error("cereal portable binary: bad endianness flag");
        swap_ = (flag == 1) != host_is_little_endian();
    }
    template <class T>
    T scalar()
    {
        T v;
        raw(&v, 1, sizeof(T));
        return v;
    }
    /// `count` elements of `elem` bytes each (byte-swapped per element if the file's endianness differs)
    void raw(void* dst, size_t count, size_t elem)
    {
        in_.read((char*)dst, (std::streamsize)(count * elem));
        if (!in_) throw std::runtime_error("cereal portable binary: truncated");
        if (swap_ && elem > 1) {
            uint8_t* p = (uint8_t*)dst;
            for (size_t i = 0; i < count; i++)
                for (size_t a = 0, b = elem - 1; a < b; a++, b--) std::swap(p[i * elem + a], p[i * elem + b]);
        }
    }
    uint64_t size_tag() { return scalar<uint64_t>(); }
    bool unique_ptr_valid()
    {
        const uint8_t v = scalar<uint8_t>();
        if (v > 1) throw std::runtime_error("cereal portable binary: bad unique_ptr flag");
        return v == 1;
    }
    void skip(uint64_t bytes)
    {
        in_.seekg((std::streamoff)bytes, std::ios::cur);
        if (!in_) throw std::runtime_error("cereal portable binary: truncated");
    }
    template <class T>
    void array(T* dst, size_t n) { raw(dst, n, sizeof(T)); }
    template <class T>
    void vector(std::vector<T>& v)
    {
        const uint64_t n = size_tag();
        if (n > (1ull << 34) / sizeof(T)) throw std::runtime_error("cereal portable binary: implausible vector length");
        v.resize((size_t)n);
        raw(v.data(), v.size(), sizeof(T));
    }
    bool swapping() const { return swap_; }

   private:
    std::istream& in_;
    bool swap_;
};

class PortableBinaryWriter {     // native endianness, like cereal's default options
   public:
    explicit PortableBinaryWriter(std::ostream& out) : out_(out)
    {
        const uint8_t flag = host_is_little_endian() ? 1 : 0;
        out_.write((const char*)&flag, 1);
    }
    template <class T>
    void scalar(T v) { out_.write((const char*)&v, sizeof(T)); }
    template <class T>
    void array(const T* src, size_t n) { out_.write((const char*)src, (std::streamsize)(n * sizeof(T))); }
    void size_tag(uint64_t n) { scalar<uint64_t>(n); }
    void unique_ptr_valid(bool v) { scalar<uint8_t>(v ? 1 : 0); }
    template <class T>
    void vector(const std::vector<T>& v)
    {
        size_tag(v.size());
        array(v.data(), v.size());
    }

   private:
    std::ostream& out_;
};

// ---- ciphertext files: a TLWE<P> or a std::vector<TLWE<P>> of level-`words` ciphertexts ----
template <class T>
inline void LoadTLWE(PortableBinaryReader& ar, T* tlwe, size_t words) { ar.array(tlwe, words); }
template <class T>
inline void SaveTLWE(PortableBinaryWriter& ar, const T* tlwe, size_t words) { ar.array(tlwe, words); }
/// std::vector<TLWE<P>>: count, then the arrays back to back; returns the count
template <class T>
inline size_t LoadTLWEVector(PortableBinaryReader& ar, std::vector<T>& flat, size_t words)
{
    const uint64_t n = ar.size_tag();
    if (n > (1ull << 32)) throw std::runtime_error("ciphertext file: implausible count");
    flat.resize((size_t)n * words);
    ar.array(flat.data(), flat.size());
    return (size_t)n;
}
template <class T>
inline void SaveTLWEVector(PortableBinaryWriter& ar, const std::vector<T>& flat, size_t words)
{
    ar.size_tag(flat.size() / words);
    ar.array(flat.data(), flat.size());
}

// ---- key files ----
struct KeyShape {      // sizes of the set the library was built for (cufhe_amd_get_params / cufhe_amd_ps_get_params)
    uint64_t n, N, k, l, t, basebit;
    uint64_t bk_bytes() const { return n * (k + 1) * l * (k + 1) * N * 4; }                 // BootstrappingKey<lvl01>: torus32
    uint64_t ksk_bytes() const { return k * N * t * (1ull << (basebit - 1)) * (n + 1) * 4; }  // KeySwitchingKey<lvl10>
    uint64_t bkfft_bytes() const { return n * (k + 1) * l * (k + 1) * N * 8; }              // BootstrappingKeyFFT: doubles
    uint64_t bkntt_bytes() const { return n * (k + 1) * l * (k + 1) * N * 8; }              // BootstrappingKeyNTT: 64-bit residues
};

/// BootstrappingKey<lvl01param> / KeySwitchingKey<lvl10param> stored on their own (ar(*ek.bklvl01) style files)
inline void LoadBootstrappingKey(PortableBinaryReader& ar, const KeyShape& s, std::vector<uint32_t>& bk)
{
    bk.resize(s.bk_bytes() / 4);
    ar.array(bk.data(), bk.size());
}
inline void LoadKeySwitchingKey(PortableBinaryReader& ar, const KeyShape& s, std::vector<uint32_t>& ksk)
{
    ksk.resize(s.ksk_bytes() / 4);
    ar.array(ksk.data(), ksk.size());
}

struct EvalKeyFound {
    uint64_t header_bytes = 0;                 // bytes between the endianness flag and the first member (lweParams)
    std::vector<uint64_t> member_bytes;        // payload of every member, 0 for an empty unique_ptr
    int bk_member = -1, ksk_member = -1;
    uint64_t bk_pos = 0, ksk_pos = 0;          // file offsets of the two payloads
};

/// Read an EvalKey archive: header (TFHEpp's lweParams, of a size this function determines), then optional members.
/// `extra_sizes`: payload sizes of member types other than those derivable from `shape` (e.g. the lvl02 / lvl2x keys
/// of a TFHEpp build that generated them).  Throws unless the file has exactly ONE consistent reading containing
/// exactly one member of the raw lvl01 bootstrapping key's size and one of the lvl10 key-switching key's size.
inline EvalKeyFound LoadEvalKey(const std::string& path, const KeyShape& shape, std::vector<uint32_t>& bk,
                                std::vector<uint32_t>& ksk, const std::vector<uint64_t>& extra_sizes = {},
                                uint64_t max_header = 4096)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    const uint64_t total = (uint64_t)f.tellg();
    f.seekg(0);
    PortableBinaryReader ar(f);
    std::vector<uint64_t> sizes = {shape.bk_bytes(), shape.ksk_bytes(), shape.bkfft_bytes()};
    if (shape.bkntt_bytes() != shape.bkfft_bytes()) sizes.push_back(shape.bkntt_bytes());
    for (uint64_t e : extra_sizes) sizes.push_back(e);
    // the flag bytes of the members: read them sparsely (a member's flag sits right behind the previous payload)
    auto byte_at = [&](uint64_t pos) -> int {
        f.clear();
        f.seekg((std::streamoff)pos);
        char c;
        f.read(&c, 1);
        return f ? (uint8_t)c : -1;
    };
    std::vector<EvalKeyFound> readings;
    for (uint64_t h = 0; h <= max_header && readings.size() < 2; h++) {
        // depth-first over "flag 0" | "flag 1 + one of the sizes"; the branching is tiny (sizes are far apart)
        struct Frame { uint64_t pos; std::vector<uint64_t> members; };
        std::vector<Frame> stack;
        stack.push_back(Frame{1 + h, {}});
        while (!stack.empty() && readings.size() < 2) {
            Frame fr = stack.back();
            stack.pop_back();
            if (fr.pos == total) {
                int nbk = 0, nksk = 0;
                EvalKeyFound r;
                r.header_bytes = h;
                r.member_bytes = fr.members;
                uint64_t at = 1 + h;
                for (size_t i = 0; i < fr.members.size(); i++) {
                    at += 1;
                    if (fr.members[i] == shape.bk_bytes()) { nbk++; r.bk_member = (int)i; r.bk_pos = at; }
                    if (fr.members[i] == shape.ksk_bytes()) { nksk++; r.ksk_member = (int)i; r.ksk_pos = at; }
                    at += fr.members[i];
                }
                // zero bytes at the end of the header also read as empty members: such readings place the two
                // keys at the same offsets and are one reading
                bool dup = false;
                for (const EvalKeyFound& o : readings) dup = dup || (o.bk_pos == r.bk_pos && o.ksk_pos == r.ksk_pos);
                if (nbk == 1 && nksk == 1 && !dup) readings.push_back(r);
                continue;
            }
            if (fr.pos > total || fr.members.size() > 64) continue;
            const int flag = byte_at(fr.pos);
            if (flag == 0) {
                Frame nx = fr;
                nx.pos += 1;
                nx.members.push_back(0);
                stack.push_back(nx);
            } else if (flag == 1) {
                for (uint64_t sz : sizes) {
                    Frame nx = fr;
                    nx.pos += 1 + sz;
                    nx.members.push_back(sz);
                    stack.push_back(nx);
                }
            }
        }
    }
    if (readings.empty()) throw std::runtime_error("EvalKey archive: no consistent reading for this parameter set (different TFHEpp version or parameters?)");
    if (readings.size() > 1) throw std::runtime_error("EvalKey archive: ambiguous (more than one consistent reading); load the keys from separate files instead");
    const EvalKeyFound& r = readings[0];
    f.clear();
    f.seekg((std::streamoff)r.bk_pos);
    bk.resize(shape.bk_bytes() / 4);
    ar.array(bk.data(), bk.size());
    f.clear();
    f.seekg((std::streamoff)r.ksk_pos);
    ksk.resize(shape.ksk_bytes() / 4);
    ar.array(ksk.data(), ksk.size());
    return r;
}

/// Write an EvalKey-shaped archive: endianness flag, `header` (stands for TFHEpp's lweParams block), then the optional
/// members in the order {bkfft: empty, bk, bkntt: empty, ksk}: cereal's encoding of std::unique_ptr members (a validity
/// byte, then the payload).  The counterpart of LoadEvalKey for round trips and for handing keys to other processes; like
/// the reader it is UNVERIFIED against a TFHEpp-produced file.
inline void SaveEvalKey(const std::string& path, const KeyShape& shape, const std::vector<uint32_t>& bk,
                        const std::vector<uint32_t>& ksk, const std::vector<uint8_t>& header = {})
{
    if (bk.size() * 4 != shape.bk_bytes() || ksk.size() * 4 != shape.ksk_bytes()) throw std::runtime_error("SaveEvalKey: key sizes do not fit the shape");
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot create " + path);
    PortableBinaryWriter ar(f);
    ar.array(header.data(), header.size());
    ar.unique_ptr_valid(false);                 // bkfftlvl01: not generated
    ar.unique_ptr_valid(true);
    ar.array(bk.data(), bk.size());             // bklvl01
    ar.unique_ptr_valid(false);                 // bknttlvl01: not generated
    ar.unique_ptr_valid(true);
    ar.array(ksk.data(), ksk.size());           // iksklvl10
    if (!f) throw std::runtime_error("write error on " + path);
}

}  // namespace cereal_io
}  // namespace cufhe
