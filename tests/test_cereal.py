"""include/cufhe_amd_cereal.hpp: TFHEpp key / ciphertext files (cereal portable binary archives).

UNVERIFIED against a TFHEpp-produced file (TFHEpp and cereal are absent from the reference tree):
what is checked here is the archive ENCODING against byte strings assembled by hand from cereal's published
format, and that an EvalKey-shaped file -- header of unknown size, optional members in an order this code does
not assume -- is read back to the same key words, or refused when it is ambiguous or of another parameter set."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r'''
#include <cstdio>
#include <sstream>
#include "cufhe_amd_cereal.hpp"
using namespace cufhe::cereal_io;
int main(int argc, char** argv) {
    const std::string mode = argv[1];
    if (mode == "scalars") {          // file made by the test: flag, u32, i64, double, vector<u16>, unique_ptr{0}, unique_ptr{1, u8}
        std::ifstream f(argv[2], std::ios::binary);
        PortableBinaryReader ar(f);
        const uint32_t a = ar.scalar<uint32_t>(); const int64_t b = ar.scalar<int64_t>(); const double c = ar.scalar<double>();
        std::vector<uint16_t> v; ar.vector(v);
        const bool p0 = ar.unique_ptr_valid(); const bool p1 = ar.unique_ptr_valid(); const uint8_t q = ar.scalar<uint8_t>();
        std::printf("%u %lld %.3f %zu %u %u %d %d %u swap=%d\n", a, (long long)b, c, v.size(), v[0], v[2], p0, p1, q, ar.swapping());
        return 0;
    }
    if (mode == "tlwe") {             // std::vector<TLWE<lvl0>>: count + raw arrays; write it back through the writer
        std::ifstream f(argv[2], std::ios::binary);
        PortableBinaryReader ar(f);
        std::vector<uint32_t> flat;
        const size_t n = LoadTLWEVector(ar, flat, 631);
        std::ofstream o(argv[3], std::ios::binary);
        PortableBinaryWriter w(o);
        SaveTLWEVector(w, flat, 631);
        std::printf("%zu\n", n);
        return 0;
    }
    if (mode == "evalkey") {
        KeyShape s{(uint64_t)atoi(argv[3]), (uint64_t)atoi(argv[4]), 1, (uint64_t)atoi(argv[5]), 8, 2};
        std::vector<uint32_t> bk, ksk;
        try {
            EvalKeyFound r = LoadEvalKey(argv[2], s, bk, ksk, {777}, 64);   // header bound below the toy payload sizes
            unsigned long long h1 = 0, h2 = 0;
            for (uint32_t w : bk) h1 = h1 * 1099511628211ull + w;
            for (uint32_t w : ksk) h2 = h2 * 1099511628211ull + w;
            std::printf("ok %llu %zu %zu %llu %llu\n", (unsigned long long)r.header_bytes, bk.size(), ksk.size(), h1, h2);
        } catch (const std::exception& e) { std::printf("error %s\n", e.what()); }
        return 0;
    }
    return 2;
}
'''


def _build(tmp_path):
    src = tmp_path / "cereal_prog.cpp"
    src.write_text(PROG)
    exe = tmp_path / "cereal_prog"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    return str(exe)


def _run(exe, *args):
    return subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=120).stdout.strip()


def _fnv(words):
    h = 0
    for w in words:
        h = (h * 1099511628211 + int(w)) % (1 << 64)
    return h


def test_archive_encoding(tmp_path):
    exe = _build(tmp_path)
    body = struct.pack("<IqdQ3HBBB", 0xDEADBEEF, -5, 2.5, 3, 7, 8, 9, 0, 1, 42)
    (tmp_path / "le.bin").write_bytes(b"\x01" + body)
    assert _run(exe, "scalars", tmp_path / "le.bin") == "3735928559 -5 2.500 3 7 9 0 1 42 swap=0"
    # the same values from a big-endian writer: flag 0, every value byte-swapped
    be = struct.pack(">IqdQ3HBBB", 0xDEADBEEF, -5, 2.5, 3, 7, 8, 9, 0, 1, 42)
    (tmp_path / "be.bin").write_bytes(b"\x00" + be)
    assert _run(exe, "scalars", tmp_path / "be.bin") == "3735928559 -5 2.500 3 7 9 0 1 42 swap=1"


def test_ciphertext_vector_round_trip(tmp_path):
    exe = _build(tmp_path)
    rng = np.random.default_rng(1)
    cts = rng.integers(0, 2**32, size=(5, 631), dtype=np.uint64).astype(np.uint32)
    (tmp_path / "c.bin").write_bytes(b"\x01" + struct.pack("<Q", 5) + cts.tobytes())
    assert _run(exe, "tlwe", tmp_path / "c.bin", tmp_path / "c2.bin") == "5"
    assert (tmp_path / "c2.bin").read_bytes() == (tmp_path / "c.bin").read_bytes()


def test_evalkey_shaped_file(tmp_path):
    """A small parameter set (n = 6, N = 16, l = 2) keeps the file tiny; the member order is deliberately not the
    order of any particular TFHEpp version: header, empty, bkfft, empty, ksk, 'other' member, bk, empty."""
    exe = _build(tmp_path)
    n, N, l = 6, 16, 2
    rng = np.random.default_rng(2)
    bk = rng.integers(2, 2**32, size=n * 2 * l * 2 * N, dtype=np.uint64).astype(np.uint32)
    ksk = rng.integers(2, 2**32, size=N * 8 * 2 * (n + 1), dtype=np.uint64).astype(np.uint32)
    bkfft = rng.integers(2, 255, size=n * 2 * l * 2 * N * 8, dtype=np.uint8)
    header = bytes(rng.integers(2, 255, size=37, dtype=np.uint8)) + b"\x00\x00"       # ends in zero bytes
    body = (b"\x00" + b"\x01" + bkfft.tobytes() + b"\x00" + b"\x01" + ksk.tobytes() +
            b"\x01" + bytes(rng.integers(2, 255, size=777, dtype=np.uint8)) + b"\x01" + bk.tobytes() + b"\x00")
    (tmp_path / "ek.bin").write_bytes(b"\x01" + header + body)
    out = _run(exe, "evalkey", tmp_path / "ek.bin", n, N, l).split()
    assert out[0] == "ok" and int(out[2]) == bk.size and int(out[3]) == ksk.size, out
    assert int(out[4]) == _fnv(bk) and int(out[5]) == _fnv(ksk)
    # another parameter set: no consistent reading, refused
    assert _run(exe, "evalkey", tmp_path / "ek.bin", n + 1, N, l).startswith("error")
    # a file with two members of the bootstrapping key's size: ambiguous, refused
    (tmp_path / "two.bin").write_bytes(b"\x01" + header + b"\x01" + bk.tobytes() + b"\x01" + bk.tobytes() + b"\x01" + ksk.tobytes())
    assert _run(exe, "evalkey", tmp_path / "two.bin", n, N, l).startswith("error")
