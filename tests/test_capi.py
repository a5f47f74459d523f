"""CPU tests of the C-ABI boundary: libcufhe_amd.so loads and exports exactly what
include/cufhe_amd.h declares; no compute call is made (there is no GPU here), but the
argument checking that does not need one is exercised."""
import ctypes
import os
import re
import subprocess

import pytest

import oracle_lib as ol

HDR = os.path.join(ol.ROOT, "include", "cufhe_amd.h")
LIB = os.path.join(ol.ROOT, "cufhe_amd", "libcufhe_amd.so")


def header_symbols():
    text = open(HDR).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cufhe_amd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build the HIP extension first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(LIB)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/cufhe_amd.h but not exported"


def test_python_binding_covers_the_header():
    from cufhe_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_params_and_sizes_match_the_oracle():
    import cufhe_amd
    p = cufhe_amd.PARAMS
    assert (p.n, p.N, p.nbit, p.k, p.l, p.Bgbit, p.t, p.basebit, p.mu) == (630, 1024, 10, 1, 3, 6, 8, 2, 1 << 29)
    assert (p.lvl0_words, p.lvl1_words) == ol.LVL_WORDS
    assert p.bk_words == ol.BK_WORDS and p.ksk_words == ol.KSK_WORDS
    assert p.bk_ntt_bytes == 61931520            # bytes one blind rotation reads


def test_errors_are_reported_not_fatal():
    import cufhe_amd
    from cufhe_amd import _lib
    lib = _lib.lib
    # wrong key sizes are rejected before any device work
    rc = lib.cufhe_amd_initialize(None, 0, None, 0)
    assert rc < 0 and b"null" in lib.cufhe_amd_last_error()
    buf = (ctypes.c_uint32 * 4)()
    rc = lib.cufhe_amd_initialize(buf, 4, buf, 4)
    assert rc < 0 and b"wrong size" in lib.cufhe_amd_last_error()
    assert lib.cufhe_amd_set_gpu_num(0) < 0
    assert lib.cufhe_amd_set_gpu_num(65) < 0 and b"64 logical devices" in lib.cufhe_amd_last_error()
    # a gate before Initialize(ek) / on a bad device index fails with a status
    assert lib.cufhe_amd_gate(5, None, 0, 0, None, None, None, None) < 0
    with pytest.raises(cufhe_amd.CufheAmdError):
        _lib.check(lib.cufhe_amd_gate(5, None, 0, 0, None, None, None, None))


def test_legacy_manual_api_host_side():
    """include/cufhe_amd_legacy.hpp (the reference README's KeyGen / Encrypt / Decrypt / key-file
    API): plain g++ host code; key generation, an encryption round trip and the key files need
    no GPU (tests/cpp/test_legacy_api.cpp --cpu-only)."""
    import subprocess
    src = os.path.join(ol.ROOT, "tests", "cpp", "test_legacy_api.cpp")
    exe = os.path.join(ol.ROOT, "tests", "cpp", "test_legacy_api")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src,
                           "-L" + os.path.join(ol.ROOT, "cufhe_amd"), "-lcufhe_amd",
                           "-Wl,-rpath," + os.path.join(ol.ROOT, "cufhe_amd")])
    out = subprocess.run([exe, "--cpu-only"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALL PASS" in out.stdout, out.stdout + out.stderr


def test_legacy_keygen_uses_a_csprng(tmp_path):
    """include/cufhe_amd_legacy.hpp draws keys and noise from ChaCha20 keyed by the OS: the block function
    matches the RFC 7539 A.1 vector, two OS-keyed generators differ, and the seeded (test-only) entry point
    does not exist unless CUFHE_AMD_INSECURE_TEST_KEYS is defined."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "kat.cpp"
    src.write_text('''
#include <cstdio>
#include "cufhe_amd_legacy.hpp"
using namespace cufhe::legacy::detail;
int main() {
    ChaCha c; uint8_t k[32] = {0}; c.set_key(k);
    const uint32_t w0 = c.next32(), w1 = c.next32();          // keystream 76 b8 e0 ad a0 f1 3d 90 ...
    ChaCha a, b;
    int diff = 0; for (int i = 0; i < 8; i++) diff += a.next64() != b.next64();
    std::printf("%08x %08x %d\\n", w0, w1, diff);
    return 0;
}''')
    exe = tmp_path / "kat"
    lib = os.path.join(root, "cufhe_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(root, "include"), "-o", str(exe), str(src),
                           "-L" + lib, "-lcufhe_amd", "-Wl,-rpath," + lib])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60).stdout.split()
    assert out[0] == "ade0b876" and out[1] == "903df1a0" and int(out[2]) == 8
    bad = tmp_path / "seeded.cpp"
    bad.write_text('#include "cufhe_amd_legacy.hpp"\nint main() { cufhe::legacy::SetSeed(1234ull); }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(root, "include"), str(bad)],
                       capture_output=True, text=True)
    assert r.returncode != 0, "SetSeed(uint64_t) must not exist without CUFHE_AMD_INSECURE_TEST_KEYS"


def test_parameter_sets_match_the_oracle_builds():
    """cufhe_amd_ps_get_params for every compiled set == the numbers the oracle library of that set was built with."""
    lib = ctypes.CDLL(LIB)
    sys_path_hack = os.path.join(ol.ROOT)
    import sys
    sys.path.insert(0, sys_path_hack)
    from cufhe_amd._lib import PsParams
    lib.cufhe_amd_ps_get_params.argtypes = [ctypes.c_int, ctypes.POINTER(PsParams)]
    names = []
    for i in range(lib.cufhe_amd_ps_count()):
        p = PsParams()
        assert lib.cufhe_amd_ps_get_params(i, ctypes.byref(p)) == 0
        names.append(p.name.decode())
        _, want = ol.set_params(ol.load_set(p.name.decode()))
        assert (p.n, p.N, p.k, p.l, p.Bgbit, p.t, p.basebit) == tuple(want[k] for k in ("n", "N", "k", "l", "Bgbit", "t", "basebit"))
        assert p.bk_words == p.n * (p.k + 1) * p.l * (p.k + 1) * p.N
        assert p.ksk_words == p.k * p.N * p.t * (1 << (p.basebit - 1)) * (p.n + 1)
    assert names == list(ol.SETS)
    assert lib.cufhe_amd_ps_get_params(99, ctypes.byref(PsParams())) < 0


def test_shim_compiles_with_the_tfhepp_branch(tmp_path):
    """include/cufhe_amd.hpp with -DCUFHE_AMD_USE_TFHEPP -- the branch a cuFHE user compiles: parameter structs, TLWE / TRLWE / TRGSW
    and EvalKey from <params.hpp> / <cloudkey.hpp>, Initialize(const TFHEpp::EvalKey&) as src/cufhe_gates_gpu.cu:42-47 -- against the
    test-only stand-in headers of tests/cpp/tfhepp_stub (TFHEpp is an empty submodule in the reference tree; the stub pins nothing
    about it, it keeps the branch from rotting).  The reference's own test programs (tests/cpp/test_gate_api.cpp) compile and link
    in that configuration; tests/test_gpu_parity.py runs the binary on the GPU."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import cpp_build
    cpp_build.build_gate_api("test_gate_api_tfhepp", tfhepp=True, opt="-O1", extra=["-Wall"])
    # and a translation unit that uses nothing but the reference's own spellings
    src = tmp_path / "user.cpp"
    src.write_text('''#define CUFHE_AMD_USE_TFHEPP
#include "cufhe_amd.hpp"
void user(const TFHEpp::EvalKey& ek) {
    cufhe::SetGPUNum(1);
    cufhe::Initialize(ek);                                  // src/cufhe_gates_gpu.cu:42-47
    cufhe::Ctxt<TFHEpp::lvl1param> a, b, c;                 // test/test_gate_gpu.cc:36-91
    cufhe::Stream st; st.Create();
    cufhe::Nand(c, a, b, st);
    cufhe::Synchronize();
    cufhe::lvl2::Initialize(ek);
    st.Destroy(); cufhe::CleanUp();
}
''')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(root, "include"),
                           "-I" + os.path.join(root, "tests", "cpp", "tfhepp_stub"), str(src)])
    # the same user with the reference's small-modulus switch defined, as its CMake does (CMakeLists.txt:26-28): Initialize(ek) then
    # loads parameter set 3; the bootstrapping TRLWE-level calls stay, CMUXNTT / TRGSW2NTT are not declared (src/cufhe_gates_gpu.cu:68-86)
    small = tmp_path / "user_small.cpp"
    small.write_text(src.read_text() + '''
void user2(cufhe::Ctxt<TFHEpp::lvl0param>& in, cufhe::Ctxt<TFHEpp::lvl0param>& out, cufhe::Stream st) {
    cufhe::cuFHETRLWElvl1 t, r;
    cufhe::GateBootstrappingTLWE2TRLWElvl01NTT(t, in, st);
    cufhe::Refresh(r, t, st);
    cufhe::SampleExtractAndKeySwitch(out, r, st);
    static_assert(cufhe::kParamSetIndex == 3, "USE_SMALL_NTT_MODULUS selects the small-modulus set");
}
''')
    flags = ["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-DUSE_SMALL_NTT_MODULUS", "-I" + os.path.join(root, "include"),
             "-I" + os.path.join(root, "tests", "cpp", "tfhepp_stub")]
    subprocess.check_call(flags + [str(small)])
    cm = tmp_path / "user_cmux.cpp"
    cm.write_text(src.read_text() + "void user3() { cufhe::cuFHETRGSWNTTlvl1 cs; (void)cs; }\n")
    assert subprocess.run(flags + [str(cm)], capture_output=True).returncode != 0, "CMUXNTT types are declared in a small-modulus build"


def test_recorded_pmc_facts_belong_to_the_committed_device_code():
    """bench.py prints `roofline.traffic` / `valu` only while profiles/kernel_facts.json was recorded for the device code in the
    tree: a kernel edit without new rocprofv3 passes (tools/profile_round.sh + tools/kernel_facts.py) must not go unnoticed."""
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    ns = {"os": os, "hashlib": hashlib, "ROOT": root}
    exec(src[src.index("DEVICE_SOURCES = "):src.index("_FACTS = None")], ns)
    facts = json.load(open(os.path.join(root, "profiles", "kernel_facts.json")))
    assert facts["source_sha256"] == ns["source_hash"](), "profiles/kernel_facts.json is stale: re-run the PMC passes"
    for k in ("blind_rotate_kernel", "blind_rotate_lvl2_kernel", "blind_rotate_ll2_kernel", "keyswitch_kernel",
              "blind_rotate_ps_batch_kernel<default>", "blind_rotate_ps_batch_kernel<k2n512>", "blind_rotate_ps_batch_kernel<cggi16>"):
        assert facts["kernels"][k]["valu_insts_per_rotation"] > 0


def test_one_selector_for_the_parameter_set(tmp_path):
    """The reference has ONE selector: the numbers of TFHEpp's parameter structs (CMakeLists.txt:8-24, include/bootstrap_gpu.cuh:51-53).
    include/cufhe_amd.hpp derives the library's compiled set from those numbers at compile time: TFHEpp built for another shape needs
    nothing naming a set of this library, and numbers NO compiled set has -- Bgbit 7; (t, basebit) = (4, 3), whose key-switching key has
    the SIZE of (8, 2) -- do not compile, with a message that names the numbers.  At run time cufhe_amd_initialize_params repeats the
    match inside the library (tests/test_gpu_paramsets.py: refused before any device work)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = os.path.join(root, "tests", "cpp", "tfhepp_stub")
    src = tmp_path / "which.cpp"
    src.write_text('#define CUFHE_AMD_USE_TFHEPP\n#include "cufhe_amd.hpp"\n#ifndef WANT\n#define WANT 0\n#endif\n'
                   'static_assert(cufhe::kParamSetIndex == WANT, "wrong set");\n'
                   '#ifndef USE_SMALL_NTT_MODULUS\nvoid f() { cufhe::cuFHETRGSWNTTlvl1 cs; static_assert(sizeof(cs.trgswhost) == '
                   'sizeof(double) * cufhe::kKeyLimbs * (TFHEpp::lvl1param::k + 1) * (TFHEpp::lvl1param::k + 1) * TFHEpp::lvl1param::l * TFHEpp::lvl1param::n, "holder"); }\n#endif\n')
    base = ["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(root, "include"), "-I" + stub, str(src)]
    for flags, want in (([], 0), (["-DUSE_CONCRETE"], 1), (["-DUSE_80BIT_SECURITY"], 2), (["-DUSE_SMALL_NTT_MODULUS"], 3)):
        subprocess.check_call(base + flags + ["-DWANT=%d" % want])
    for flags in (["-DTFHEPP_STUB_BGBIT=7"], ["-DTFHEPP_STUB_T=4", "-DTFHEPP_STUB_BASEBIT=3"], ["-DUSE_80BIT_SECURITY", "-DUSE_SMALL_NTT_MODULUS"]):
        r = subprocess.run(base + flags, capture_output=True, text=True)
        assert r.returncode != 0 and "no parameter set compiled into libcufhe_amd.so" in r.stderr, (flags, r.stderr[-1500:])
    # level_of<P>() goes by TYPE: a third parameter struct is refused, whatever its n
    bad = tmp_path / "level.cpp"
    bad.write_text('#include "cufhe_amd.hpp"\nstruct other { using T = uint32_t; static constexpr uint32_t n = 630, k = 1; };\n'
                   'int f() { return cufhe::detail::level_of<other>(); }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(root, "include"), str(bad)], capture_output=True, text=True)
    assert r.returncode != 0 and "specialised for P = TFHEpp::lvl0param and TFHEpp::lvl1param" in r.stderr


def test_header_table_of_sets_matches_the_library():
    """detail::kCompiledSets of include/cufhe_amd.hpp (what the compile-time match runs on) == cufhe_amd_ps_get_params of the library,
    set by set; cufhe_amd_find_param_set finds each and refuses near misses with the numbers in the text."""
    from cufhe_amd import _lib
    from cufhe_amd._lib import PsParams
    lib = _lib.lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "cufhe_amd.hpp")).read()
    table = text[text.index("constexpr SetNumbers kCompiledSets[] = {"):]
    table = table[:table.index("};")]
    rows = [tuple(int(v.replace("kSmallNttP", str((625 << 20) + 1))) for v in re.findall(r"\{([^}]*)\}", line)[0].split(","))
            for line in table.splitlines()[1:] if "{" in line]
    assert len(rows) == lib.cufhe_amd_ps_count()
    for i, row in enumerate(rows):
        p = PsParams()
        assert lib.cufhe_amd_ps_get_params(i, ctypes.byref(p)) == 0
        assert row == (p.n, p.nbit, p.k, p.l, p.Bgbit, p.t, p.basebit, p.key_limbs, p.small_ntt_modulus), (i, row)
        q = _lib.ParamNumbers(p.n, p.nbit, p.k, p.l, p.Bgbit, p.t, p.basebit, p.small_ntt_modulus)
        assert lib.cufhe_amd_find_param_set(ctypes.byref(q)) == i
    for miss in (_lib.ParamNumbers(630, 10, 1, 3, 7, 8, 2, 0), _lib.ParamNumbers(630, 10, 1, 3, 6, 4, 3, 0)):
        assert lib.cufhe_amd_find_param_set(ctypes.byref(miss)) == -1
        assert b"no compiled parameter set has n=630" in lib.cufhe_amd_last_error()
    buf = (ctypes.c_uint32 * 4)()
    assert lib.cufhe_amd_initialize_params(ctypes.byref(miss), buf, 4, buf, 4) == -1      # refused before the keys are looked at
