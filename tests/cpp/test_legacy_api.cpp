// test_legacy_api.cpp -- the reference's user-manual program (README.md:46-82,
// test/test_api_gpu.cu:74-215) against include/cufhe_amd_legacy.hpp: KeyGen, Encrypt/Decrypt,
// Initialize(pub_key), every gate in place on many streams, decrypt == plain function.
//   test_legacy_api --cpu-only   key generation, encryption round trip and key files only
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CUFHE_AMD_INSECURE_TEST_KEYS      // reproducible keys: this is a test
#include "../../include/cufhe_amd_legacy.hpp"

using namespace cufhe::legacy;

// plain functions of test/test_api_gpu.cu:25-72
static void NandCheck(Ptxt& out, const Ptxt& a, const Ptxt& b) { out.message_ = 1 - a.message_ * b.message_; }
static void OrCheck(Ptxt& out, const Ptxt& a, const Ptxt& b) { out.message_ = (a.message_ + b.message_) > 0; }
static void OrYNCheck(Ptxt& out, const Ptxt& a, const Ptxt& b) { out.message_ = (a.message_ + (1 - b.message_)) > 0; }
static void OrNYCheck(Ptxt& out, const Ptxt& a, const Ptxt& b) { out.message_ = ((1 - a.message_) + b.message_) > 0; }
static void AndCheck(Ptxt& out, const Ptxt& a, const Ptxt& b) { out.message_ = a.message_ * b.message_; }
static void AndYNCheck(Ptxt& out, const Ptxt& a, const Ptxt& b) { out.message_ = a.message_ * (1 - b.message_); }
static void AndNYCheck(Ptxt& out, const Ptxt& a, const Ptxt& b) { out.message_ = (1 - a.message_) * b.message_; }
static void XorCheck(Ptxt& out, const Ptxt& a, const Ptxt& b) { out.message_ = (a.message_ + b.message_) & 1; }
static void MuxCheck(Ptxt& out, const Ptxt& c, const Ptxt& a, const Ptxt& b) { out.message_ = c.message_ ? a.message_ : b.message_; }

int main(int argc, char** argv)
{
    const bool cpu_only = argc > 1 && !strcmp(argv[1], "--cpu-only");
    const int kNumSMs = 32, kNumTests = kNumSMs * 8;
    SetSeed(20261003);
    srand(7);
    PriKey pri_key;
    PubKey pub_key;
    KeyGen(pub_key, pri_key);

    Ptxt* pt = new Ptxt[3 * kNumTests];
    bool correct = true;
    {
        TFHEpp::TLWE<TFHEpp::lvl0param> ct;       // a Ctxt needs a device; a bare TLWE does not
        for (int i = 0; i < kNumTests; i++) {
            pt[i].message_ = rand() % Ptxt::kPtxtSpace;
            Encrypt(ct, pt[i], pri_key);
            Decrypt(pt[kNumTests + i], ct, pri_key);
            if (pt[kNumTests + i].message_ != pt[i].message_) correct = false;
        }
    }
    std::printf("encrypt/decrypt: %s\n", correct ? "PASS" : "FAIL");

    // key files
    {
        const char* f1 = "/tmp/cufhe_amd_pri.key";
        const char* f2 = "/tmp/cufhe_amd_pub.key";
        WritePriKeyToFile(pri_key, f1);
        WritePubKeyToFile(pub_key, f2);
        PriKey p2;
        PubKey q2;
        ReadPriKeyFromFile(p2, f1);
        ReadPubKeyFromFile(q2, f2);
        const bool same = p2.lvl0_key == pri_key.lvl0_key && p2.lvl1_key == pri_key.lvl1_key && q2.bk == pub_key.bk && q2.ksk == pub_key.ksk;
        std::printf("key files: %s\n", same ? "PASS" : "FAIL");
        correct = correct && same;
        std::remove(f1);
        std::remove(f2);
    }
    if (cpu_only) {
        std::printf("%s\n", correct ? "ALL PASS" : "FAILED");
        return correct ? 0 : 1;
    }

    Ctxt* ct = new Ctxt[3 * kNumTests];
    Initialize(pub_key);
    Stream* st = new Stream[kNumSMs];
    for (int i = 0; i < kNumSMs; i++) st[i].Create();
    for (int i = 0; i < 3 * kNumTests; i++) {
        pt[i] = rand() % Ptxt::kPtxtSpace;
        Encrypt(ct[i], pt[i], pri_key);
    }
    Synchronize();
    for (int i = 0; i < kNumTests; i++) Nand(ct[i], ct[i], ct[i + kNumTests], st[i % kNumSMs]);
    for (int i = 0; i < kNumTests; i++) Or(ct[i], ct[i], ct[i + kNumTests], st[i % kNumSMs]);
    for (int i = 0; i < kNumTests; i++) OrYN(ct[i], ct[i], ct[i + kNumTests], st[i % kNumSMs]);
    for (int i = 0; i < kNumTests; i++) OrNY(ct[i], ct[i], ct[i + kNumTests], st[i % kNumSMs]);
    for (int i = 0; i < kNumTests; i++) And(ct[i], ct[i], ct[i + kNumTests], st[i % kNumSMs]);
    for (int i = 0; i < kNumTests; i++) AndYN(ct[i], ct[i], ct[i + kNumTests], st[i % kNumSMs]);
    for (int i = 0; i < kNumTests; i++) AndNY(ct[i], ct[i], ct[i + kNumTests], st[i % kNumSMs]);
    for (int i = 0; i < kNumTests; i++) Xor(ct[i], ct[i], ct[i + kNumTests], st[i % kNumSMs]);
    for (int i = 0; i < kNumTests; i++) Mux(ct[i], ct[i], ct[i + kNumTests], ct[i + 2 * kNumTests], st[i % kNumSMs]);
    Synchronize();

    int cnt_failures = 0;
    for (int i = 0; i < kNumTests; i++) {
        NandCheck(pt[i], pt[i], pt[i + kNumTests]);
        OrCheck(pt[i], pt[i], pt[i + kNumTests]);
        OrYNCheck(pt[i], pt[i], pt[i + kNumTests]);
        OrNYCheck(pt[i], pt[i], pt[i + kNumTests]);
        AndCheck(pt[i], pt[i], pt[i + kNumTests]);
        AndYNCheck(pt[i], pt[i], pt[i + kNumTests]);
        AndNYCheck(pt[i], pt[i], pt[i + kNumTests]);
        XorCheck(pt[i], pt[i], pt[i + kNumTests]);
        MuxCheck(pt[i], pt[i], pt[i + kNumTests], pt[i + 2 * kNumTests]);
        Ptxt got;
        Decrypt(got, ct[i], pri_key);
        if (got.message_ != pt[i].message_) cnt_failures++;
    }
    std::printf("chained gates: %s (%d/%d failures)\n", cnt_failures ? "FAIL" : "PASS", cnt_failures, kNumTests);
    for (int i = 0; i < kNumSMs; i++) st[i].Destroy();
    delete[] st;
    delete[] ct;
    delete[] pt;
    CleanUp();
    const bool ok = correct && cnt_failures == 0;
    std::printf("%s\n", ok ? "ALL PASS" : "FAILED");
    return ok ? 0 : 1;
}
