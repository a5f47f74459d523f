// paramsets.inc.h -- host side of the parameter-set-generic gate path (kernels_ps.hip.h); included by
// capi.hip.  The reference chooses its parameter set when it is built (CMakeLists.txt:8-24); here every
// set of kernels_ps.hip.h is compiled in and chosen by index at run time ("param_set" puts the whole per-gate
// API on one, as the reference's build-time choice does).  Both gate orders of the reference: on lvl0 ciphertexts
// of the chosen set blind rotate -> sample extract -> key switch (__HomGate__ br -> iks, src/bootstrap_gpu.cu:402-421;
// Mux :515-588), on lvl1 ciphertexts (k N + 1 words) key switch of the linear combination -> blind rotate -> sample
// extract (__HomGate__ iks -> br, :383-400; Mux :706-780); Not / Copy :681-703.

namespace {

long g_ps_batch_threshold = -1;    // rotations per launch from which the wave-per-rotation kernel is used (-1: by cost)
// By cost (tools/ps_latency.py, tools/ps_times.py; blind rotation + key switch, MI355X): the workgroup-per-rotation kernel takes
// 4.3 / 3.4 / 3.7 ms per started round of one rotation per CU (default / k2n512 / cggi16; 4.2 / 3.0 / 3.5 for a few rotations), a round of the wave-per-rotation kernel
// (up to eight per CU) 21 / 18.5 / 26 ms: the second wins from the fifth / sixth / seventh started round on (the seventh of cggi16: a tie).
template <class PS> long ps_auto_batch(int cus) { return (PS::limbs > 1 ? 6L : PS::Nbit == 9 ? 5L : 4L) * std::max(1, cus) + 1; }

struct PsState {
    bool ready = false, lds_opt_in = false, ks_lds_opt_in = false;
    double* bk_ntt = nullptr;
    uint32_t* ksk = nullptr;
    uint32_t* ksk_padded = nullptr;   // sets whose key switch has the default shape: rows padded for keyswitch_kernel
};

// kN = 1024, n = 630, t = 8, basebit = 2: the hand-scheduled shared-table key switch of kernels.hip.h applies as it is
template <class PS>
constexpr bool ps_ks_is_default_shape = PS::k * PsDims<PS>::N == kN && PS::n == kLvl0N && PS::t == kKsT && PS::basebit == kKsBasebit;
static_assert(PsKs<PsDefault>::row_pad == kKsRowPad, "a set with the default key-switch shape shares keyswitch_kernel<KsShapeDefault>");

template <class PS>
int ps_launch_keyswitch(DeviceState& s, PsState& ps, hipStream_t st, const LinDesc* d, size_t count)
{
    if (count == 0) return 0;
    EventPair ev{};
    if (int rc = prof_begin(s, st, ev)) return rc;
    // keyswitch_kernel over the set's shape (j cut into runs that fill the CUs) at any count; the workgroup-per-ciphertext kernel
    // only by "ks_wg_threshold" (or when the padded table could not be built)
    if (ps.ksk_padded && !(g_ks_wg_threshold > 0 && (long)count <= g_ks_wg_threshold)) {
        if constexpr (ps_ks_is_default_shape<PS>) {
            if (int rc = launch_keyswitch_shared<KsShapeDefault>(s, st, d, count, ps.ksk_padded, &s.ks_lds_opt_in)) return rc;
        } else {
            if (int rc = launch_keyswitch_shared<KsShapePs<PS>>(s, st, d, count, ps.ksk_padded, &ps.ks_lds_opt_in)) return rc;
        }
        HIP_TRY(hipGetLastError());
        return prof_end(s, st, ev, count, true);
    }
    hipLaunchKernelGGL(keyswitch_ps_kernel<PS>, dim3((unsigned)count), dim3(kKsThreads), 0, st, d, (int)count, ps.ksk);
    HIP_TRY(hipGetLastError());
    return prof_end(s, st, ev, count, true);
}
PsState g_ps[kParamSets][kMaxLogicalDevices];

PsState& ps_state(int set, int device) { return g_ps[set][device]; }

template <class F>
int ps_dispatch(int set, F f)
{
    switch (set) {
        case 0: return f(PsDefault{});
        case 1: return f(PsK2N512{});
        case 2: return f(PsCggi16{});
        case 3: return f(PsSmallMod{});
    }
    return fail(-1, "unknown parameter set");
}

template <class PS>
const typename Poly<PS::Nbit>::Tables* ps_tables(DeviceState& s)
{
    if constexpr (PS::Nbit == 10) return s.tables;
    else return s.tables512 + 2;            // the stand-alone 512-point negacyclic transform
}

// the wave-per-rotation kernel of a 1024-point set runs the radix-4 transform: its tables
template <class PS>
const typename Poly<PS::Nbit>::Tables* ps_tables_batch(DeviceState& s)
{
    if constexpr (PS::Nbit == 10) return PsbPolyOf<PS>::kR4Tables ? s.tables_r4 : s.tables;
    else return s.tables512 + 2;
}

template <class PS>
int ps_launch_blind_rotate(DeviceState& s, PsState& ps, hipStream_t st, const LinDesc* d, size_t count, int steps, uint32_t* dump)
{
    if (count == 0) return 0;
    EventPair ev{};
    if (int rc = prof_begin(s, st, ev)) return rc;
    if (!ps.lds_opt_in) {
        HIP_TRY(hipFuncSetAttribute((const void*)blind_rotate_ps_kernel<PS>, hipFuncAttributeMaxDynamicSharedMemorySize, PsLds<PS>::bytes));
        HIP_TRY(hipFuncSetAttribute((const void*)blind_rotate_ps_batch_kernel<PS>, hipFuncAttributeMaxDynamicSharedMemorySize, PsbLds<PS>::bytes));
        ps.lds_opt_in = true;
    }
    if ((long)count >= (g_ps_batch_threshold < 0 ? ps_auto_batch<PS>(cus_of(s)) : g_ps_batch_threshold)) {
        // one wave per rotation, 8 rotations per workgroup share the key rows (throughput shape)
        const unsigned blocks = (unsigned)((count + PsbLds<PS>::waves - 1) / PsbLds<PS>::waves);
        hipLaunchKernelGGL(blind_rotate_ps_batch_kernel<PS>, dim3(blocks), dim3(PsbLds<PS>::threads), PsbLds<PS>::bytes, st, d, (int)count,
                           ps.bk_ntt, ps_tables_batch<PS>(s), steps, dump);
    } else {
        // one workgroup per rotation (latency shape)
        hipLaunchKernelGGL(blind_rotate_ps_kernel<PS>, dim3((unsigned)count), dim3(PsLds<PS>::threads), PsLds<PS>::bytes, st, d, (int)count,
                           ps.bk_ntt, kPsWgR4 ? ps_tables_batch<PS>(s) : ps_tables<PS>(s), steps, dump);
    }
    HIP_TRY(hipGetLastError());
    return prof_end(s, st, ev, count, false);
}

template <class PS, class GetGate>
int ps_run_gates(int set, int device, void* stream, int level, size_t count, GetGate get)
{
    using D = PsDims<PS>;
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    PsState& ps = ps_state(set, device);
    if (!ps.ready) return fail(-3, "cufhe_amd_ps_initialize has not been called for this parameter set and device");
    if (level != 0 && level != 1) return fail(-1, "level must be 0 or 1");
    if (count == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t negmu = 0u - kMu;
    size_t nrot = 0;
    for (size_t g = 0; g < count; g++) {
        const int op = get(g).op;
        if (op < 0 || op >= CUFHE_AMD_NUM_OPS) return fail(-1, "unknown gate op");
        if (op == CUFHE_AMD_MUX || op == CUFHE_AMD_NMUX) nrot += 2;
        else if (op < CUFHE_AMD_MUX) nrot += 1;
    }
    Scratch sc;
    if (int rc = open_scratch(s, st, nrot * (D::lvl1_words + D::lvl0_words) * sizeof(uint32_t) + (count * 5 + 8) * sizeof(LinDesc) + 8192, &sc)) return rc;
    uint32_t *tmp1 = nullptr, *tmp0 = nullptr;     // lvl1 / lvl0 temporaries, one per rotation
    if (nrot) {
        if (int rc = sc.alloc((void**)&tmp1, nrot * D::lvl1_words * sizeof(uint32_t))) return rc;
        if (level == 1)
            if (int rc = sc.alloc((void**)&tmp0, nrot * D::lvl0_words * sizeof(uint32_t))) return rc;
    }
    std::vector<LinDesc> rot, ks, lin;
    size_t ir = 0;
    for (size_t g = 0; g < count; g++) {
        const GateRef gr = get(g);
        if (!gr.out || !gr.in0) return fail(-1, "null ciphertext pointer");
        if (gr.op == CUFHE_AMD_NOT || gr.op == CUFHE_AMD_COPY) {
            lin.push_back({gr.in0, gr.in0, gr.out, gr.op == CUFHE_AMD_NOT ? -1 : 1, 0, 0u, 0u});
            continue;
        }
        if (!gr.in1) return fail(-1, "gate needs a second operand");
        if (gr.op == CUFHE_AMD_MUX || gr.op == CUFHE_AMD_NMUX) {
            if (!gr.in2) return fail(-1, "mux needs a third operand");
            uint32_t* ta = tmp1 + (ir + 0) * D::lvl1_words;
            uint32_t* tb = tmp1 + (ir + 1) * D::lvl1_words;
            const bool neg = gr.op == CUFHE_AMD_NMUX;
            if (level == 0) {   // src/bootstrap_gpu.cu:515-588
                rot.push_back({gr.in0, gr.in1, ta, 1, 1, negmu, 0u});
                rot.push_back({gr.in0, gr.in2, tb, -1, 1, negmu, 0u});
                ks.push_back({ta, tb, gr.out, neg ? -1 : 1, neg ? -1 : 1, neg ? negmu : kMu, 0u});
            } else {            // src/bootstrap_gpu.cu:706-780: two key switches, two rotations, the sum of the extracted ciphertexts
                uint32_t* t0a = tmp0 + (ir + 0) * D::lvl0_words;
                uint32_t* t0b = tmp0 + (ir + 1) * D::lvl0_words;
                ks.push_back({gr.in0, gr.in1, t0a, 1, 1, negmu, 0u});
                ks.push_back({gr.in0, gr.in2, t0b, -1, 1, negmu, 0u});
                rot.push_back({t0a, t0a, ta, 1, 0, 0u, 0u});
                rot.push_back({t0b, t0b, tb, 1, 0, 0u, 0u});
                lin.push_back({ta, tb, gr.out, neg ? -1 : 1, neg ? -1 : 1, neg ? negmu : kMu, 0u});
            }
            ir += 2;
            continue;
        }
        const int ca = kGateTab[gr.op][0], cb = kGateTab[gr.op][1];
        const uint32_t off = (uint32_t)kGateTab[gr.op][2] * kMu;
        if (level == 0) {       // __HomGate__ br -> iks, src/bootstrap_gpu.cu:402-421
            uint32_t* t1 = tmp1 + ir * D::lvl1_words;
            rot.push_back({gr.in0, gr.in1, t1, ca, cb, off, 0u});
            ks.push_back({t1, t1, gr.out, 1, 0, 0u, 0u});
        } else {                // __HomGate__ iks -> br, src/bootstrap_gpu.cu:383-400
            uint32_t* t0 = tmp0 + ir * D::lvl0_words;
            ks.push_back({gr.in0, gr.in1, t0, ca, cb, off, 0u});
            rot.push_back({t0, t0, gr.out, 1, 0, 0u, 0u});
        }
        ir += 1;
    }
    LinDesc *drot, *dks, *dlin;
    if (int rc = upload_descs(s, sc, rot, &drot)) return rc;
    if (int rc = upload_descs(s, sc, ks, &dks)) return rc;
    if (int rc = upload_descs(s, sc, lin, &dlin)) return rc;
    if (level == 0) {
        if (int rc = ps_launch_blind_rotate<PS>(s, ps, st, drot, rot.size(), PS::n, nullptr)) return rc;
        if (int rc = ps_launch_keyswitch<PS>(s, ps, st, dks, ks.size())) return rc;
    } else {
        if (int rc = ps_launch_keyswitch<PS>(s, ps, st, dks, ks.size())) return rc;
        if (int rc = ps_launch_blind_rotate<PS>(s, ps, st, drot, rot.size(), PS::n, nullptr)) return rc;
    }
    return launch_lincomb(st, dlin, lin.size(), level ? D::lvl1_words : D::lvl0_words);
}

template <class GetGate>
int run_gates_ps(int set, int device, void* stream, int level, size_t count, GetGate get)
{
    return ps_dispatch(set, [&](auto psx) { return ps_run_gates<decltype(psx)>(set, device, stream, level, count, get); });
}

// The TRLWE-level operations of the per-gate API on a set (run_trlwe_ops, capi.hip, over PS): lvl0 TLWE -> TRLWE
// (__BlindRotateGlobal__, src/bootstrap_gpu.cu:317-323), TRLWE -> TRLWE (__SEIandBootstrap2TRLWE__, :325-364), TRLWE -> lvl0 TLWE
// (__SEIandKS__, src/keyswitch_gpu.cu:26-40) as one launch sequence, and the CMUXNTT calls of the level (src/bootstrap_gpu.cu:197-285:
// in the reference the set chosen at build time serves them too; only its small-modulus build leaves them out, src/cufhe_gates_gpu.cu:68-86).
template <class PS>
int ps_launch_cmux(DeviceState& s, hipStream_t st, const CmuxDesc* d, size_t count)
{
    using PO = Poly<PS::Nbit>;
    if (count == 0) return 0;
    const unsigned blocks = (unsigned)((count + kNttWavesPerBlock - 1) / kNttWavesPerBlock);
    hipLaunchKernelGGL(cmux_desc_ps_kernel<PS>, dim3(blocks), dim3(kNttThreads), PO::table_bytes + kNttWavesPerBlock * PO::tile_bytes, st, d,
                       (int)count, ps_tables<PS>(s));
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class PS>
int ps_run_trlwe_ops(int set, int device, void* stream, const GateRef* g, size_t n)
{
    using D = PsDims<PS>;
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    PsState& ps = ps_state(set, device);
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    constexpr size_t trlwe_words = (size_t)D::K1 * D::N;
    size_t n_se = 0, n_rot = 0, n_t0 = 0, n_cmux = 0;
    for (size_t i = 0; i < n; i++) {
        if (!g[i].out || !g[i].in0) return fail(-1, "null operand");
        switch (g[i].op) {
            case CUFHE_AMD_TL_BOOTSTRAP: n_rot++; break;
            case CUFHE_AMD_TL_REFRESH: n_se++; n_rot++; n_t0++; break;
            case CUFHE_AMD_TL_SEIKS: n_se++; break;
            case CUFHE_AMD_TL_CMUX:
                if (PS::small_modulus) return fail(-1, "CMUXNTT: the small-modulus build of the reference has none (src/cufhe_gates_gpu.cu:68-86)");
                if (!g[i].in1 || !g[i].in2) return fail(-1, "CMUXNTT: null operand");
                n_cmux++;
                break;
            default: return fail(-1, "unknown TRLWE-level op");
        }
    }
    if (n_cmux < n && !ps.ready) return fail(-3, "cufhe_amd_ps_initialize has not been called for this parameter set and device");
    if (n_cmux && !s.ntt_ready) return fail(-3, "Initialize() has not been called for this device");
    Scratch sc;
    if (int rc = open_scratch(s, st, (n_se * D::lvl1_words + n_t0 * D::lvl0_words + n_rot * trlwe_words) * 4 + (3 * n + 8) * sizeof(LinDesc) +
                                         n_cmux * sizeof(CmuxDesc) + 16384, &sc))
        return rc;
    if (n_cmux) {      // the CMUXNTT calls of this level: independent of its other operations (the scheduler's contract); needs no key
        if constexpr (!PS::small_modulus) {
            std::vector<CmuxDesc> cm;
            cm.reserve(n_cmux);
            for (size_t i = 0; i < n; i++)
                if (g[i].op == CUFHE_AMD_TL_CMUX) cm.push_back({g[i].in0, g[i].in1, g[i].out, (const double*)g[i].in2});
            CmuxDesc* dcm;
            if (int rc = upload_descs(s, sc, cm, &dcm)) return rc;
            if (int rc = ps_launch_cmux<PS>(s, st, dcm, cm.size())) return rc;
        }
        if (n_cmux == n) return 0;
    }
    uint32_t *t1 = nullptr, *t0 = nullptr, *dump = nullptr;
    if (n_se) if (int rc = sc.alloc((void**)&t1, n_se * D::lvl1_words * 4)) return rc;
    if (n_t0) if (int rc = sc.alloc((void**)&t0, n_t0 * D::lvl0_words * 4)) return rc;
    if (n_rot) if (int rc = sc.alloc((void**)&dump, n_rot * trlwe_words * 4)) return rc;
    std::vector<LinDesc> se, ks, rot, scat;
    size_t i_se = 0, i_t0 = 0, i_rot = 0;
    for (size_t i = 0; i < n; i++) {
        if (g[i].op == CUFHE_AMD_TL_CMUX) continue;
        if (g[i].op == CUFHE_AMD_TL_BOOTSTRAP) {
            rot.push_back({g[i].in0, g[i].in0, nullptr, 1, 0, 0u, 0u});
        } else {
            uint32_t* a = t1 + i_se++ * D::lvl1_words;
            se.push_back({g[i].in0, g[i].in0, a, 1, 0, 0u, 0u});
            if (g[i].op == CUFHE_AMD_TL_SEIKS) {
                ks.push_back({a, a, g[i].out, 1, 0, 0u, 0u});
                continue;
            }
            uint32_t* b = t0 + i_t0++ * D::lvl0_words;
            ks.push_back({a, a, b, 1, 0, 0u, 0u});
            rot.push_back({b, b, nullptr, 1, 0, 0u, 0u});
        }
        uint32_t* d = dump + i_rot++ * trlwe_words;
        scat.push_back({d, d, g[i].out, 1, 0, 0u, 0u});
    }
    LinDesc *dse, *dks, *drot, *dscat;
    if (int rc = upload_descs(s, sc, se, &dse)) return rc;
    if (int rc = upload_descs(s, sc, ks, &dks)) return rc;
    if (int rc = upload_descs(s, sc, rot, &drot)) return rc;
    if (int rc = upload_descs(s, sc, scat, &dscat)) return rc;
    if (!se.empty()) {
        hipLaunchKernelGGL(sample_extract_ps_kernel<PS>, dim3((unsigned)(se.size() < 2048 ? se.size() : 2048)), dim3(256), 0, st, dse, (int)se.size());
        HIP_TRY(hipGetLastError());
    }
    if (int rc = ps_launch_keyswitch<PS>(s, ps, st, dks, ks.size())) return rc;
    if (int rc = ps_launch_blind_rotate<PS>(s, ps, st, drot, rot.size(), PS::n, dump)) return rc;
    return launch_lincomb(st, dscat, scat.size(), (int)trlwe_words);
}

int run_trlwe_ops_ps(int set, int device, void* stream, const GateRef* g, size_t n)
{
    return ps_dispatch(set, [&](auto psx) { return ps_run_trlwe_ops<decltype(psx)>(set, device, stream, g, n); });
}

// words of a level-0 / level-1 ciphertext, (level 2) a TRLWE or (level 3) a TRGSW in the NTT domain (uint32 words: two per double) of a set
int ps_ctxt_words(int set, int level)
{
    int w = 0;
    (void)ps_dispatch(set, [&](auto psx) {
        using D = PsDims<decltype(psx)>;
        w = level == 3 ? (int)(2 * D::bk_ntt_step_doubles) : level == 2 ? D::K1 * D::N : level ? D::lvl1_words : D::lvl0_words;
        return 0;
    });
    return w;
}

// TRGSW2NTT on host memory over a set (cufhe_amd_trgsw_to_ntt_host while "param_set" is active): the same staging as the BASELINE
// path, the set's sizes and key-conversion kernel (limbs included)
int ps_trgsw_to_ntt_host(int set, int device, void* stream, const uint32_t* trgsw_host, double* trgsw_ntt_host)
{
    return ps_dispatch(set, [&](auto psx) -> int {
        using PS = decltype(psx);
        using D = PsDims<PS>;
        using PO = Poly<PS::Nbit>;
        if (PS::small_modulus) return fail(-1, "TRGSW2NTT: the small-modulus build of the reference has none (src/bootstrap_gpu.cu:73-95)");
        DeviceState& s = g_dev[device];
        hipStream_t st = (hipStream_t)stream;
        constexpr size_t in_bytes = D::bk_step_polys * D::N * sizeof(uint32_t), out_bytes = D::bk_ntt_step_doubles * sizeof(double);
        Scratch sc;
        if (int rc = open_scratch(s, st, in_bytes + out_bytes + 4096, &sc)) return rc;
        uint32_t* d_in;
        double* d_out;
        if (int rc = sc.alloc((void**)&d_in, in_bytes)) return rc;
        if (int rc = sc.alloc((void**)&d_out, out_bytes)) return rc;
        PinnedBlock* blk = nullptr;
        if (int rc = acquire_staging(s, in_bytes + out_bytes, &blk)) return rc;
        StagingOwner owner{s, blk};
        staging_hold(s, blk, true);          // the host reads the result out of the block after the stream has finished with it
        memcpy(blk->host, trgsw_host, in_bytes);
        HIP_TRY(hipMemcpyAsync(d_in, blk->host, in_bytes, hipMemcpyHostToDevice, st));
        const size_t waves = D::bk_step_polys * PS::limbs;
        hipLaunchKernelGGL(bk_to_ntt_ps_kernel<PS>, dim3((unsigned)((waves + kNttWavesPerBlock - 1) / kNttWavesPerBlock)), dim3(kNttThreads),
                           PO::table_bytes + kNttWavesPerBlock * PO::tile_bytes, st, d_out, d_in, (size_t)D::bk_step_polys, ps_tables<PS>(s),
                           balanced(powmod_u64(D::N, fpf::P_U64 - 2)));
        HIP_TRY(hipGetLastError());
        char* pin_out = (char*)blk->host + in_bytes;
        HIP_TRY(hipMemcpyAsync(pin_out, d_out, out_bytes, hipMemcpyDeviceToHost, st));
        if (int rc = staging_done_after(s, blk, st)) return rc;
        owner.recorded();
        HIP_TRY(hipStreamSynchronize(st));
        memcpy(trgsw_ntt_host, pin_out, out_bytes);
        return device_fault(device);
    });
}

void ps_release(int device)
{
    for (int set = 0; set < kParamSets; set++) {
        PsState& ps = ps_state(set, device);
        if (!ps.ready) continue;
        (void)hipFree(ps.bk_ntt);
        (void)hipFree(ps.ksk);
        if (ps.ksk_padded) (void)hipFree(ps.ksk_padded);
        ps = PsState{};
    }
}

}  // namespace

extern "C" {

int cufhe_amd_ps_count(void) { return kParamSets; }

int cufhe_amd_ps_get_params(int set, cufhe_amd_ps_params* p)
{
    if (!p) return fail(-1, "null");
    return ps_dispatch(set, [&](auto ps) {
        using PS = decltype(ps);
        using D = PsDims<PS>;
        memset(p, 0, sizeof(*p));
        strncpy(p->name, PS::name, sizeof(p->name) - 1);
        p->n = PS::n; p->N = D::N; p->nbit = PS::Nbit; p->k = PS::k; p->l = PS::l; p->Bgbit = PS::Bgbit;
        p->t = PS::t; p->basebit = PS::basebit; p->key_limbs = PS::limbs; p->key_limb_bits = PS::limb_bits; p->mu = kMu;
        p->lvl0_words = D::lvl0_words; p->lvl1_words = D::lvl1_words;
        p->bk_words = D::bk_words; p->ksk_words = D::ksk_words;
        p->bk_ntt_bytes = (uint64_t)PS::n * D::bk_ntt_step_doubles * sizeof(double);
        p->small_ntt_modulus = PS::small_modulus ? smallmod::P : 0u;
        return 0;
    });
}

int cufhe_amd_ps_initialize(int set, const uint32_t* bk, size_t bk_words, const uint32_t* ksk, size_t ksk_words)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!bk || !ksk) return fail(-1, "null key pointer");
    return ps_dispatch(set, [&](auto psx) -> int {
        using PS = decltype(psx);
        using D = PsDims<PS>;
        using PO = Poly<PS::Nbit>;
        if (bk_words != D::bk_words) return fail(-1, "bootstrapping key has the wrong size for this parameter set");
        if (ksk_words != D::ksk_words) return fail(-1, "key-switching key has the wrong size for this parameter set");
        // build first, swap last (as cufhe_amd_initialize): a failure leaves every device with the keys of this set it had
        std::vector<PsState> built((size_t)g_gpu_num);
        std::vector<uint32_t*> d_bk((size_t)g_gpu_num, nullptr);
        struct Undo {
            std::vector<PsState>& b; std::vector<uint32_t*>& d; bool armed = true;
            ~Undo()
            {
                for (size_t i = 0; i < b.size(); i++) {
                    if (!d[i] && !b[i].bk_ntt && !b[i].ksk && !b[i].ksk_padded) continue;
                    (void)hipSetDevice(phys_device((int)i));
                    (void)hipFree(d[i]);
                    if (armed) { (void)hipFree(b[i].bk_ntt); (void)hipFree(b[i].ksk); (void)hipFree(b[i].ksk_padded); }
                }
            }
        } undo{built, d_bk};
        for (int i = 0; i < g_gpu_num; i++) {
            if (int rc = ensure_ntt(i)) return rc;
            DeviceState& s = g_dev[i];
            PsState& ps = built[(size_t)i];
            HIP_TRY(hipSetDevice(phys_device(i)));
            HIP_TRY(init_malloc((void**)&ps.bk_ntt, (size_t)PS::n * D::bk_ntt_step_doubles * sizeof(double)));
            HIP_TRY(init_malloc((void**)&ps.ksk, D::ksk_words * sizeof(uint32_t)));
            HIP_TRY(hipMemcpy(ps.ksk, ksk, D::ksk_words * sizeof(uint32_t), hipMemcpyHostToDevice));
            {      // the same table with rows padded to a multiple of 64 words, for the shared-table kernels
                constexpr size_t w0 = D::lvl0_words, pad = PsKs<PS>::row_pad;
                const size_t ksk_rows = D::ksk_words / w0;
                HIP_TRY(init_malloc((void**)&ps.ksk_padded, ksk_rows * pad * sizeof(uint32_t)));
                HIP_TRY(hipMemset(ps.ksk_padded, 0, ksk_rows * pad * sizeof(uint32_t)));
                HIP_TRY(hipMemcpy2D(ps.ksk_padded, pad * sizeof(uint32_t), ksk, w0 * sizeof(uint32_t), w0 * sizeof(uint32_t), ksk_rows,
                                    hipMemcpyHostToDevice));
            }
            HIP_TRY(init_malloc((void**)&d_bk[(size_t)i], D::bk_words * sizeof(uint32_t)));
            HIP_TRY(hipMemcpy(d_bk[(size_t)i], bk, D::bk_words * sizeof(uint32_t), hipMemcpyHostToDevice));
            const size_t polys = D::bk_words / D::N, waves = polys * PS::limbs;
            const unsigned blocks = (unsigned)((waves + kNttWavesPerBlock - 1) / kNttWavesPerBlock);
            hipLaunchKernelGGL(bk_to_ntt_ps_kernel<PS>, dim3(blocks), dim3(kNttThreads), PO::table_bytes + kNttWavesPerBlock * PO::tile_bytes, 0,
                               ps.bk_ntt, d_bk[(size_t)i], polys, ps_tables<PS>(s), balanced(powmod_u64(D::N, fpf::P_U64 - 2)));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());        // also: nothing on this device still reads the keys that are about to go
        }
        for (int i = 0; i < g_gpu_num; i++) {
            PsState& ps = ps_state(set, i);
            (void)hipSetDevice(phys_device(i));
            if (ps.ready) {
                (void)hipFree(ps.bk_ntt);
                (void)hipFree(ps.ksk);
                if (ps.ksk_padded) (void)hipFree(ps.ksk_padded);
            }
            ps = built[(size_t)i];
            ps.ready = true;
        }
        undo.armed = false;
        return 0;
    });
}

/* The reference has ONE selector for its parameters: TFHEpp's macro fixes the numbers and every kernel is a template over them
 * (CMakeLists.txt:8-24, include/bootstrap_gpu.cuh:51-53).  Here the caller hands over the numbers it was compiled with and the
 * library picks the compiled set that has exactly those -- key SIZES alone do not see Bgbit, and t * 2^(basebit-1) is 16 for both
 * (8, 2) and (4, 3). */
int cufhe_amd_find_param_set(const cufhe_amd_param_numbers* q)
{
    if (!q) return fail(-1, "null");
    for (int set = 0; set < kParamSets; set++) {
        cufhe_amd_ps_params p;
        if (cufhe_amd_ps_get_params(set, &p)) continue;
        if (p.n == q->n && p.nbit == q->nbit && p.k == q->k && p.l == q->l && p.Bgbit == q->Bgbit && p.t == q->t && p.basebit == q->basebit &&
            p.small_ntt_modulus == q->small_ntt_modulus)
            return set;
    }
    char buf[320];
    snprintf(buf, sizeof buf, "no compiled parameter set has n=%u nbit=%u k=%u l=%u Bgbit=%u t=%u basebit=%u small_ntt_modulus=%u "
             "(cufhe_amd_ps_get_params lists the compiled sets)", q->n, q->nbit, q->k, q->l, q->Bgbit, q->t, q->basebit, q->small_ntt_modulus);
    return fail(-1, buf);
}

int cufhe_amd_initialize_params(const cufhe_amd_param_numbers* numbers, const uint32_t* bk, size_t bk_words, const uint32_t* ksk, size_t ksk_words)
{
    const int set = cufhe_amd_find_param_set(numbers);
    if (set < 0) return set;
    if (set == 0) {        // the BASELINE numbers: the hand-scheduled kernels
        if (int rc = cufhe_amd_initialize(bk, bk_words, ksk, ksk_words)) return rc;
        return cufhe_amd_set_option("param_set", -1);
    }
    if (int rc = cufhe_amd_initialize_ntt()) return rc;
    if (int rc = cufhe_amd_ps_initialize(set, bk, bk_words, ksk, ksk_words)) return rc;
    return cufhe_amd_set_option("param_set", set);
}

int cufhe_amd_ps_gate_batch_level(int set, int device, void* stream, int level, size_t count, const int32_t* ops, int ops_stride,
                                  uint32_t* out, const uint32_t* in0, const uint32_t* in1, const uint32_t* in2, size_t stride_words)
{
    if (!ops) return fail(-1, "null ops");
    return ps_dispatch(set, [&](auto psx) {
        return ps_run_gates<decltype(psx)>(set, device, stream, level, count, [&](size_t g) {
            return GateRef{ops[g * (size_t)ops_stride], out + g * stride_words, in0 ? in0 + g * stride_words : nullptr,
                           in1 ? in1 + g * stride_words : nullptr, in2 ? in2 + g * stride_words : nullptr};
        });
    });
}

int cufhe_amd_ps_gate_batch(int set, int device, void* stream, size_t count, const int32_t* ops, int ops_stride,
                            uint32_t* out, const uint32_t* in0, const uint32_t* in1, const uint32_t* in2, size_t stride_words)
{
    return cufhe_amd_ps_gate_batch_level(set, device, stream, 0, count, ops, ops_stride, out, in0, in1, in2, stride_words);
}

int cufhe_amd_ps_blind_rotate_batch(int set, int device, void* stream, size_t count, const uint32_t* tlwe0, uint32_t* acc, int steps)
{
    if (int rc = use_device(device)) return rc;
    if (!tlwe0 || !acc) return fail(-1, "null pointer");
    return ps_dispatch(set, [&](auto psx) -> int {
        using PS = decltype(psx);
        using D = PsDims<PS>;
        DeviceState& s = g_dev[device];
        PsState& ps = ps_state(set, device);
        if (!ps.ready) return fail(-3, "cufhe_amd_ps_initialize has not been called for this parameter set and device");
        const int st_steps = (steps < 0 || steps > PS::n) ? PS::n : steps;
        hipStream_t st = (hipStream_t)stream;
        std::vector<LinDesc> rot(count);
        for (size_t g = 0; g < count; g++) rot[g] = {tlwe0 + g * D::lvl0_words, tlwe0 + g * D::lvl0_words, nullptr, 1, 0, 0u, 0u};
        Scratch sc;
        if (int rc = open_scratch(s, st, count * sizeof(LinDesc) + 4096, &sc)) return rc;
        LinDesc* d;
        if (int rc = upload_descs(s, sc, rot, &d)) return rc;
        return ps_launch_blind_rotate<PS>(s, ps, st, d, count, st_steps, acc);
    });
}

int cufhe_amd_ps_trlwe_op_batch(int set, int device, void* stream, int op, size_t count, uint32_t* out, const uint32_t* in)
{
    if (!out || !in) return fail(-1, "null pointer");
    if (op != CUFHE_AMD_TL_BOOTSTRAP && op != CUFHE_AMD_TL_REFRESH && op != CUFHE_AMD_TL_SEIKS) return fail(-1, "unknown TRLWE-level op");
    const size_t w0 = (size_t)ps_ctxt_words(set, 0), wt = (size_t)ps_ctxt_words(set, 2);
    if (!w0) return fail(-1, "unknown parameter set");
    const size_t win = op == CUFHE_AMD_TL_BOOTSTRAP ? w0 : wt, wout = op == CUFHE_AMD_TL_SEIKS ? w0 : wt;
    std::vector<GateRef> g(count);
    for (size_t i = 0; i < count; i++) g[i] = GateRef{op, out + i * wout, in + i * win, nullptr, nullptr};
    return run_trlwe_ops_ps(set, device, stream, g.data(), count);
}

/* TRGSW2NTT / CMUXNTT on a set, device-resident (src/bootstrap_gpu.cu:75-94,197-285 instantiated for the set the build selected):
 * trgsw[count][(k+1)l][k+1][N] torus words -> trgsw_ntt[count][limbs][(k+1)l][k+1][N] doubles; res = c0 + trgsw [x] (c1 - c0) on
 * TRLWEs [count][(k+1)N].  Needs Initialize() only. */
int cufhe_amd_ps_trgsw_to_ntt_batch(int set, int device, void* stream, size_t count, const uint32_t* trgsw, double* trgsw_ntt)
{
    if (int rc = use_device(device)) return rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = ensure_ntt(device)) return rc;
    }
    if (!trgsw || !trgsw_ntt) return fail(-1, "null pointer");
    if (count == 0) return 0;
    return ps_dispatch(set, [&](auto psx) -> int {
        using PS = decltype(psx);
        using D = PsDims<PS>;
        using PO = Poly<PS::Nbit>;
        if (PS::small_modulus) return fail(-1, "TRGSW2NTT: the small-modulus build of the reference has none (src/bootstrap_gpu.cu:73-95)");
        const size_t polys = count * D::bk_step_polys, waves = polys * PS::limbs;
        hipLaunchKernelGGL(bk_to_ntt_ps_kernel<PS>, dim3((unsigned)((waves + kNttWavesPerBlock - 1) / kNttWavesPerBlock)), dim3(kNttThreads),
                           PO::table_bytes + kNttWavesPerBlock * PO::tile_bytes, (hipStream_t)stream, trgsw_ntt, trgsw, polys,
                           ps_tables<PS>(g_dev[device]), balanced(powmod_u64(D::N, fpf::P_U64 - 2)));
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

int cufhe_amd_ps_cmux_batch(int set, int device, void* stream, size_t count, const double* trgsw_ntt, const uint32_t* c1,
                            const uint32_t* c0, uint32_t* res)
{
    if (int rc = use_device(device)) return rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = ensure_ntt(device)) return rc;
    }
    if (!trgsw_ntt || !c1 || !c0 || !res) return fail(-1, "null pointer");
    if (count == 0) return 0;
    return ps_dispatch(set, [&](auto psx) -> int {
        using PS = decltype(psx);
        using D = PsDims<PS>;
        if constexpr (PS::small_modulus) {
            return fail(-1, "CMUXNTT: the small-modulus build of the reference has none (src/cufhe_gates_gpu.cu:68-86)");
        } else {
            DeviceState& s = g_dev[device];
            hipStream_t st = (hipStream_t)stream;
            constexpr size_t tw = (size_t)D::K1 * D::N;
            std::vector<CmuxDesc> cm(count);
            for (size_t g = 0; g < count; g++) cm[g] = {c1 + g * tw, c0 + g * tw, res + g * tw, trgsw_ntt + g * D::bk_ntt_step_doubles};
            Scratch sc;
            if (int rc = open_scratch(s, st, count * sizeof(CmuxDesc) + 4096, &sc)) return rc;
            CmuxDesc* d;
            if (int rc = upload_descs(s, sc, cm, &d)) return rc;
            return ps_launch_cmux<PS>(s, st, d, count);
        }
    });
}

int cufhe_amd_ps_keyswitch_batch(int set, int device, void* stream, size_t count, const uint32_t* tlwe1, uint32_t* tlwe0)
{
    if (int rc = use_device(device)) return rc;
    if (!tlwe0 || !tlwe1) return fail(-1, "null pointer");
    return ps_dispatch(set, [&](auto psx) -> int {
        using PS = decltype(psx);
        using D = PsDims<PS>;
        DeviceState& s = g_dev[device];
        PsState& ps = ps_state(set, device);
        if (!ps.ready) return fail(-3, "cufhe_amd_ps_initialize has not been called for this parameter set and device");
        if (count == 0) return 0;
        hipStream_t st = (hipStream_t)stream;
        std::vector<LinDesc> ks(count);
        for (size_t g = 0; g < count; g++) ks[g] = {tlwe1 + g * D::lvl1_words, tlwe1 + g * D::lvl1_words, tlwe0 + g * D::lvl0_words, 1, 0, 0u, 0u};
        Scratch sc;
        if (int rc = open_scratch(s, st, count * sizeof(LinDesc) + 4096, &sc)) return rc;
        LinDesc* d;
        if (int rc = upload_descs(s, sc, ks, &d)) return rc;
        return ps_launch_keyswitch<PS>(s, ps, st, d, count);
    });
}

}  // extern "C"
