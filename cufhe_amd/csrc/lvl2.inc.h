// lvl2.inc.h -- host side of the N = 2048 / 64-bit-torus gate path (included by capi.hip).
// Gates take and return lvl0 ciphertexts; the bootstrap runs through the lvl2 ring:
// blind rotate lvl02 -> sample extract -> key switch lvl20, i.e. __HomGate__ (br -> iks,
// src/bootstrap_gpu.cu:402-421) and the Mux of :515-588 instantiated at brP = lvl02,
// iksP = lvl20.  Kernels: kernels_lvl2.hip.h.

namespace {

constexpr uint64_t kPsi4096 = 245080461804091ull;     // psi^2 = PSI_2048, psi^1024 = ROOT4

// Tables of the two half transforms: root_h[m + g] = root[2m + h m + g], root[i] = psi^bitrev11(i)
void build_tables_lvl2(NttTables (&t)[2])
{
    std::vector<double> R(k2N), Rinv(k2N);
    const uint64_t psi_inv = powmod_u64(kPsi4096, fpf::P_U64 - 2);
    for (uint32_t i = 0; i < (uint32_t)k2N; i++) {
        R[i] = balanced(powmod_u64(kPsi4096, bitrev(i, k2Nbit)));
        Rinv[i] = balanced(powmod_u64(psi_inv, bitrev(i, k2Nbit)));
    }
    for (int h = 0; h < 2; h++) {
        std::vector<double> fwd(kN, 0.0), inv(kN, 0.0);
        for (int m = 1; m < kN; m <<= 1)
            for (int g = 0; g < m; g++) {
                fwd[m + g] = R[2 * m + h * m + g];
                inv[m + g] = Rinv[2 * m + h * m + g];
            }
        fill_tables(t[h], fwd, inv);
    }
}

// Tables of the four quarter transforms (kernels_lvl2q.hip.h): root_q[m + g] = root[4m + q m + g]
void build_tables_lvl2q(Ntt512Tables (&t)[4])
{
    std::vector<double> R(k2N), Rinv(k2N);
    const uint64_t psi_inv = powmod_u64(kPsi4096, fpf::P_U64 - 2);
    for (uint32_t i = 0; i < (uint32_t)k2N; i++) {
        R[i] = balanced(powmod_u64(kPsi4096, bitrev(i, k2Nbit)));
        Rinv[i] = balanced(powmod_u64(psi_inv, bitrev(i, k2Nbit)));
    }
    auto top = [](int idx) { int m = 1; while (2 * m <= idx) m *= 2; return m; };
    for (int q = 0; q < 4; q++)
        fill_tables_512(t[q], [&](int idx) { const int m = top(idx); return R[4 * m + q * m + (idx - m)]; },
                        [&](int idx) { const int m = top(idx); return Rinv[4 * m + q * m + (idx - m)]; });
}

int ensure_tables_lvl2(int device)
{
    DeviceState& s = g_dev[device];
    if (s.tables2 && s.tables2q) return 0;
    HIP_TRY(hipSetDevice(phys_device(device)));
    if (!s.cus) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, phys_device(device)));
        s.cus = prop.multiProcessorCount;
    }
    if (!s.tables2) {
        static NttTables host[2];
        build_tables_lvl2(host);
        HIP_TRY(hipMalloc((void**)&s.tables2, sizeof(host)));
        HIP_TRY(hipMemcpy(s.tables2, host, sizeof(host), hipMemcpyHostToDevice));
    }
    if (!s.tables2q) {
        static Ntt512Tables hostq[4];
        build_tables_lvl2q(hostq);
        for (int q = 0; q < 4; q++)
            if (!fill_r4_products_512(hostq[q])) return fail(-2, "quarter tables: the stage-b twiddles of a block are not I apart (radix-4 form)");
        HIP_TRY(hipMalloc((void**)&s.tables2q, sizeof(hostq)));
        HIP_TRY(hipMemcpy(s.tables2q, hostq, sizeof(hostq), hipMemcpyHostToDevice));
    }
    return 0;
}

// The half-transform kernel's layout of the key (495 MB) is built on first use: the quarter-transform kernel serves every launch above
// one rotation per CU, so a process that only ever runs batches never pays for the second layout.  The torus-domain key is kept on the
// HOST for that (165 MB of ordinary memory, once per process), not on the devices.
std::vector<uint64_t> g_bk2_host;
std::mutex g_bk2_mu;
int ensure_bk2_half_layout(DeviceState& s)
{
    std::lock_guard<std::mutex> lk(g_bk2_mu);
    if (s.bk2_ntt) return 0;
    if (g_bk2_host.empty()) return fail(-3, "cufhe_amd_lvl2_initialize has not been called");
    const size_t want_bk = g_bk2_host.size();
    uint64_t* d_bk = nullptr;
    double* half = nullptr;
    struct Undo { uint64_t*& d; double*& h; bool armed = true; ~Undo() { (void)hipFree(d); if (armed) (void)hipFree(h); } } undo{d_bk, half};
    HIP_TRY(init_malloc((void**)&half, (size_t)kLvl0N * k2BkStepDoubles * sizeof(double)));
    HIP_TRY(init_malloc((void**)&d_bk, want_bk * sizeof(uint64_t)));
    HIP_TRY(hipMemcpy(d_bk, g_bk2_host.data(), want_bk * sizeof(uint64_t), hipMemcpyHostToDevice));
    const size_t polys = want_bk / k2N, waves = polys * k2Limbs;
    const unsigned blocks = (unsigned)((waves + kNttWavesPerBlock - 1) / kNttWavesPerBlock);
    hipLaunchKernelGGL(bk2_to_ntt_kernel, dim3(blocks), dim3(kNttThreads), kNttWavesPerBlock * kTileBytes, 0,
                       half, d_bk, polys, s.tables2, balanced(powmod_u64(k2N, fpf::P_U64 - 2)));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());        // complete before any stream's kernel reads it
    undo.armed = false;
    s.bk2_ntt = half;
    return 0;
}

void lvl2_release_host_key()
{
    std::lock_guard<std::mutex> lk(g_bk2_mu);
    std::vector<uint64_t>().swap(g_bk2_host);
}

int launch_blind_rotate_lvl2(DeviceState& s, hipStream_t st, const RotDesc2* d, size_t count, int steps, uint64_t* acc_dump)
{
    if (count == 0) return 0;
    EventPair ev{};
    if (s.profiling) {
        HIP_TRY(hipEventCreate(&ev.a));
        HIP_TRY(hipEventCreate(&ev.b));
        HIP_TRY(hipEventRecord(ev.a, st));
    }
    const bool quarters = g_lvl2_kernel < 0 ? (long)count > (cus_of(s) > 0 ? cus_of(s) : 256) : g_lvl2_kernel == 1;
    if (quarters) {
        // four quarter waves per rotation, two rotations per CU (kernels_lvl2q.hip.h)
        if (!s.br2q_lds_opt_in) {
            HIP_TRY(hipFuncSetAttribute((const void*)blind_rotate_lvl2q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kQLdsBytes));
            s.br2q_lds_opt_in = true;
        }
        hipLaunchKernelGGL(blind_rotate_lvl2q_kernel, dim3((unsigned)count), dim3(kQThreads), kQLdsBytes, st, d, (int)count,
                           s.bk2q_ntt, s.tables2q, steps, acc_dump);
    } else {
        if (int rc = ensure_bk2_half_layout(s)) return rc;
        if (!s.br2_lds_opt_in) {
            HIP_TRY(hipFuncSetAttribute((const void*)blind_rotate_lvl2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, k3LdsBytes));
            s.br2_lds_opt_in = true;
        }
        hipLaunchKernelGGL(blind_rotate_lvl2_kernel, dim3((unsigned)count), dim3(k2Threads), k3LdsBytes, st, d, (int)count,
                           s.bk2_ntt, s.tables2, steps, acc_dump);
    }
    HIP_TRY(hipGetLastError());
    if (s.profiling) {
        HIP_TRY(hipEventRecord(ev.b, st));
        ev.units = count;
        std::lock_guard<std::mutex> lk(s.staging_mu);
        s.br_events.push_back(ev);
    }
    return 0;
}

int launch_keyswitch_lvl2(DeviceState& s, hipStream_t st, const LinDesc64* d, size_t count)
{
    if (count == 0) return 0;
    EventPair ev{};
    if (s.profiling) {
        HIP_TRY(hipEventCreate(&ev.a));
        HIP_TRY(hipEventCreate(&ev.b));
        HIP_TRY(hipEventRecord(ev.a, st));
    }
    // keyswitch_kernel over the lvl20 shape (j cut into runs that fill the CUs) at any count; the workgroup-per-ciphertext kernel
    // (2.1 us per ciphertext) only by "ks_wg_threshold"
    if (g_ks_wg_threshold > 0 && (long)count <= g_ks_wg_threshold) {
        hipLaunchKernelGGL(keyswitch_lvl2_kernel, dim3((unsigned)count), dim3(kKsThreads), 0, st, d, (int)count, s.ksk2);
    } else {
        if (int rc = launch_keyswitch_shared<KsShapeLvl2>(s, st, d, count, s.ksk2, &s.ks2_lds_opt_in)) return rc;
    }
    HIP_TRY(hipGetLastError());
    if (s.profiling) {
        HIP_TRY(hipEventRecord(ev.b, st));
        ev.units = count;
        std::lock_guard<std::mutex> lk(s.staging_mu);
        s.ks_events.push_back(ev);
    }
    return 0;
}

template <class GetGate>
int run_gates_lvl2(int device, void* stream, size_t count, GetGate get)
{
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    if (!s.keys2_ready) return fail(-3, "cufhe_amd_lvl2_initialize has not been called for this device");
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
    {
        const size_t need = nrot * k2Words * sizeof(uint64_t) + nrot * sizeof(RotDesc2) +
                            count * (sizeof(LinDesc64) + sizeof(LinDesc)) + 8192;
        if (int rc = open_scratch(s, st, need, &sc)) return rc;
    }
    uint64_t* tmp2 = nullptr;                 // one lvl2 TLWE per rotation
    if (nrot)
        if (int rc = sc.alloc((void**)&tmp2, nrot * k2Words * sizeof(uint64_t))) return rc;
    std::vector<RotDesc2> rot;
    std::vector<LinDesc64> ks;
    std::vector<LinDesc> lin;
    rot.reserve(nrot); ks.reserve(count); lin.reserve(count);
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
            uint64_t* ta = tmp2 + (ir + 0) * k2Words;
            uint64_t* tb = tmp2 + (ir + 1) * k2Words;
            const bool neg = gr.op == CUFHE_AMD_NMUX;
            rot.push_back({gr.in0, gr.in1, ta, 1, 1, negmu, 0u});
            rot.push_back({gr.in0, gr.in2, tb, -1, 1, negmu, 0u});
            ks.push_back({ta, tb, gr.out, neg ? -1 : 1, neg ? -1 : 1, neg ? 0ull - k2Mu : k2Mu});
            ir += 2;
            continue;
        }
        uint64_t* t = tmp2 + ir * k2Words;
        rot.push_back({gr.in0, gr.in1, t, kGateTab[gr.op][0], kGateTab[gr.op][1], (uint32_t)kGateTab[gr.op][2] * kMu, 0u});
        ks.push_back({t, t, gr.out, 1, 0, 0ull});
        ir += 1;
    }
    RotDesc2* drot;
    LinDesc64* dks;
    LinDesc* dlin;
    if (int rc = upload_descs(s, sc, rot, &drot)) return rc;
    if (int rc = upload_descs(s, sc, ks, &dks)) return rc;
    if (int rc = upload_descs(s, sc, lin, &dlin)) return rc;
    if (int rc = launch_blind_rotate_lvl2(s, st, drot, rot.size(), kLvl0N, nullptr)) return rc;
    if (int rc = launch_keyswitch_lvl2(s, st, dks, ks.size())) return rc;
    if (int rc = launch_lincomb(st, dlin, lin.size(), kLvl0Words)) return rc;
    return 0;
}

}  // namespace

extern "C" {

int cufhe_amd_lvl2_get_params(cufhe_amd_lvl2_params* p)
{
    if (!p) return fail(-1, "null");
    p->n = kLvl0N; p->N = k2N; p->nbit = k2Nbit; p->k = 1; p->l = k2L; p->Bgbit = k2Bgbit;
    p->t = k2KsT; p->basebit = k2KsBasebit;
    p->lvl0_words = kLvl0Words; p->lvl2_words = k2Words;
    p->mu = k2Mu;
    p->bk_words = (uint64_t)kLvl0N * k2BkRows * 2 * k2N;
    p->ksk_words = (uint64_t)k2N * k2KsT * k2KsNumBase * kKsRowWords;
    p->bk_ntt_bytes = (uint64_t)kLvl0N * k2BkStepDoubles * sizeof(double);
    return 0;
}

int cufhe_amd_lvl2_initialize(const uint64_t* bk, size_t bk_words, const uint32_t* ksk, size_t ksk_words)
{
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t want_bk = (size_t)kLvl0N * k2BkRows * 2 * k2N;
    const size_t want_ksk = (size_t)k2N * k2KsT * k2KsNumBase * kKsRowWords;
    if (!bk || !ksk) return fail(-1, "null key pointer");
    if (bk_words != want_bk) return fail(-1, "lvl02 bootstrapping key has the wrong size for this parameter set");
    if (ksk_words != want_ksk) return fail(-1, "lvl20 key-switching key has the wrong size for this parameter set");
    // build first, swap last (as cufhe_amd_initialize): the quarter-transform layout and the key-switching key of every device beside
    // what is loaded; the half-transform layout follows on first use (ensure_bk2_half_layout)
    struct Built { double* bk2q = nullptr; uint32_t* ksk2 = nullptr; uint64_t* d_bk = nullptr; };
    std::vector<Built> built((size_t)g_gpu_num);
    struct Undo {
        std::vector<Built>& b; bool armed = true;
        ~Undo()
        {
            for (size_t i = 0; i < b.size(); i++) {
                if (!b[i].bk2q && !b[i].ksk2 && !b[i].d_bk) continue;
                (void)hipSetDevice(phys_device((int)i));
                (void)hipFree(b[i].d_bk);
                if (armed) { (void)hipFree(b[i].bk2q); (void)hipFree(b[i].ksk2); }
            }
        }
    } undo{built};
    for (int i = 0; i < g_gpu_num; i++) {
        if (int rc = ensure_tables_lvl2(i)) return rc;
        DeviceState& s = g_dev[i];
        Built& b = built[(size_t)i];
        HIP_TRY(hipSetDevice(phys_device(i)));
        HIP_TRY(init_malloc((void**)&b.bk2q, (size_t)kLvl0N * k2BkStepDoubles * sizeof(double)));
        const size_t ksk_rows = want_ksk / kKsRowWords;
        HIP_TRY(init_malloc((void**)&b.ksk2, ksk_rows * kKsRowPad * sizeof(uint32_t)));
        HIP_TRY(hipMemset(b.ksk2, 0, ksk_rows * kKsRowPad * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy2D(b.ksk2, kKsRowPad * sizeof(uint32_t), ksk, kKsRowWords * sizeof(uint32_t),
                            kKsRowWords * sizeof(uint32_t), ksk_rows, hipMemcpyHostToDevice));
        HIP_TRY(init_malloc((void**)&b.d_bk, want_bk * sizeof(uint64_t)));
        HIP_TRY(hipMemcpy(b.d_bk, bk, want_bk * sizeof(uint64_t), hipMemcpyHostToDevice));
        const size_t polys = want_bk / k2N;
        const size_t waves = polys * k2Limbs;
        const unsigned blocks = (unsigned)((waves + kNttWavesPerBlock - 1) / kNttWavesPerBlock);
        hipLaunchKernelGGL(bk2q_to_ntt_kernel, dim3(blocks), dim3(kNttThreads), kNttWavesPerBlock * kTile512Bytes, 0,
                           b.bk2q, b.d_bk, polys, s.tables2q, balanced(powmod_u64(k2N, fpf::P_U64 - 2)));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());        // also: nothing on this device still reads the keys that are about to go
    }
    {
        std::lock_guard<std::mutex> lk2(g_bk2_mu);
        g_bk2_host.assign(bk, bk + want_bk);
        for (int i = 0; i < g_gpu_num; i++) {
            DeviceState& s = g_dev[i];
            (void)hipSetDevice(phys_device(i));
            if (s.keys2_ready) { (void)hipFree(s.bk2q_ntt); (void)hipFree(s.ksk2); }
            (void)hipFree(s.bk2_ntt);          // the half layout of the OLD key, if it was ever built
            s.bk2_ntt = nullptr;
            s.bk2q_ntt = built[(size_t)i].bk2q;
            s.ksk2 = built[(size_t)i].ksk2;
            s.keys2_ready = true;
        }
    }
    undo.armed = false;        // the guard still frees the torus-domain staging copies
    return 0;
}

int cufhe_amd_lvl2_gate_batch(int device, void* stream, size_t count, const int32_t* ops, int ops_stride,
                              uint32_t* out, const uint32_t* in0, const uint32_t* in1, const uint32_t* in2,
                              size_t stride_words)
{
    if (!ops) return fail(-1, "null ops");
    return run_gates_lvl2(device, stream, count, [&](size_t g) {
        return GateRef{ops[g * (size_t)ops_stride], out + g * stride_words, in0 ? in0 + g * stride_words : nullptr,
                       in1 ? in1 + g * stride_words : nullptr, in2 ? in2 + g * stride_words : nullptr};
    });
}

int cufhe_amd_lvl2_blind_rotate_batch(int device, void* stream, size_t count, const uint32_t* tlwe0, uint64_t* acc, int steps)
{
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    if (!s.keys2_ready) return fail(-3, "cufhe_amd_lvl2_initialize has not been called for this device");
    if (!tlwe0 || !acc) return fail(-1, "null pointer");
    if (steps < 0 || steps > kLvl0N) steps = kLvl0N;
    hipStream_t st = (hipStream_t)stream;
    std::vector<RotDesc2> rot(count);
    for (size_t g = 0; g < count; g++) rot[g] = {tlwe0 + g * kLvl0Words, tlwe0 + g * kLvl0Words, nullptr, 1, 0, 0u, 0u};
    Scratch sc;
    if (int rc = open_scratch(s, st, count * sizeof(RotDesc2) + 4096, &sc)) return rc;
    RotDesc2* d;
    if (int rc = upload_descs(s, sc, rot, &d)) return rc;
    return launch_blind_rotate_lvl2(s, st, d, count, steps, acc);
}

int cufhe_amd_lvl2_keyswitch_batch(int device, void* stream, size_t count, const uint64_t* tlwe2, uint32_t* tlwe0)
{
    if (int rc = use_device(device)) return rc;
    DeviceState& s = g_dev[device];
    if (!s.keys2_ready) return fail(-3, "cufhe_amd_lvl2_initialize has not been called for this device");
    if (!tlwe0 || !tlwe2) return fail(-1, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    std::vector<LinDesc64> ks(count);
    for (size_t g = 0; g < count; g++) ks[g] = {tlwe2 + g * k2Words, tlwe2 + g * k2Words, tlwe0 + g * kLvl0Words, 1, 0, 0ull};
    Scratch sc;
    if (int rc = open_scratch(s, st, count * sizeof(LinDesc64) + 4096, &sc)) return rc;
    LinDesc64* d;
    if (int rc = upload_descs(s, sc, ks, &d)) return rc;
    return launch_keyswitch_lvl2(s, st, d, count);
}

}  // extern "C"
