// Halo-staged stride-1 3x3 implicit-GEMM convolution for gfx950 (round 6): the ping-pong kernel of gemm_pp.hip with the A operand staged ONCE per
// 32-channel chunk and re-used for the nine taps.
//
// Why.  gemm_pp_kernel<BM, conv> walks k = (32-channel chunk, tap) and DMAs, for every one of the nine taps, the BM x 32 block of the image shifted by
// that tap: 16 (BM = 256) / 8 (BM = 128) one-KB global_load_lds pieces of A + 20 of B per k-step and CU.  The main-loop ablation of round 5
// (profiles/r05_pingpong_conv_ablation.txt) shows everything-but-the-MFMAs (DMA issue, waits, fragment restart) longer than the MFMAs themselves, and the
// round-6 counter pass (profiles/r06_pmc_lds_mainloop.txt) shows the LDS array 18 % busy and the matrix pipe 51 %: the loop is bound by operand DELIVERY
// (pieces issued per MFMA), not by the LDS.  The nine taps of a chunk read the SAME pixels shifted by one row / one column, so here the tile's pixels plus a
// one-pixel halo travel to LDS once per chunk -- (R + 2) image rows of W + 2 pixels for a tile of R = BM / W rows -- and the taps are shifted fragment
// windows of that image: 27 pieces of A per NINE k-steps instead of 144 (BM = 256, W = 64), 207 instead of 324 in all (-36 %).
//
// LDS image of a chunk.  Pixel slot P = r * S + c, S = W + 8 slots per row (c = 0 is x = -1, c = W + 1 is x = W; the remaining six pad the row stride to a
// multiple of 8 so that every tap shift and every row crossing moves P by k * 8 + {0, 1, 2}), 64 bytes per slot = four 16-byte k-chunks; slot-chunk s of
// pixel P holds k-chunk s ^ 2 * ((P >> 2) & 1).  A 16-lane service group of a ds_read_b128 whose lanes read 16 CONSECUTIVE pixels at ANY offset then covers
// the 64 banks exactly once (the 16-row-group permutation G = {0, 3, 2, 1} of the other kernels is conflict-free only for windows aligned to four pixels;
// a tap shift of one pixel is not): lanes with equal P mod 4 come as (lg, P >> 2) = (a, u), (a ^ 1, u + 1), (a ^ 1, u + 2), (a, u + 3) and the four
// resulting slots a ^ h(u), a ^ 1 ^ h(u + 1), ... are distinct iff h alternates between {0, 1} and {2, 3} with period 2.  Halo pixels outside the image
// (and the pad slots) are DMA'd from the zero page; the source permutation is applied to the per-lane global address, the LDS side stays lane-linear.
//
// Schedule.  Ring of four B stages (20 pieces per k-step) exactly as in gemm_pp.hip, two A buffers; the A pieces of chunk c + 1 ride with the B loads issued
// during taps 0 .. NAP - 1 of chunk c (one per wave and tap; NAP = 4 / 3 / 2), so the counted vmcnt in front of barrier B_i allows
// NLB + [an A piece rode with the group issued one step earlier] loads in flight -- a compile-time pattern because the nine taps are unrolled.  The A
// buffer being refilled held chunk c - 1, whose last reader finished before B_9c (the same argument as the B ring's); its first reader starts after
// B_9c+9, five barriers behind the wait that retires the last A piece.
//
// Results are BIT-identical to gemm_pp_kernel<BM, conv>: same k order (chunk, tap), same MFMA sequence per accumulator, same epilogues.
#include "gemm_pp_device.h"

template <int BM, int W>
struct halo_geo {
    static constexpr int R = BM / W, S = W + 8;
    static constexpr int NPX = (R + 2) * S;
    static constexpr int NPIECE = NPX / 16;                              // one-KB pieces per chunk
    static constexpr int NAP = (NPIECE + PP_NW - 1) / PP_NW;             // per wave
    static constexpr int ABUF = NPIECE * PP_GROUP;                       // halfs per buffer
    static_assert(BM % W == 0 && (BM / 2) % W == 0 && NPX % 16 == 0 && W % 16 == 0, "tile = whole image rows, wave tiles too");
};
constexpr int HALO_BST = PP_NGB * PP_GROUP;                              // halfs per B stage (20 KB)
template <int BM, int W> constexpr size_t halo_lds() { return (size_t)(2 * halo_geo<BM, W>::ABUF + PP_NST * HALO_BST + PP_NW * PP_GROUP) * sizeof(f16); }

#ifndef HALO_A_AUX
#define HALO_A_AUX 0          // cache policy of the A pieces (measurement: -DHALO_A_AUX=2 = nt, streamed-once hint)
#endif
__device__ __forceinline__ int halo_swz(int P) { return ((P >> 2) & 1) << 1; }

// CV = 1 / 3: plain / with the GroupNorm-statistics epilogue (the numbering of gemm_pp_kernel's convolution variants)
template <int BM, int W, int CV, bool PRIO>
__global__ __launch_bounds__(512) void conv_halo_kernel(fd_gemm_desc p, int ntm, int ntn, int gn) {
    FD_WG_TRACE(6);
#if __HIP_DEVICE_COMPILE__     // the host pass does not know __amdgpu_buffer_rsrc_t and silently drops the kernel's host stub with it
    using G = halo_geo<BM, W>;
    constexpr bool WSTATS = CV >= 2;
    constexpr int WTM = BM / 2, WTN = 80, TM = WTM / 16, TN = 5;
    constexpr int NAP = G::NAP, S = G::S;
    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const bool lead = wave < 4;

    int tile = xcd_remap(blockIdx.x, gridDim.x);
    int mt, nt;
    if (gn >= ntn) {
        mt = tile / ntn;
        nt = tile - mt * ntn;
    } else {
        const int per = ntm * gn, nbands = (ntn + gn - 1) / gn;
        const int band = min(tile / per, nbands - 1);
        const int r = tile - band * per;
        const int w = band == nbands - 1 ? ntn - band * gn : gn;
        mt = r / w;
        nt = band * gn + (r - mt * w);
    }
    const int m0 = mt * BM, n0 = nt * PP_BN;
    const int img = m0 / (W * W), y0 = (m0 - img * W * W) / W;           // square maps, the tile lies inside one image (checked by the launcher)

    const int nch = p.Cin >> 5, nk = nch * 9;

    f16* const abuf = smem;
    f16* const ring = smem + 2 * G::ABUF;
    f16* const dump = ring + PP_NST * HALO_BST + wave * PP_GROUP;

    // Operand DMA goes through BUFFER descriptors (buffer_load_dwordx4 ... lds: 32-bit per-lane byte offset + scalar offset, range-checked by the
    // hardware): a lane whose halo pixel lies outside the image carries an out-of-range offset and the load returns zeros for it -- no per-lane pointer
    // select (which the compiler turns into divergent control flow inside the unrolled taps: +70 live registers and spills reloaded behind
    // s_waitcnt vmcnt(0), i.e. a drained DMA ring) and no 64-bit address arithmetic at all in the loop.
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(uint32_t)((int64_t)p.Bn * W * W * p.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(uint32_t)((int64_t)p.N * p.ldb * 2), 0x00020000);
    // A pieces of this wave: piece j = wave + 8 i covers pixel slots [16 j, 16 j + 16); this lane writes slot-chunk lane & 3 of pixel 16 j + (lane >> 2)
    uint32_t a_src[NAP];
#pragma unroll
    for (int i = 0; i < NAP; ++i) {
        const int j = wave + PP_NW * i;
        const int P = j * 16 + (lane >> 2);
        const int r = P / S, c = P - r * S;
        const int x = c - 1, y = y0 - 1 + r;
        const bool valid = j < G::NPIECE && x >= 0 && x < W && y >= 0 && y < W;
        a_src[i] = valid ? (uint32_t)((((int64_t)(img * W + y) * W + x) * p.lda + (((lane & 3) ^ halo_swz(P)) << 3)) * 2) : 0xFFFFFFF0u;     // bytes; < 2^32 (checked by fd_gemm)
    }
    // B groups: waves 0-3 stage groups w, w + 8, w + 16, waves 4-7 groups w, w + 8 (16 rows x 64 bytes each, gemm_glds_kernel's image); N is a multiple of
    // 320 here, so every row exists; group w + 8 i is 128 i rows further on (scalar offset)
    const uint32_t b_src = (uint32_t)((n0 + wave * 16 + (lane >> 2)) * (int)p.ldb + ((lane & 3) ^ swz_g(lane >> 4)) * 8) * 2;
    const uint32_t b_gstep = (uint32_t)(PP_NW * 16 * (int)p.ldb) * 2;

    // the B loads of k-step (cc, tap) into ring slot (9 cc + tap) & 3; past the last step: into the dump group (uniform load counts; whatever they read)
    auto issueB = [&](int cc, int tap, auto nl_c) {
        constexpr int NL = decltype(nl_c)::value;
        const bool live = cc < nch;
        f16* st = live ? ring + ((cc * 9 + tap) & 3) * HALO_BST + wave * PP_GROUP : dump;
        const int gstep = live ? PP_NW * PP_GROUP : 0;
        uint32_t kk = live ? (uint32_t)(tap * p.Cin + (cc << 5)) * 2 : 0u;
        asm volatile("" : "+s"(kk));         // opaque per step: nothing derived from it is hoisted out of the unrolled taps
#pragma unroll
        for (int i = 0; i < NL; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(st + i * gstep), 16, b_src, kk + i * b_gstep, 0, 0);
    };
    // A piece i of chunk cc into buffer cc & 1
    auto issueA = [&](int cc, auto i_c) {
        constexpr int i = decltype(i_c)::value;
        const int j = wave + PP_NW * i;
        const bool live = cc < nch && j < G::NPIECE;
        uint32_t c32 = live ? (uint32_t)(cc << 6) : 0u;
        asm volatile("" : "+s"(c32));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(live ? abuf + (cc & 1) * G::ABUF + j * PP_GROUP : dump), 16, a_src[i], c32, 0, HALO_A_AUX);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    // A fragments: lane (l15, lg) of fragment 0 reads pixel slot P0 + (tap shift) = wave row * S + l15 + ky * S + kx, k-chunk lg; the permutation depends on kx only
    const int P0 = (wm * WTM / W) * S + l15;
    uint32_t a_base[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) a_base[kx] = lds0 + (uint32_t)((P0 + kx) * 32 + ((lg ^ halo_swz(P0 + kx)) << 3)) * 2;
    const uint32_t bfrag = (uint32_t)(l15 * 32 + ((lg ^ swz_g(l15 >> 2)) * 8)) * 2;
    const uint32_t b_frag = lds0 + (uint32_t)(2 * G::ABUF + (wn * (WTN / 16)) * PP_GROUP) * 2 + bfrag;
    constexpr uint32_t ABUF_B = G::ABUF * 2, BST_B = HALO_BST * 2, GROUP_B = PP_GROUP * 2;

    auto run = [&](auto nlb_c, auto lead_c) {
        constexpr int NLB = decltype(nlb_c)::value;
        constexpr bool LEAD = decltype(lead_c)::value;
        std::integral_constant<int, NLB> nl;
        static_for<0, NAP>([&](auto ic) { issueA(0, ic); });
        issueB(0, 0, nl); issueB(0, 1, nl); issueB(0, 2, nl);
        wait_vm<2 * NLB>();                  // chunk 0 of A and L_0 landed
        raw_barrier();                       // B_-1
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const uint32_t ab = (uint32_t)(c & 1) * ABUF_B;
            static_for<0, 9>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int ky = t / 3, kx = t % 3;
                // loads allowed in flight at B_i: the group issued one step earlier = its B loads + the A piece that rode with them (taps 0 .. NAP - 1)
                constexpr int NW = NLB + ((((t + 8) % 9) < NAP) ? 1 : 0);
                constexpr int t3 = (t + 3) % 9;
                const int c3 = c + (t + 3) / 9;
                uint32_t so = (uint32_t)((c * 9 + t) & 3) * BST_B, ao = ab + (uint32_t)(ky * S * 64);
                asm volatile("" : "+s"(so), "+s"(ao));       // opaque per step (see issueB)
                const uint32_t aa = a_base[kx] + ao;
                auto issue = [&] {
                    issueB(c3, t3, nl);      // into the slot of step i - 1: every wave is past B_i, i.e. done with it
                    if constexpr (t < NAP) issueA(c + 1, tc);
#ifdef HALO_GN_PROBE
                    // MEASUREMENT ONLY (scratch/r06 probe; results are not meaningful): what would GroupNorm + SiLU applied to the staged A operand cost inside this
                    // loop?  Chunk c + 1 has landed behind B_9c+5; during taps 6 .. 8 every thread takes its share of the buffer's 16-byte items through
                    // LDS -> fp32 scale / shift (read from an LDS table) -> SiLU -> fp16 -> LDS, the arithmetic of gn_apply_kernel<silu>.
                    if constexpr (t >= 6) {
                        constexpr int NITEM = G::NPIECE * 64, PER = (NITEM + 511) / 512;
                        constexpr int J0 = t == 6 ? 0 : (t == 7 ? (PER + 2) / 3 : 2 * ((PER + 2) / 3)), J1 = t == 6 ? (PER + 2) / 3 : (t == 7 ? 2 * ((PER + 2) / 3) : PER);
                        f16* nb = abuf + ((c + 1) & 1) * G::ABUF;
                        const float* tab = (const float*)(ring + PP_NST * HALO_BST);        // the dump groups stand in for the 32-channel (a, b) table
#pragma unroll
                        for (int j = J0; j < J1; ++j) {
                            const int it = min(tid + 512 * j, NITEM - 1);      // branch-free (a divergent branch in the unrolled taps makes the allocator spill): the tail lanes redo the last item
                            {
                                f16x8 v = *(f16x8*)(nb + it * 8);
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const f32x4 sc = *(const f32x4*)(tab + (it & 3) * 16 + h * 4), sh = *(const f32x4*)(tab + (it & 3) * 16 + 8 + h * 4);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[h * 4 + e] = (f16)silu_f((float)v[h * 4 + e] * sc[e] + sh[e]);
                                    asm volatile("" : "+v"(v));
                                }
                                *(f16x8*)(nb + it * 8) = v;
                            }
                        }
                    }
#endif
                };
                if constexpr (LEAD) {
                    auto mid = [&] {
                        wait_vm<NW>();       // this wave's L_i+1 landed
                        raw_barrier();       // B_i, crossed in mid-step
                    };
                    mma_k32_mid<TM, TN, 2, GROUP_B, PRIO, decltype(mid)&, 0, W>(acc, aa, b_frag + so, mid);
                    issue();
                } else {
                    wait_vm<NW>();
                    raw_barrier();           // B_i, crossed at the step boundary
                    issue();
                    auto mid = [] {};
                    mma_k32_mid<TM, TN, 2, GROUP_B, PRIO, decltype(mid)&, 0, W>(acc, aa, b_frag + so, mid);
                }
            });
        }
    };
    if (lead) run(std::integral_constant<int, 3>{}, std::true_type{});
    else run(std::integral_constant<int, 2>{}, std::false_type{});

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // drain the pad loads before the operand area is reused by the epilogue
    __syncthreads();

    const bool lds_epi = p.out_dtype == FD_OUT_F16 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (!p.residual || (p.ldr & 7) == 0) &&
                         (!p.rowbias || (p.ld_rowbias & 3) == 0);
    constexpr int TMC = BM == 256 ? TM / 2 : TM;
    static_assert(PP_NW * TMC * 16 * (WTN + 4) <= 2 * G::ABUF + PP_NST * HALO_BST, "epilogue staging does not fit the operand area");
    if (lds_epi) {
        gemm_epilogue_lds<TM, TN, TMC, WSTATS>(p, acc, smem + wave * (TMC * 16) * (WTN + 4), m0 + wm * WTM, n0 + wn * WTN, lane, 0, 0);
    } else {
        gemm_epilogue<TM, TN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, l15, lg, 0, 0);
    }
#endif
}

template <int BM, int W, int CV, bool PRIO>
static void launch_halo(const fd_gemm_desc& d, hipStream_t s, int ntm, int ntn, int gn) {
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)conv_halo_kernel<BM, W, CV, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)halo_lds<BM, W>());
    });
    constexpr size_t lds = halo_lds<BM, W>();
    hipLaunchKernelGGL((conv_halo_kernel<BM, W, CV, PRIO>), dim3(ntm * ntn), dim3(512), lds, s, d, ntm, ntn, gn);
}

template <int BM, int W>
static void launch_halo_w(const fd_gemm_desc& d, hipStream_t s, bool prio) {
    const int ntm = d.M / BM, ntn = d.N / PP_BN;
    const long l2_budget = 3 * 1024 * 1024;
    long gnl = l2_budget / ((long)PP_BN * d.K * 2);
    const int gn = (int)(gnl < 1 ? 1 : (gnl > ntn ? ntn : gnl));
    (void)prio;      // the s_setprio form only (policy bit 4 of gemm.hip is on in every build since round 3)
    if (d.gn_stats) launch_halo<BM, W, 3, true>(d, s, ntm, ntn, gn);
    else launch_halo<BM, W, 1, true>(d, s, ntm, ntn, gn);
}

// stride-1 3x3 convolutions of square 16^2 / 32^2 / 64^2 maps whose tiles are whole image rows of one image, no split-K
bool fd_conv_halo_eligible(const fd_gemm_desc& d, int bm) {
    if (!d.conv || d.conv_mode != FD_CONV_NORMAL || d.batch > 1 || (d.N % 320) != 0 || (d.Cin & 31) != 0 || d.K2 != 0 || d.act == FD_ACT_GEGLU) return false;
    if (d.H != d.W || d.Ho != d.H || d.Wo != d.W || (d.W != 16 && d.W != 32 && d.W != 64)) return false;
    return (d.W * d.W) % bm == 0 && d.M % bm == 0 && (bm / 2) % d.W == 0 && (int64_t)d.N * d.ldb < (1LL << 31);
}

int fd_conv_halo_launch(const fd_gemm_desc& d, hipStream_t s, bool prio, int bm) {
#ifdef HALO_QUICK
    launch_halo<HALO_QUICK, 64, 1, true>(d, s, 1, 1, 1);
    return 0;
#else
    if (bm == 256) {
        if (d.W == 64) launch_halo_w<256, 64>(d, s, prio);
        else if (d.W == 32) launch_halo_w<256, 32>(d, s, prio);
        else launch_halo_w<256, 16>(d, s, prio);
    } else {
        if (d.W == 64) launch_halo_w<128, 64>(d, s, prio);
        else if (d.W == 32) launch_halo_w<128, 32>(d, s, prio);
        else launch_halo_w<128, 16>(d, s, prio);
    }
    return fd_check_launch("fd_gemm(conv halo)");
#endif
}

FD_WGT_SETTER(gemm_halo)
