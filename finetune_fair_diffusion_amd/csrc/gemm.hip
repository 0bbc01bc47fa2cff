// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950.
//   C[M,N] = act(alpha*(A.B^T + A2.B2^T) + bias + rowbias) + residual       (fp16 in, fp32 acc)
// Two kernel families, both staging operands global -> LDS with global_load_lds_dwordx4 (no VGPR round trip) into
// unpadded rows whose 16-byte slots are XOR-permuted so the ds_read_b128 fragment reads are bank-conflict free, and
// both issuing v_mfma_f32_16x16x32_f16 with the operands swapped (D^T = B.A^T) so a lane ends up holding 4
// consecutive N of one M row (8-byte epilogue stores along the contiguous dimension):
//   gemm_glds_kernel  4 waves, BK = 32, tiles 128x128 / 128x64 / 64x64  (short K, small problems, strided batch)
//   gemm_big_kernel   8 or 16 waves, BK = 64, tiles 256x320 / 128x320 / 128x160 / 256x128, optional split-K
#include "gemm_device.h"

template <int BM, int BN, bool CONV>
__global__ __launch_bounds__(256) void gemm_glds_kernel(fd_gemm_desc p, int ntm, int ntn) {
    FD_WG_TRACE(1);
    constexpr int TM = BM / 32, TN = BN / 32;
    constexpr int AI = BM / 64, BI = BN / 64;     // 16-row groups per wave per k-tile
    constexpr int EPI_HALFS = 4 * (BM / 2) * (BN / 2 + 4);           // 4 waves x (wave tile + row pad)
    constexpr int OPS_HALFS = 2 * (BM + BN) * 32;
    __shared__ __attribute__((aligned(16))) f16 smem[OPS_HALFS > EPI_HALFS ? OPS_HALFS : EPI_HALFS];
    f16* As = smem;
    f16* Bs = smem + 2 * BM * 32;
    // the zero page's address comes through the GOT: pinned in SGPRs once -- left to the compiler it is re-loaded (s_load + lgkmcnt(0)) in every
    // k-tile, in front of the next tile's global_load_lds issue
    const f16* zp = fd_zero_page;
    asm volatile("" : "+s"(zp));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;

    const int tile = xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int z = blockIdx.y;

    const f16* A = (const f16*)p.A + (int64_t)z * p.sA;
    const f16* B = (const f16*)p.B + (int64_t)z * p.sB;
    const f16* A2 = (const f16*)p.A2;
    const f16* B2 = (const f16*)p.B2;
    const int nk1 = (p.K + 31) >> 5, nk2 = (p.K2 + 31) >> 5, nk = nk1 + nk2;

    // this lane's slot in a 16-row group: row lane>>2, 16-byte slot lane&3 holding k-chunk (slot ^ G[row>>2])
    const int lrow = lane >> 2;
    const int kchunk = ((lane & 3) ^ swz_g(lane >> 4)) * 8;
    int arow[AI], brow[BI];
    ConvRow crow[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        arow[i] = (wave * AI + i) * 16 + lrow;
        if (CONV) {
            const int m = m0 + arow[i];
            const int hw = p.Ho * p.Wo;
            crow[i].valid = m < p.M;
            const int mm = crow[i].valid ? m : 0;
            crow[i].b = mm / hw;
            const int r = mm - crow[i].b * hw;
            crow[i].oy = r / p.Wo;
            crow[i].ox = r - crow[i].oy * p.Wo;
        }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) brow[i] = (wave * BI + i) * 16 + lrow;
    const int cpt = CONV ? (p.Cin >> 5) : 1;

    auto issue = [&](int kt, int buf) {
        if (CONV) {
            const int tap = kt / cpt;
            const int c0 = (kt - tap * cpt) << 5;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                int iy = crow[i].oy, ix = crow[i].ox;
                bool ok = crow[i].valid;
                if (p.conv_mode == FD_CONV_NORMAL) {
                    iy += ky - 1; ix += kx - 1;
                    ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                } else if (p.conv_mode == FD_CONV_STRIDE2) {
                    iy = 2 * iy + ky - 1; ix = 2 * ix + kx - 1;
                    ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                } else if (p.conv_mode == FD_CONV_UP2) {
                    iy += ky - 1; ix += kx - 1;
                    ok = ok && iy >= 0 && iy < 2 * p.H && ix >= 0 && ix < 2 * p.W;
                    iy >>= 1; ix >>= 1;
                } else {
                    iy += ky - 1; ix += kx - 1;
                    ok = ok && iy >= 0 && ix >= 0 && !(iy & 1) && !(ix & 1);
                    iy >>= 1; ix >>= 1;
                    ok = ok && iy < p.H && ix < p.W;
                }
                const f16* src = ok ? A + (((int64_t)crow[i].b * p.H + iy) * p.W + ix) * p.lda + c0 + kchunk : zp;
                glds16(src, As + (buf * BM + (wave * AI + i) * 16) * 32);
            }
            const int kk = kt * 32 + kchunk;
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int n = n0 + brow[i];
                const f16* src = (n < p.N) ? B + (int64_t)n * p.ldb + kk : zp;
                glds16(src, Bs + (buf * BN + (wave * BI + i) * 16) * 32);
            }
        } else {
            const bool seg2 = kt >= nk1;
            const f16* Ap = seg2 ? A2 : A;
            const f16* Bp = seg2 ? B2 : B;
            const int64_t la = seg2 ? p.lda2 : p.lda, lb = seg2 ? p.ldb2 : p.ldb;
            const int Kseg = seg2 ? p.K2 : p.K;
            const int kk = (seg2 ? kt - nk1 : kt) * 32 + kchunk;
            const bool kok = kk < Kseg;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int m = m0 + arow[i];
                const f16* src = (kok && m < p.M) ? Ap + (int64_t)m * la + kk : zp;
                glds16(src, As + (buf * BM + (wave * AI + i) * 16) * 32);
            }
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int n = n0 + brow[i];
                const f16* src = (kok && n < p.N) ? Bp + (int64_t)n * lb + kk : zp;
                glds16(src, Bs + (buf * BN + (wave * BI + i) * 16) * 32);
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offset inside a 16-row group: row l15, slot (lg ^ G[l15>>2])
    const int frag_off = l15 * 32 + ((lg ^ swz_g(l15 >> 2)) * 8);

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
        f16x8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const f16x8*)(As + (buf * BM + wm * (BM / 2) + i * 16) * 32 + frag_off);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *(const f16x8*)(Bs + (buf * BN + wn * (BN / 2) + j * 16) * 32 + frag_off);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = FD_MFMA_16x16x32(bf[j], af[i], acc[i][j]);
    }

    const bool lds_epi = p.out_dtype == FD_OUT_F16 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (!p.residual || (p.ldr & 7) == 0) &&
                         (!p.rowbias || (p.ld_rowbias & 3) == 0);
    if (p.act == FD_ACT_GEGLU) {      // fd_gemm checked the preconditions (fp16 out, aligned, unbatched)
        __syncthreads();
        gemm_epilogue_geglu_lds<TM, TN, TM>(p, acc, smem + wave * (BM / 2) * (BN / 2 + 4), m0 + wm * (BM / 2), n0 + wn * (BN / 2), lane);
    } else if (lds_epi) {
        __syncthreads();   // every wave is done reading the operand stages before they are reused as epilogue staging
        gemm_epilogue_lds<TM, TN>(p, acc, smem + wave * (BM / 2) * (BN / 2 + 4), m0 + wm * (BM / 2), n0 + wn * (BN / 2), lane,
                                  (int64_t)z * p.sC, (int64_t)z * p.sR);
    } else {
        gemm_epilogue<TM, TN>(p, acc, m0 + wm * (BM / 2), n0 + wn * (BN / 2), l15, lg, (int64_t)z * p.sC, (int64_t)z * p.sR);
    }
}


// ---- pinned fragment schedule for the 8/16-wave main loop.
// Written as plain loads + MFMAs, the compiler (at the register cap of these kernels) sinks every streamed-fragment ds_read to
// directly in front of its consumers and waits lgkmcnt(0): read -> full LDS latency -> TN MFMAs, with only the second wave of the
// SIMD to cover it (measured: MFMA loop alone at ~55 % of the 1.6 PFLOP/s the same loop reaches with the reads ahead of use,
// scratch/mb_mfma_peak.hip).  Here the ds_read_b128 are inline asm with hand-counted s_waitcnt lgkmcnt(n): the resident operand's
// fragments are read once per k-step, the streamed operand runs PD fragments ahead through a ring of PD+1 registers, and a
// sched_barrier after each fragment's MFMAs keeps that order.  LDS reads return in order, so "n = reads issued after the one
// needed" is exact (and only ever conservative if the compiler adds LGKM operations of its own).
// ======================================================================================= 8/16-wave, BK = 64
// Staged bytes per FLOP bound the main loop (the L2 -> LDS path itself sustains 50-60 B/clk/CU, scratch/mb_l2_lds.hip; what is
// shared is the LDS array between the direct-to-LDS writes and the fragment reads), so arithmetic intensity -- tile size -- is the
// lever.  This kernel uses 8 or 16 waves (512 threads), a BM x BN x 64 tile (N tiles of 320/160 match the U-Net's channel counts, which are
// all multiples of 320), full 128-byte rows per operand row (one L2 line per row per k-tile), direct-to-LDS
// loads, the conflict-free XOR slot permutation chunk = slot ^ (row & 7), and two LDS stages.
// CONV: 0 = dense operands, 1 = the 3x3 gathers (stride 1 / stride 2 / nearest-up2 / transposed stride 2), 2 = the Upsample2D phase pair
// (FD_CONV_UP2P, FD_CONV_UP2P_BWD) -- its own instantiation: compiled into variant 1 the extra gather arithmetic cost the 8-wave 256x320
// and the 512x128 gathers 34 and 52 spilled registers
// CV = 3 / 4 / 6: variants 0 / 1 / 2 with the GroupNorm-statistics epilogue (fd_gemm_desc.gn_stats)
template <int BM, int BN, int WGM, int WGN, int CV>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_big_kernel(fd_gemm_desc p, int ntm, int ntn, int gn) {
    FD_WG_TRACE(2);
    static_assert(CV != 5, "CV = 5 was the LayerNorm second output (scratch/gemm_ln_epilogue_experiment.h)");
    constexpr int CONV = CV == 6 ? 2 : CV >= 3 ? CV - 3 : CV;      // CV = 6: the phase pair (variant 2) with the statistics epilogue (FD_CONV_UP2PI)
    constexpr bool WSTATS = CV == 3 || CV == 4 || CV == 6;
    constexpr int NW = WGM * WGN;                   // 8 or 16 waves
    static_assert(NW == 8 || NW == 16, "8 or 16 waves");
    constexpr int WTM = BM / WGM, WTN = BN / WGN;   // wave tile
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int NA = BM / 8, NB = BN / 8;         // 8-row groups (one glds instruction each) per k-tile
    constexpr int AI = (NA + NW - 1) / NW, BI = (NB + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    f16* As = smem;                      // [2][BM][64]
    f16* Bs = smem + 2 * BM * 64;        // [2][BN][64]
    const f16* zp = fd_zero_page;        // GOT load hoisted out of the main loop (see gemm_glds_kernel)
    asm volatile("" : "+s"(zp));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = wave / WGN, wn = wave % WGN;

    // tile order: an XCD walks a contiguous range of tile ids.  Row-major ids (n fastest) re-stream the whole B operand for every
    // row of tiles once B outgrows the 4 MB L2 (measured on 4096x10240x1280: 431 MB fetched for 121 MB of operands); banding the
    // n-tiles in groups of ``gn`` whose B slab fits the L2 (m fastest inside a band) reads B about once and A once per band.
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    // FD_CONV_UP2P: the four output phases (py, px) of conv3x3(nearest-up2(x)) are four 2x2-tap problems over the low-res input with
    // their own pre-summed weights; they share one launch, phase-major in the tile index, weights and output
    int phase = 0, up_phase = -1;      // up_phase >= 0: FD_CONV_UP2PI, the epilogue maps low-res rows of this phase to rows of the [Bn, 2H, 2W, N] result
    if (CONV == 2 && (p.conv_mode == FD_CONV_UP2P || p.conv_mode == FD_CONV_UP2PI)) {
        phase = tile / (ntm * ntn);
        tile -= phase * (ntm * ntn);
        p.B = (const f16*)p.B + (int64_t)phase * p.N * p.ldb;
        if (p.conv_mode == FD_CONV_UP2PI) {
            up_phase = phase;
            p.conv_mode = FD_CONV_UP2P;            // same gather from here on
        } else p.C = (f16*)p.C + (int64_t)phase * p.M * p.ldc;
    }
    const int ph_y = phase >> 1, ph_x = phase & 1;
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
    const int m0 = mt * BM, n0 = nt * BN;

    const f16* A = (const f16*)p.A;
    const f16* B = (const f16*)p.B;
    const f16* A2 = (const f16*)p.A2;
    const f16* B2 = (const f16*)p.B2;
    const int nk1 = (p.K + 63) >> 6, nk2 = (p.K2 + 63) >> 6, nk = nk1 + nk2;

    const int lrow = lane >> 3;                              // row inside the 8-row group
    const int kchunk = ((lane & 7) ^ lrow) * 8;              // k-chunk this lane's slot holds
    // per staged row: image index and packed (oy, ox); rows beyond M get coordinates that fail every bounds check
    int crow_b[AI], crow_yx[AI];
    if (CONV) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int m = m0 + (wave + i * NW) * 8 + lrow;
            const int hw = p.Ho * p.Wo;
            const bool valid = m < p.M && (wave + i * NW) < NA;
            const int mm = valid ? m : 0;
            const int b = mm / hw;
            const int r = mm - b * hw;
            const int oy = r / p.Wo;
            const int ox = r - oy * p.Wo;
            crow_b[i] = b;
            crow_yx[i] = valid ? ((oy << 16) | ox) : (int)0x80008000u;   // (-32768, -32768)
            if (CONV == 1 && p.conv_mode == FD_CONV_NORMAL) {
                // stride-1 convs (all ResBlock convs): per row the element offset of the centre pixel and a 9-bit tap-validity mask,
                // so that a k-step costs one add + one select per row instead of the coordinate arithmetic below
                int mask = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
                    if (valid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1 << t;
                }
                crow_b[i] = (int)(((int64_t)(b * p.H + oy) * p.W + ox) * p.lda) + kchunk;   // < 2^31 elements (checked by the launcher)
                crow_yx[i] = mask;
            }
        }
    }
    auto issue = [&](int kt, int buf, int part) {
        if (CONV) {
            // k order = (64-channel chunk, tap): the 9 taps of one chunk re-read the same 128-byte lines shifted by a
            // pixel, so the tile's working set per chunk (~48 KB) stays in L1/L2 instead of cycling all Cin channels
            const int ntap = CONV == 2 ? (p.conv_mode == FD_CONV_UP2P ? 4 : 16) : 9;
            const int cc = kt / ntap;
            const int tap = kt - cc * ntap;
            const int c0 = cc << 6;
            const int ky = tap / 3, kx = tap - ky * 3;
            if (CONV == 1 && p.conv_mode == FD_CONV_NORMAL) {
                const int toff = ((ky - 1) * p.W + (kx - 1)) * (int)p.lda + c0;
#pragma unroll
                for (int i = 0; i < AI; ++i) {
                    const int g = wave + i * NW;
                    if (g < NA && (part & 1)) {
                        const bool ok = (crow_yx[i] >> tap) & 1;
                        const f16* src = ok ? A + (int64_t)(crow_b[i] + toff) : zp;
                        glds16(src, As + (buf * BM + g * 8) * 64);
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int g = wave + i * NW;
                if (g < NA && (part & 1)) {
                    int iy = crow_yx[i] >> 16, ix = (int)(short)(crow_yx[i] & 0xffff);
                    bool ok = true;
                    if (CONV == 2) {
                        if (p.conv_mode == FD_CONV_UP2P) {                 // tap = dy*2+dx over the low-res input, shifted by the phase
                            iy += (tap >> 1) + ph_y - 1; ix += (tap & 1) + ph_x - 1;
                            ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                        } else {                                           // tap = ((py*2+px)*2+dy)*2+dx; source = high-res gradient [Bn,H,W], H = 2*Ho
                            const int qy = (tap >> 3) & 1, qx = (tap >> 2) & 1;
                            const int u = iy - ((tap >> 1) & 1) - qy + 1, v = ix - (tap & 1) - qx + 1;
                            ok = ok && u >= 0 && u < p.Ho && v >= 0 && v < p.Wo;
                            iy = 2 * u + qy; ix = 2 * v + qx;
                        }
                    } else if (p.conv_mode == FD_CONV_NORMAL) {
                        iy += ky - 1; ix += kx - 1;
                        ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    } else if (p.conv_mode == FD_CONV_STRIDE2) {
                        iy = 2 * iy + ky - 1; ix = 2 * ix + kx - 1;
                        ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    } else if (p.conv_mode == FD_CONV_UP2) {
                        iy += ky - 1; ix += kx - 1;
                        ok = ok && iy >= 0 && iy < 2 * p.H && ix >= 0 && ix < 2 * p.W;
                        iy >>= 1; ix >>= 1;
                    } else {
                        iy += ky - 1; ix += kx - 1;
                        ok = ok && iy >= 0 && ix >= 0 && !(iy & 1) && !(ix & 1);
                        iy >>= 1; ix >>= 1;
                        ok = ok && iy < p.H && ix < p.W;
                    }
                    const f16* src = ok ? A + (((int64_t)crow_b[i] * p.H + iy) * p.W + ix) * p.lda + c0 + kchunk : zp;
                    glds16(src, As + (buf * BM + g * 8) * 64);
                }
            }
            const int kk = tap * p.Cin + c0 + kchunk;
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int g = wave + i * NW;
                if (g < NB && (part & 2)) {
                    const int n = n0 + g * 8 + lrow;
                    const f16* src = (n < p.N) ? B + (int64_t)n * p.ldb + kk : zp;
                    glds16(src, Bs + (buf * BN + g * 8) * 64);
                }
            }
        } else {
            const bool seg2 = kt >= nk1;
            const f16* Ap = seg2 ? A2 : A;
            const f16* Bp = seg2 ? B2 : B;
            const int64_t la = seg2 ? p.lda2 : p.lda, lb = seg2 ? p.ldb2 : p.ldb;
            const int Kseg = seg2 ? p.K2 : p.K;
            const int kk = (seg2 ? kt - nk1 : kt) * 64 + kchunk;
            const bool kok = kk < Kseg;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int g = wave + i * NW;
                if (g < NA && (part & 1)) {
                    const int m = m0 + g * 8 + lrow;
                    const f16* src = (kok && m < p.M) ? Ap + (int64_t)m * la + kk : zp;
                    glds16(src, As + (buf * BM + g * 8) * 64);
                }
            }
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int g = wave + i * NW;
                if (g < NB && (part & 2)) {
                    const int n = n0 + g * 8 + lrow;
                    const f16* src = (kok && n < p.N) ? Bp + (int64_t)n * lb + kk : zp;
                    glds16(src, Bs + (buf * BN + g * 8) * 64);
                }
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read: row l15 of a 16-row tile, k-chunk (ks*4 + lg) lives in slot chunk ^ (row & 7)
    const int frow = l15 * 64;
    const int fsw = l15 & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    // same-box A/B of the whole training step: the 8-wave variants gain 5-7 % in situ (256x320 and 256x256 gathers); the 16-wave ones
    // (4 waves per SIMD already cover the LDS latency) do not move, and 512x128 spills 60 registers under the pinned order: they
    // keep the compiler-scheduled fragment loop
    constexpr bool PINNED = NW == 8;
#ifndef FD_PD
#define FD_PD 2        // depth of the streamed operand's register ring in the pinned loop (3 and 4 measured: see profiles/r02_gemm_ring_depth_ab.txt)
#endif

    // split-K: blockIdx.y owns the k-tiles [kbeg, kend) and writes raw fp32 partials to the workspace
    const int nsplit = gridDim.y;
    const int kbeg = (int)((int64_t)nk * blockIdx.y / nsplit), kend = (int)((int64_t)nk * (blockIdx.y + 1) / nsplit);
    issue(kbeg, 0, 3);
    for (int kt = kbeg; kt < (FD_DBG_GE(p, 4) ? kbeg : kend); ++kt) {   // FD_GEMM_DBG=4/5: epilogue only
        const int buf = (kt - kbeg) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // (issuing the B half between the two k-steps was measured: it pushes the 16-wave variant into scratch, 8x slower)
        if (kt + 1 < kend && !FD_DBG_IS(p, 1)) issue(kt + 1, buf ^ 1, 3);
        if (FD_DBG_IS(p, 2)) continue;
        if constexpr (PINNED) {
            const uint32_t a_base = lds0 + (uint32_t)((buf * BM + wm * WTM) * 64 + frow) * 2;
            const uint32_t b_base = lds0 + (uint32_t)((2 * BM + buf * BN + wn * WTN) * 64 + frow) * 2;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint32_t slot = (uint32_t)(((ks * 4 + lg) ^ fsw) * 16);
                mma_k32<TM, TN, FD_PD>(acc, a_base + slot, b_base + slot);
            }
        } else {
            const f16* Ab = As + (buf * BM + wm * WTM) * 64 + frow;
            const f16* Bb = Bs + (buf * BN + wn * WTN) * 64 + frow;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int slot = ((ks * 4 + lg) ^ fsw) * 8;
                // keep the smaller operand set resident and stream the other one with a 2-deep register prefetch, the
                // ds_read of fragment i+2 pinned in front of the MFMAs of fragment i (LDS latency hidden under 2x TN MFMAs)
                if (TN <= TM) {
                    f16x8 bf[TN];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[j] = *(const f16x8*)(Bb + j * 16 * 64 + slot);
                    f16x8 a0 = *(const f16x8*)(Ab + slot);
                    f16x8 a1 = *(const f16x8*)(Ab + (TM > 1 ? 1 : 0) * 16 * 64 + slot);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        f16x8 a2 = a1;
                        if (i + 2 < TM) a2 = *(const f16x8*)(Ab + (i + 2) * 16 * 64 + slot);
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = FD_MFMA_16x16x32(bf[j], a0, acc[i][j]);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);
                        a0 = a1;
                        a1 = a2;
                    }
                } else {
                    f16x8 af[TM];
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[i] = *(const f16x8*)(Ab + i * 16 * 64 + slot);
                    f16x8 b0 = *(const f16x8*)(Bb + slot);
                    f16x8 b1 = *(const f16x8*)(Bb + (TN > 1 ? 1 : 0) * 16 * 64 + slot);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f16x8 b2 = b1;
                        if (j + 2 < TN) b2 = *(const f16x8*)(Bb + (j + 2) * 16 * 64 + slot);
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[i][j] = FD_MFMA_16x16x32(b0, af[i], acc[i][j]);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
                        b0 = b1;
                        b1 = b2;
                    }
                }
            }
        }
    }

    if (nsplit > 1) {
        float* ws = (float*)p.workspace + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + lg * 4;
                if (n < p.N) *(f32x4*)(ws + (int64_t)m * p.N + n) = acc[i][j];   // split-K is only chosen when N % 4 == 0
            }
        }
        return;
    }
    const bool lds_epi = p.out_dtype == FD_OUT_F16 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (!p.residual || (p.ldr & 7) == 0) &&
                         (!p.rowbias || (p.ld_rowbias & 3) == 0);
    if (FD_DBG_IS(p, 3)) {   // measurement only (FD_GEMM_DBG=3): no epilogue at all
        if (acc[0][0][0] == 12345.678f) ((f16*)p.C)[0] = (f16)1.f;
    } else if (p.act == FD_ACT_GEGLU) {
        constexpr int LDS_HALFS = 2 * (BM + BN) * 64;
        constexpr int TMC = (NW * WTM * (WTN + 4) <= LDS_HALFS) ? TM : TM / 2;
        __syncthreads();
        gemm_epilogue_geglu_lds<TM, TN, TMC>(p, acc, smem + wave * (TMC * 16) * (WTN + 4), m0 + wm * WTM, n0 + wn * WTN, lane);
    } else if (lds_epi) {
        // stage the wave tile through the (now idle) operand LDS so that stores are 16 bytes per lane over whole row segments
        constexpr int LDS_HALFS = 2 * (BM + BN) * 64;
        constexpr int TMC = (NW * WTM * (WTN + 4) <= LDS_HALFS) ? TM : TM / 2;
        static_assert(NW * TMC * 16 * (WTN + 4) <= LDS_HALFS, "epilogue staging does not fit");
        __syncthreads();
        gemm_epilogue_lds<TM, TN, TMC, WSTATS>(p, acc, smem + wave * (TMC * 16) * (WTN + 4), m0 + wm * WTM, n0 + wn * WTN, lane, 0, 0, up_phase);
    } else {
        gemm_epilogue<TM, TN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, l15, lg, 0, 0, up_phase);
    }
}

// sum the split-K slabs in a fixed order and apply the epilogue
__global__ void splitk_reduce_kernel(fd_gemm_desc p, int nsplit) {
    FD_WG_TRACE(3);
    const int64_t n4 = (int64_t)p.M * p.N / 4;
    const float* ws = (const float*)p.workspace;
    const f16* R = (const f16*)p.residual;
    const f16* RB = (const f16*)p.rowbias;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 4;
        const int m = (int)(e / p.N), n = (int)(e - (int64_t)m * p.N);
        f32x4 a = *(const f32x4*)(ws + e);
        for (int s = 1; s < nsplit; ++s) a += *(const f32x4*)(ws + (int64_t)s * p.M * p.N + e);
        const int rbrow = RB ? m / p.rows_per_batch : 0;
        float v[4];
        const float al = n < p.colscale_cols ? p.alpha * p.colscale : p.alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = a[r] * al;
            if (p.bias) x += p.bias[n + r];
            if (RB) x += (float)RB[(int64_t)rbrow * p.ld_rowbias + n + r];
            x = apply_act(x, p.act);
            if (R) x += (float)R[(int64_t)m * p.ldr + n + r];
            v[r] = x;
        }
        if (p.out_dtype == FD_OUT_F32) *(f32x4*)((float*)p.C + (int64_t)m * p.ldc + n) = (f32x4){v[0], v[1], v[2], v[3]};
        else *(f16x4*)((f16*)p.C + (int64_t)m * p.ldc + n) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    }
}

static int launch_splitk_reduce(const fd_gemm_desc& d, hipStream_t s, int nsplit) {
    int64_t blocks = ((int64_t)d.M * d.N / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, d, nsplit);
    return fd_check_launch("fd_gemm(split-K reduce)");
}

template <int BM, int BN, int WGM, int WGN>
static int launch_big(const fd_gemm_desc& d, hipStream_t s, int nsplit = 1) {
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    constexpr size_t lds = (size_t)2 * (BM + BN) * 64 * sizeof(f16);
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)gemm_big_kernel<BM, BN, WGM, WGN, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gemm_big_kernel<BM, BN, WGM, WGN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gemm_big_kernel<BM, BN, WGM, WGN, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    // n-tiles per band: the band's B slab (gn * BN rows of K halfs, per split) should fit an XCD's L2 next to the streaming A tiles
    static const long l2_budget = bench_env("FD_GEMM_L2_KB") ? atol(bench_env("FD_GEMM_L2_KB")) * 1024 : 3 * 1024 * 1024;
    const long ktot = ((long)d.K + d.K2) / nsplit;
    const int nph = (d.conv && (d.conv_mode == FD_CONV_UP2P || d.conv_mode == FD_CONV_UP2PI)) ? 4 : 1;
    long gnl = l2_budget / ((long)BN * ktot * 2);
    const int gn = (int)(gnl < 1 ? 1 : (gnl > ntn ? ntn : gnl));
    if constexpr (BN / WGN == 80) {
        if (d.gn_stats && nsplit == 1 && d.conv && d.conv_mode == FD_CONV_UP2PI) {       // the phase pair writing the interleaved result, with statistics
            static std::once_flag once_st6;
            std::call_once(once_st6, [] {
                (void)hipFuncSetAttribute((const void*)gemm_big_kernel<BM, BN, WGM, WGN, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            });
            hipLaunchKernelGGL((gemm_big_kernel<BM, BN, WGM, WGN, 6>), dim3(ntm * ntn * nph, 1), dim3(WGM * WGN * 64), lds, s, d, ntm, ntn, gn);
            return fd_check_launch("fd_gemm(big, phase pair with statistics epilogue)");
        }
        if (d.gn_stats && nsplit == 1 && !(d.conv && d.conv_mode >= FD_CONV_UP2P)) {     // the statistics-epilogue instantiations (fd_gemm checked eligibility)
            static std::once_flag once_st;
            std::call_once(once_st, [] {
                (void)hipFuncSetAttribute((const void*)gemm_big_kernel<BM, BN, WGM, WGN, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                (void)hipFuncSetAttribute((const void*)gemm_big_kernel<BM, BN, WGM, WGN, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            });
            if (d.conv) hipLaunchKernelGGL((gemm_big_kernel<BM, BN, WGM, WGN, 4>), dim3(ntm * ntn, 1), dim3(WGM * WGN * 64), lds, s, d, ntm, ntn, gn);
            else hipLaunchKernelGGL((gemm_big_kernel<BM, BN, WGM, WGN, 3>), dim3(ntm * ntn, 1), dim3(WGM * WGN * 64), lds, s, d, ntm, ntn, gn);
            return fd_check_launch("fd_gemm(big, statistics epilogue)");
        }
    }
    if (d.conv && d.conv_mode >= FD_CONV_UP2P)
        hipLaunchKernelGGL((gemm_big_kernel<BM, BN, WGM, WGN, 2>), dim3(ntm * ntn * nph, nsplit), dim3(WGM * WGN * 64), lds, s, d, ntm, ntn, gn);
    else if (d.conv) hipLaunchKernelGGL((gemm_big_kernel<BM, BN, WGM, WGN, 1>), dim3(ntm * ntn, nsplit), dim3(WGM * WGN * 64), lds, s, d, ntm, ntn, gn);
    else hipLaunchKernelGGL((gemm_big_kernel<BM, BN, WGM, WGN, 0>), dim3(ntm * ntn, nsplit), dim3(WGM * WGN * 64), lds, s, d, ntm, ntn, gn);
    if (nsplit > 1) return launch_splitk_reduce(d, s, nsplit);
    return fd_check_launch("fd_gemm(big)");
}


// ======================================================================================= skinny N (LoRA down-projections)
// C[M, N <= 64] = A[M,K] . B[N,K]^T with nothing else in the epilogue: t = x.down^T of every LoRALinearLayer (N = rank padded to
// 8, or 3 stacked ranks for q/k/v) and dt = g.up of its backward.  4149 launches per training step took 3.7 % of it on the 64x64
// tile at 43 TFLOP/s; the problem is a pure stream over A.  Here a wave owns 16 rows: it loads its A fragments straight from
// global memory in MFMA layout (16 B per lane, 64 B contiguous per row and k-step), B (a few KB, L1-resident) likewise; no LDS,
// no barrier in the main loop.  KS waves of a workgroup split K in slabs of 320 and are summed through LDS in a fixed order.
template <int NT, int KS, int RT>
__global__ __launch_bounds__(KS * 64) void gemm_skinny_kernel(fd_gemm_desc p) {
    FD_WG_TRACE(4);
    __shared__ float red[KS > 1 ? (KS - 1) * RT * NT * 256 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const f16* A[RT];
    bool mok[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int m = (blockIdx.x * RT + r) * 16 + l15;
        mok[r] = m < p.M;
        A[r] = (const f16*)p.A + (int64_t)(mok[r] ? m : 0) * p.lda + lg * 8;
    }
    const f16* B = (const f16*)p.B + lg * 8;
    f32x4 acc[RT][NT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[r][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nks = p.K >> 5;                                  // k-steps of 32
    const int per = (nks + KS - 1) / KS;                       // k-steps per wave
    const int kbeg = wave * per, kend = min(nks, kbeg + per);
    constexpr int CH = 10 / RT;                                // k-steps in flight per wave (40 VGPRs of A)
    for (int k0 = kbeg; k0 < kend; k0 += CH) {
        f16x8 af[RT][CH];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                af[r][c] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (k0 + c < kend && mok[r]) af[r][c] = *(const f16x8*)(A[r] + (int64_t)(k0 + c) * 32);
            }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (k0 + c < kend) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int n = j * 16 + l15;
                    f16x8 bf = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    if (n < p.N) bf = *(const f16x8*)(B + (int64_t)n * p.ldb + (int64_t)(k0 + c) * 32);   // one B fragment feeds RT row tiles
#pragma unroll
                    for (int r = 0; r < RT; ++r) acc[r][j] = FD_MFMA_16x16x32(bf, af[r][c], acc[r][j]);
                }
            }
        }
    }
    if (KS > 1) {
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int j = 0; j < NT; ++j) *(f32x4*)(red + ((((wave - 1) * RT + r) * NT + j) * 64 + lane) * 4) = acc[r][j];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 1; w < KS; ++w)
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[r][j] += *(const f32x4*)(red + ((((w - 1) * RT + r) * NT + j) * 64 + lane) * 4);
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (!mok[r]) continue;
        f16* C = (f16*)p.C + (int64_t)((blockIdx.x * RT + r) * 16 + l15) * p.ldc;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = j * 16 + lg * 4;                    // lane holds C[m][n..n+3] (swapped-operand MFMA layout)
            if (n < p.N) *(f16x4*)(C + n) = (f16x4){(f16)acc[r][j][0], (f16)acc[r][j][1], (f16)acc[r][j][2], (f16)acc[r][j][3]};
        }
    }
}

static bool skinny_ok(const fd_gemm_desc& d) {
    static const bool off = bench_env("FD_GEMM_NOSKINNY") != nullptr;   // A/B switch for measurement
    // rocprofv3 kernel times, hot caches (scratch/prof_skinny.sh), this kernel vs the 64x64 / 128x64 tiles: M=4096 K=1280: 5.2 vs 14.6 us,
    // M=16384 K=640: 6.8 vs 8.6, M=65536 K=320: 10.4 vs 10.3 at N=8 (both at the A stream's bandwidth) but 12.8 vs 10.7 at N=24, where
    // the B fragments re-read through L1 by every wave outweigh the A stream: wide-M problems with more than one column tile stay on the LDS tiles
    if (d.N > 16 && d.M >= 32768) return false;
    return !off && !d.conv && d.batch <= 1 && d.K2 == 0 && d.colscale_cols == 0 && d.N <= 64 && (d.N & 3) == 0 && (d.K & 31) == 0 && d.M >= 1024 && !d.bias && !d.rowbias &&
           !d.residual && d.act == FD_ACT_NONE && d.alpha == 1.f && d.out_dtype == FD_OUT_F16 && (d.ldc & 3) == 0;
}

template <int NT, int RT>
static int launch_skinny_rt(const fd_gemm_desc& d, hipStream_t s) {
    const int blocks = (d.M + 16 * RT - 1) / (16 * RT);
    const int ks = d.K >= 1280 ? 4 : d.K >= 640 ? 2 : 1;     // slabs of >= 320 per wave
    if (ks == 4) hipLaunchKernelGGL((gemm_skinny_kernel<NT, 4, RT>), dim3(blocks), dim3(256), 0, s, d);
    else if (ks == 2) hipLaunchKernelGGL((gemm_skinny_kernel<NT, 2, RT>), dim3(blocks), dim3(128), 0, s, d);
    else hipLaunchKernelGGL((gemm_skinny_kernel<NT, 1, RT>), dim3(blocks), dim3(64), 0, s, d);
    return fd_check_launch("fd_gemm(skinny)");
}

template <int NT>
static int launch_skinny(const fd_gemm_desc& d, hipStream_t s) {
    // measurement switch: two row tiles per wave share each B fragment (13.4 us on the N=24 case above: still behind the LDS tile)
    static const int rt = bench_env("FD_GEMM_SKINNY_RT") ? atoi(bench_env("FD_GEMM_SKINNY_RT")) : 0;
    const bool two = rt == 2;
    return two ? launch_skinny_rt<NT, 2>(d, s) : launch_skinny_rt<NT, 1>(d, s);
}

template <int BM, int BN>
static int launch(const fd_gemm_desc& d, hipStream_t s) {
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    dim3 grid(ntm * ntn, d.batch > 0 ? d.batch : 1);
    if (d.conv) hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, true>), grid, dim3(256), 0, s, d, ntm, ntn);
    else hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, false>), grid, dim3(256), 0, s, d, ntm, ntn);
    return fd_check_launch("fd_gemm");
}

// gemm_pp.hip: the 8-wave 256x320 ping-pong kernel (BK = 32, four-stage ring, two wave groups half a k-step apart)
bool fd_gemm_pp_eligible(const fd_gemm_desc& d);
int fd_gemm_launch_pp(const fd_gemm_desc& d, hipStream_t s, bool prio, int bm, int nsplit);
bool fd_conv_halo_takes(const fd_gemm_desc& d, int bm, int nsplit);     // gemm_halo.hip: stride-1 3x3 convolutions with the halo-staged A operand
// (round 5: the persistent streaming form of the 128x320 ping-pong kernel -- policy bit 128, measured and never selected -- and the register-B experiments
// live in scratch/ with their measurements: gemm_pps_experiment.hip, gemm_rb_experiment.hip, gemm_rbk_experiment.hip)
// Which problems the ping-pong kernels take (bits): 1 = stride-1 3x3 convolutions on the 256x320 tile, 2 = every dense GEMM on it,
// 4 = s_setprio around the MFMA streams, 8 = stride-1 convolutions on the 128x320 tile, 16 = dense GEMMs on it, 32 = dense 256x320 GEMMs
// with K <= 384 and N >= 2560 only (the FF1 projections of the 64^2 level), 64 = split-K launches of the 128x320 tile too.
// Measured on the step's shapes against the lockstep kernels (profiles/r03_gemm_pingpong_ab_v2.txt, isolated launches, us):
//   conv 640 -> 320 @64^2 250 -> 207, 960 -> 320 361 -> 296, 320 -> 320 122 -> 110; conv 1280 -> 640 @32^2 359 -> 288, 1920 -> 640 524 -> 415;
//   conv 1280 -> 1280 @16^2 (128x320) 223 -> 191, 2560 -> 1280 429 -> 359 (+17..26 %); FF1 @64^2 206 -> 185 (+10 %);
//   s_setprio is worth 3-9 % of that (it does nothing in a lockstep loop, as the guide says).
// Left on the lockstep kernels: the other dense shapes (short K loops: four waves per SIMD hide more than the role split returns,
// 65536x320x1280 65 vs 76 us; 128x320 dense within +-3 %) and the split-K launches of the 8^2 level (45 k-steps per slice: 54 vs 61 us).
// Whole step, same box, bench-hooks library (profiles/r03_step_ab_pingpong.txt): 1497 ms without, 1453-1469 ms with.
#ifndef FD_GEMM_PP_DEFAULT
#define FD_GEMM_PP_DEFAULT (1 | 4 | 8 | 32)
#endif
static int pp_mode() {
#ifdef FD_BENCH_HOOKS
    const char* e = getenv("FD_GEMM_PP");      // measurement build: re-read on every call so one process can A/B the variants
    return e ? atoi(e) : FD_GEMM_PP_DEFAULT;
#else
    return FD_GEMM_PP_DEFAULT;
#endif
}
// 0 = not taken, else the tile height (256 / 128)
static int pp_takes(const fd_gemm_desc& d, int sel) {
    const int t = sel % 1000000, split = sel / 1000000, m = pp_mode();
    if (!m || !fd_gemm_pp_eligible(d)) return 0;
    if (t == 256320 && split == 0) {
        if (d.conv) return (m & 1) ? 256 : 0;
        if ((m & 2) || ((m & 32) && d.K + d.K2 <= 384 && d.N >= 2560)) return 256;
        return 0;
    }
    if (t == 128320 && (split == 0 || (m & 64))) return (m & (d.conv ? 8 : 16)) ? 128 : 0;
    return 0;
}

// tile choice (BM*1000+BN): big tiles when the grid still fills 256 CUs a few times over
static int gemm_tile(const fd_gemm_desc& d);
extern "C" int fd_gemm_tile(const fd_gemm_desc* dp) {
    FD_REQUIRE_DESC(dp, fd_gemm_desc, "fd_gemm_tile");
    return gemm_tile(*dp);
}
static int gemm_tile(const fd_gemm_desc& d) {
    const long nb = d.batch > 1 ? d.batch : 1;
    static const bool nobig = bench_env("FD_GEMM_NOBIG") != nullptr;
    static const int force = bench_env("FD_GEMM_FORCE") ? atoi(bench_env("FD_GEMM_FORCE")) : 0;   // measurement only: force a big-tile code for dense
    if (skinny_ok(d)) return 16000 + (d.N + 15) / 16 * 16;      // 16 rows per wave x N padded to 16: the LoRA down-projections
    if (force && nb == 1 && !d.conv) return force;
    // big-tile (BK=64, 8-wave) variants: unbatched, K-tiles of 64 must not straddle a conv tap
    static const int bigk = bench_env("FD_GEMM_BIGK") ? atoi(bench_env("FD_GEMM_BIGK")) : 320;
    if (!nobig && nb == 1 && (d.conv ? (d.Cin & 63) == 0 : (d.K + d.K2) >= bigk)) {
        const long phs = (d.conv && (d.conv_mode == FD_CONV_UP2P || d.conv_mode == FD_CONV_UP2PI)) ? 4 : 1;    // four phases share the launch
        const long m256 = phs * ((d.M + 255) / 256), m128 = phs * ((d.M + 127) / 128);
        // Minimum tile counts for the 320-wide tiles.  Round 1 tuned them per kernel in isolation (200 / 160: "fill 256 CUs"); under the
        // multi-stream schedule of round 2 a launch that fills half the chip with efficient tiles beats one that fills it with smaller
        // tiles or split-K partials, because another stream's launch takes the other half.  Whole-step A/B on one box
        // (profiles/r02_tile_policy_ab.txt): 200/160 -> 1594-1606 ms, 128/100 -> 1548, 100/80 -> 1533-1538, 64/48 -> 1627, 32/32 -> 1663.
        static const long t256 = bench_env("FD_GEMM_T256") ? atol(bench_env("FD_GEMM_T256")) : 100;
        static const long t128 = bench_env("FD_GEMM_T128") ? atol(bench_env("FD_GEMM_T128")) : 80;
        static const long maxsplit = bench_env("FD_GEMM_MAXSPLIT") ? atol(bench_env("FD_GEMM_MAXSPLIT")) : 8;
        static const long tc256 = bench_env("FD_CONV_T256") ? atol(bench_env("FD_CONV_T256")) : 0;      // measurement only: its own threshold for the stride-1 3x3 convolutions
        if (d.N % 320 == 0) {
            if (m256 * (d.N / 320) >= ((tc256 && d.conv && d.conv_mode == FD_CONV_NORMAL) ? tc256 : t256)) return 256320;
            if (m128 * (d.N / 320) >= t128) return 128320;
        }
        const bool can_split = d.workspace && (d.N & 3) == 0 && (d.ldc & 3) == 0 && d.act != FD_ACT_GEGLU && phs == 1;
        const long nk = (d.K + 63) / 64 + (d.K2 + 63) / 64;
        auto split_for = [&](long blocks, int tilecode) -> int {   // split K so ~256 blocks exist, >= 8 k-tiles each
            long split = blocks > 0 ? 256 / blocks : 1;
            if (split > maxsplit) split = maxsplit;
            while (split > 1 && (nk / split < 8 || (int64_t)split * d.M * d.N * 4 > d.workspace_bytes)) --split;
            return split > 1 ? (int)(split * 1000000 + tilecode) : 0;
        };
        // mid-size M with long K (16x16 level): the 128x320 tile split over K beats 128x160 without a split (intensity 91 vs 71 FLOP/B)
        if (can_split && nk >= 64 && d.N % 320 == 0 && m128 * (d.N / 320) >= 32) {
            const int c = split_for(m128 * (d.N / 320), 128320);
            if (c) return c;
        }
        static const long t160 = bench_env("FD_GEMM_T160") ? atol(bench_env("FD_GEMM_T160")) : 160;
        if (d.N % 160 == 0 && m128 * (d.N / 160) >= t160) return 128160;
        // VAE decoder widths (256 / 512 channels at 256^2 / 512^2 pixels): 256x256 halves the A re-reads of the 256x128 tile
        static const bool no256 = bench_env("FD_GEMM_NO256256") != nullptr;
        static const long tvae = bench_env("FD_GEMM_TVAE") ? atol(bench_env("FD_GEMM_TVAE")) : 200;
        if (!no256 && d.N % 256 == 0 && m256 * (d.N / 256) >= tvae) return 256256;
        static const bool no512 = bench_env("FD_GEMM_NO512128") != nullptr;
        if (!no512 && d.N == 128 && ((d.M + 511) / 512) >= 400) return 512128;   // 128-channel layers at 512^2: the tallest tile that fits the LDS
        if (d.N % 128 == 0 && m256 * (d.N / 128) >= tvae) return 256128;
        // small-M, long-K (8x8 level): split K so that all 256 CUs get a block
        if (can_split && d.N % 160 == 0) {
            const int c = split_for(m128 * (d.N / 160), 128160);
            if (c) return c;
        }
    }
    const long t128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128) * nb;
    const bool n64 = (d.N % 128) != 0 && (d.N % 128) <= 64;  // e.g. N=320: 128x64 tiles waste nothing
    if (t128 >= 512 && !n64) return 128128;
    const long t12864 = (long)((d.M + 127) / 128) * ((d.N + 63) / 64) * nb;
    if (t12864 >= 512) return 128064;
    return 64064;
}

// ONE dispatch decision shared by fd_gemm, fd_gemm_kernel_name and fd_gemm_stats_rows (the launcher and the name function had drifted
// apart once): which kernel family takes the problem, with which tile / wave grid / split-K factor, and whether that kernel can write gn_stats.
enum { GK_SKINNY, GK_GLDS, GK_BIG, GK_PP };
struct GemmPlan { int kind, bm, bn, wgm, wgn, nsplit, cv; bool stats_ok; };
static GemmPlan gemm_plan(const fd_gemm_desc& d /* K2 already normalised */) {
    GemmPlan g = {};
    const int sel = gemm_tile(d), t = sel % 1000000;
    static const bool w16 = bench_env("FD_GEMM_W8") == nullptr;   // 16-wave variants by default (A/B switch for measurement)
    g.nsplit = sel >= 1000000 ? sel / 1000000 : 1;
    g.bm = t / 1000; g.bn = t % 1000;
    g.cv = d.conv ? (d.conv_mode >= FD_CONV_UP2P ? 2 : 1) : 0;
    // order: ping-pong kernel (split-K slices too under policy bit 64), then the lockstep tiles
    if (const int bm = pp_takes(d, sel)) { g.kind = GK_PP; g.bm = bm; g.bn = 320; g.wgm = 2; g.wgn = 4; }
    else if (g.nsplit > 1) { g.kind = GK_BIG; g.wgm = 4; g.wgn = t == 128320 ? 4 : 2; }
    else switch (t) {
        case 256320: g.kind = GK_BIG; g.wgm = (w16 && !d.conv) ? 4 : 2; g.wgn = 4; break;
        case 128320: g.kind = GK_BIG; g.wgm = w16 ? 4 : 2; g.wgn = 4; break;
        case 128160: g.kind = GK_BIG; g.wgm = 4; g.wgn = 2; break;
        case 256256: g.kind = GK_BIG; g.wgm = 2; g.wgn = 4; break;
        case 512128: g.kind = GK_BIG; g.wgm = 8; g.wgn = 2; break;
        case 256128: g.kind = GK_BIG; g.wgm = 4; g.wgn = 2; break;
        default: g.kind = g.bm == 16 ? GK_SKINNY : GK_GLDS; break;
    }
    // the statistics epilogue lives in gemm_epilogue_lds<..., true>: 80-column wave tiles, fp16 LDS-staged epilogue, no split-K (its epilogue runs in
    // splitk_reduce_kernel), not the phase-major output of the up-sampling pair (a chunk there is not a run of pixels of one image)
    const bool lds_epi = d.out_dtype == FD_OUT_F16 && (d.N & 7) == 0 && (d.ldc & 7) == 0 && (!d.residual || (d.ldr & 7) == 0) &&
                         (!d.rowbias || (d.ld_rowbias & 3) == 0);
    g.stats_ok = (g.kind == GK_PP || (g.kind == GK_BIG && g.bn / g.wgn == 80)) && g.nsplit == 1 && lds_epi && d.act != FD_ACT_GEGLU &&
                 d.batch <= 1 && (d.N % 80) == 0 && (g.cv < 2 || (d.conv_mode == FD_CONV_UP2PI && (d.M & 31) == 0));
    return g;
}

// Name of the kernel fd_gemm launches for this problem, spelled as rocprofv3 --kernel-trace prints it (template arguments included), so
// that bench.py's live HIP-event roofline and the committed profiles/ summaries key the same thing.  Split-K launches add a
// ``splitk_reduce_kernel`` that the bench's events bracket together with the GEMM.
extern "C" int fd_gemm_kernel_name(const fd_gemm_desc* dp, char* buf, int n) {
    FD_REQUIRE_DESC(dp, fd_gemm_desc, "fd_gemm_kernel_name");
    fd_gemm_desc d = *dp;
    if (d.K2 <= 0 || !d.A2) d.K2 = 0;
    const GemmPlan g = gemm_plan(d);
    const bool st = d.gn_stats && g.stats_ok;
    switch (g.kind) {
        case GK_PP:
            if (fd_conv_halo_takes(d, g.bm, g.nsplit)) snprintf(buf, n, "conv_halo_kernel<%d, %d, %d, %s>", g.bm, d.W, st ? 3 : 1, (pp_mode() & 4) ? "true" : "false");
            else snprintf(buf, n, "gemm_pp_kernel<%d, %d, %s>", g.bm, (d.conv ? 1 : 0) + (st ? 2 : 0), (pp_mode() & 4) ? "true" : "false");
            break;
        case GK_BIG: snprintf(buf, n, "gemm_big_kernel<%d, %d, %d, %d, %d>", g.bm, g.bn, g.wgm, g.wgn, st ? (g.cv == 2 ? 6 : g.cv + 3) : g.cv); break;
        case GK_SKINNY: snprintf(buf, n, "gemm_skinny_kernel<%d, %d, 1>", g.bn / 16, d.K >= 1280 ? 4 : d.K >= 640 ? 2 : 1); break;
        default: snprintf(buf, n, "gemm_glds_kernel<%d, %d, %s>", g.bm, g.bn, d.conv ? "true" : "false"); break;
    }
    return g.nsplit > 1 ? g.nsplit : 0;   // split-K factor (0 = none)
}

// Rows per chunk of fd_gemm_desc.gn_stats: 32 for every kernel that has the statistics epilogue (gemm_epilogue_lds<..., true> forms the sums in
// canonical 32-row chunks whatever its tile), 0 when the kernel fd_gemm would launch has none.
extern "C" int fd_gemm_stats_rows(const fd_gemm_desc* dp) {
    FD_REQUIRE_DESC(dp, fd_gemm_desc, "fd_gemm_stats_rows");
    fd_gemm_desc d = *dp;
    if (d.K2 <= 0 || !d.A2) d.K2 = 0;
    return gemm_plan(d).stats_ok ? 32 : 0;
}

extern "C" int fd_gemm(const fd_gemm_desc* dp, void* stream) {
    FD_REQUIRE_DESC(dp, fd_gemm_desc, "fd_gemm");
    fd_gemm_desc d = *dp;
#ifdef FD_BENCH_HOOKS
    static const char* dbg = bench_env("FD_GEMM_DBG");   // measurement only: 1 = no loads after the first tile, 2 = no MFMAs
    if (dbg && d.batch <= 1) d.batch = -atoi(dbg);
#endif
    FD_REQUIRE(d.A && d.B && d.C, "fd_gemm: null operand");
    FD_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "fd_gemm: bad shape M=%d N=%d K=%d", d.M, d.N, d.K);
    FD_REQUIRE((d.K & 7) == 0 && (d.lda & 7) == 0 && (d.ldb & 7) == 0, "fd_gemm: K, lda, ldb must be multiples of 8");
    if (d.K2 > 0) {
        FD_REQUIRE(!d.conv, "fd_gemm: second K-slab not supported with conv gather");
        FD_REQUIRE(d.A2 && d.B2 && (d.K2 & 7) == 0 && (d.lda2 & 7) == 0 && (d.ldb2 & 7) == 0, "fd_gemm: bad second slab");
        FD_REQUIRE(d.batch <= 1, "fd_gemm: second slab is not batched");
    } else d.K2 = 0;
    if (d.conv) {
        const bool up_fwd = d.conv_mode == FD_CONV_UP2P || d.conv_mode == FD_CONV_UP2PI;
        const int ntap = up_fwd ? 4 : d.conv_mode == FD_CONV_UP2P_BWD ? 16 : 9;
        FD_REQUIRE((d.Cin & 31) == 0 && d.K == ntap * d.Cin, "fd_gemm(conv): Cin must be a multiple of 32 and K == taps*Cin");
        if (up_fwd) FD_REQUIRE(!d.residual && !d.rowbias, "fd_gemm(conv up2 phases): bias-only epilogue");
        if (d.conv_mode == FD_CONV_UP2PI) FD_REQUIRE(d.out_dtype == FD_OUT_F16, "fd_gemm(conv up2 phases, interleaved result): fp16 output");
        FD_REQUIRE(d.M == d.Bn * d.Ho * d.Wo, "fd_gemm(conv): M != B*Ho*Wo");
        FD_REQUIRE(d.batch <= 1, "fd_gemm(conv): not batched");
        FD_REQUIRE((int64_t)d.Bn * d.H * d.W * d.lda < (1LL << 31), "fd_gemm(conv): input larger than 2^31 elements");
    }
    if (d.rowbias) FD_REQUIRE(d.rows_per_batch > 0, "fd_gemm: rows_per_batch");
    FD_REQUIRE(d.colscale_cols >= 0 && (d.colscale_cols & 3) == 0 && (d.colscale_cols == 0 || (d.act != FD_ACT_GEGLU && d.batch <= 1)),
               "fd_gemm: colscale_cols=%d must be a multiple of 4 (not with GEGLU / batched launches)", d.colscale_cols);
    if (d.act == FD_ACT_GEGLU)
        // in this mode ``residual`` is an optional second OUTPUT [M, N] (ldr): the pre-gate projection, interleaved like B
        FD_REQUIRE(!d.conv && d.batch <= 1 && d.out_dtype == FD_OUT_F16 && !d.rowbias && d.alpha == 1.f && (d.N & 15) == 0 &&
                       (d.ldc & 7) == 0 && d.K2 == 0 && (!d.residual || (d.ldr & 7) == 0),
                   "fd_gemm(GEGLU): needs a plain fp16 GEMM with N %% 16 == 0 and ldc, ldr %% 8 == 0");
    hipStream_t s = (hipStream_t)stream;
    const GemmPlan g = gemm_plan(d);
    if (d.gn_stats) FD_REQUIRE(g.stats_ok, "fd_gemm: gn_stats set but the kernel for M=%d N=%d K=%d has no statistics epilogue (ask fd_gemm_stats_rows first)",
                               d.M, d.N, d.K);
    if (d.conv && d.conv_mode >= FD_CONV_UP2P)
        FD_REQUIRE(g.kind == GK_BIG, "fd_gemm(conv up2 phases): shape not taken by the big-tile kernels (Cin %% 64, enough tiles); use FD_CONV_UP2");
    switch (g.kind) {
        case GK_PP: {
            const int rc = fd_gemm_launch_pp(d, s, (pp_mode() & 4) != 0, g.bm, g.nsplit);
            if (rc != 0 || g.nsplit == 1) return rc;
            return launch_splitk_reduce(d, s, g.nsplit);
        }
        case GK_SKINNY:
            switch (g.bn) {
                case 16: return launch_skinny<1>(d, s);
                case 32: return launch_skinny<2>(d, s);
                case 48: return launch_skinny<3>(d, s);
                default: return launch_skinny<4>(d, s);
            }
        case GK_BIG:
            switch (g.bm * 1000 + g.bn) {
                case 256320: return g.wgm == 4 ? launch_big<256, 320, 4, 4>(d, s, g.nsplit) : launch_big<256, 320, 2, 4>(d, s, g.nsplit);
                case 128320: return g.wgm == 4 ? launch_big<128, 320, 4, 4>(d, s, g.nsplit) : launch_big<128, 320, 2, 4>(d, s, g.nsplit);
                case 128160: return launch_big<128, 160, 4, 2>(d, s, g.nsplit);
                case 256256: return launch_big<256, 256, 2, 4>(d, s, g.nsplit);
                case 512128: return launch_big<512, 128, 8, 2>(d, s, g.nsplit);
                default: return launch_big<256, 128, 4, 2>(d, s, g.nsplit);
            }
        default:
            if (g.bm == 128 && g.bn == 128) return launch<128, 128>(d, s);
            if (g.bm == 128 && g.bn == 64) return launch<128, 64>(d, s);
            return launch<64, 64>(d, s);
    }
}

FD_WGT_SETTER(gemm)
