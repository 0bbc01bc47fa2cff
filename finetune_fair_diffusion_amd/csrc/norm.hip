// GroupNorm (channels-last, 2-source concat, fused SiLU) and LayerNorm, forward and backward.
// HBM-bound kernels: 16-byte fp16 vectors, fp32 statistics, fixed reduction order (deterministic).
#include "common.h"

#define GN_MAX_CHUNKS 64

// A/B switches exist only in the bench-hooks build (make BENCH_HOOKS=1), as in gemm_device.h
#ifdef FD_BENCH_HOOKS
#include <cstdlib>
static inline const char* bench_env(const char* name) { return getenv(name); }
#else
static inline const char* bench_env(const char*) { return nullptr; }
#endif

struct GNArgs {
    const f16* x1; const f16* x2; int C1, C2;
    const f16* dy;
    int B, HW, G, rows_per_chunk, nchunks;
    float eps;
    const float* gamma; const float* beta; const float* mean_rstd;
    int silu;
};

__device__ __forceinline__ f16x8 gn_load(const GNArgs& a, int64_t pix, int c) {
    if (c < a.C1) return *(const f16x8*)(a.x1 + pix * a.C1 + c);
    return *(const f16x8*)(a.x2 + pix * a.C2 + (c - a.C1));
}

// MODE 0: per-group (sum x, sum x^2). MODE 1: per-group (sum dz*gamma, sum dz*gamma*xhat)
template <int MODE>
__global__ void gn_reduce_kernel(GNArgs a, float* partial /* [B,nchunks,G,2] */) {
    FD_WG_TRACE(11);
    extern __shared__ float part[];  // [blockDim.x][16]: per-thread per-channel partial sums (fixed-order reduction below)
    const int C = a.C1 + a.C2, V = C >> 3, cg = C / a.G;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int v = threadIdx.x % V, rsub = threadIdx.x / V, rpb = blockDim.x / V;
    const int c0 = v * 8;
    float s0[8], s1[8];
    float gm[8], bt[8], mu[8], rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s0[j] = s1[j] = 0.f;
        if (MODE == 1) {
            const int g = (c0 + j) / cg;
            gm[j] = a.gamma[c0 + j];
            bt[j] = a.beta[c0 + j];
            mu[j] = a.mean_rstd[(b * a.G + g) * 2];
            rs[j] = a.mean_rstd[(b * a.G + g) * 2 + 1];
        }
    }
    const int r0 = chunk * a.rows_per_chunk;
    const int r1 = min(r0 + a.rows_per_chunk, a.HW);
    // GN_U rows per thread in flight: with one 16-byte load per loop trip a CU holds ~15 KB of requests, below what the HBM latency x
    // bandwidth product needs (the two-launch kernels ran at ~2 TB/s); issuing the loads of GN_U rows before touching any of them
    // multiplies the bytes in flight.  The accumulation order per thread (row by row) is unchanged: same bits as before.
    constexpr int GN_U = 4;
    for (int rb = r0 + rsub; rb < r1; rb += rpb * GN_U) {
        f16x8 xs[GN_U], ds[GN_U];
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int r = rb + u * rpb;
            if (r < r1) {
                const int64_t pix = (int64_t)b * a.HW + r;
                xs[u] = gn_load(a, pix, c0);
                if (MODE == 1) ds[u] = *(const f16x8*)(a.dy + pix * C + c0);
            }
        }
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int r = rb + u * rpb;
            if (r >= r1) break;
            const f16x8 xv = xs[u];
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = (float)xv[j];
                    s0[j] += x;
                    s1[j] += x * x;
                }
            } else {
                const f16x8 dv = ds[u];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xv[j] - mu[j]) * rs[j];
                    float dz = (float)dv[j];
                    if (a.silu) dz *= silu_grad_f(xh * gm[j] + bt[j]);
                    const float t = dz * gm[j];
                    s0[j] += t;
                    s1[j] += t * xh;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        part[threadIdx.x * 16 + j] = s0[j];
        part[threadIdx.x * 16 + 8 + j] = s1[j];
    }
    __syncthreads();
    // one thread per group sums its channels over all row sub-groups in a fixed order (bit-reproducible, no atomics)
    if (threadIdx.x < a.G) {
        const int g = threadIdx.x;
        float t0 = 0.f, t1 = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; ++c) {
            const int tv = c >> 3, j = c & 7;
            for (int rr = 0; rr < rpb; ++rr) {
                t0 += part[(rr * V + tv) * 16 + j];
                t1 += part[(rr * V + tv) * 16 + 8 + j];
            }
        }
        partial[((int64_t)(b * a.nchunks + chunk)) * a.G * 2 + g * 2] = t0;
        partial[((int64_t)(b * a.nchunks + chunk)) * a.G * 2 + g * 2 + 1] = t1;
    }
}

// every apply block reduces the (<= 64) chunk partials of its image itself: saves a separate finalize launch per GroupNorm
template <int MODE>
__device__ __forceinline__ void gn_block_stats(const GNArgs& a, const float* partial, int b, float n, float* st /* LDS [G*2] */,
                                               float* out /* global [B,G,2] or nullptr */) {
    if (threadIdx.x < a.G) {
        const int g = threadIdx.x;
        float a0 = 0.f, a1 = 0.f;
        for (int c = 0; c < a.nchunks; ++c) {
            a0 += partial[((int64_t)(b * a.nchunks + c) * a.G + g) * 2];
            a1 += partial[((int64_t)(b * a.nchunks + c) * a.G + g) * 2 + 1];
        }
        float v0, v1;
        if (MODE == 0) {
            v0 = a0 / n;
            v1 = rsqrtf(fmaxf(a1 / n - v0 * v0, 0.f) + a.eps);
        } else {
            v0 = a0 / n;
            v1 = a1 / n;
        }
        st[g * 2] = v0;
        st[g * 2 + 1] = v1;
        if (out) { out[(b * a.G + g) * 2] = v0; out[(b * a.G + g) * 2 + 1] = v1; }
    }
    __syncthreads();
}

// Statistics that arrive from the PRODUCERS of x1 / x2 (fd_gemm_desc.gn_stats): per row chunk and 10-channel unit the (sum, sum of squares) of
// the stored values.  A group is a run of units; the (unit, chunk) pairs of a group, unit-major, are cut into blockDim / G slices that are summed
// by different threads and combined in slice order -- the geometry depends on (C, HW, G, chunk heights) only: bit-reproducible.
// per1 / per2 > 0: the source's chunks are phase-major (FD_CONV_UP2PI): chunk c of image b is row (c / per) * (B * per) + b * per + c % per of the table
struct GNUnits { const float* st1; const float* st2; int nch1, nch2, per1, per2; };

__device__ __forceinline__ void gn_block_stats_units(const GNArgs& a, const GNUnits& u, int b, float n, float* st /* LDS [G*2] */,
                                                     float* ps /* LDS [blockDim / G][G][2] */, float* out /* global [B,G,2] or nullptr */) {
    const int C = a.C1 + a.C2, upg = C / a.G / 10, U1 = a.C1 / 10, U2 = a.C2 / 10;
    const int nparts = blockDim.x / a.G;
    const int g = threadIdx.x % a.G, part = threadIdx.x / a.G;
    if (part < nparts) {
        int L = 0;
        for (int k = 0; k < upg; ++k) L += (g * upg + k < U1) ? u.nch1 : u.nch2;
        const int lo = L * part / nparts, hi = L * (part + 1) / nparts;
        float a0 = 0.f, a1 = 0.f;
        int base = 0;
        for (int k = 0; k < upg; ++k) {
            const int uu = g * upg + k;
            const bool first = uu < U1;
            const float* sp = first ? u.st1 : u.st2;
            const int nch = first ? u.nch1 : u.nch2, U = first ? U1 : U2, ul = first ? uu : uu - U1, per = first ? u.per1 : u.per2;
            const int c0 = max(lo - base, 0), c1 = min(hi - base, nch);
            for (int c = c0; c < c1; ++c) {
                const int64_t row = per > 0 ? (int64_t)(c / per) * (a.B * per) + b * per + c % per : (int64_t)b * nch + c;
                const float2 v = *(const float2*)(sp + (row * U + ul) * 2);
                a0 += v.x;
                a1 += v.y;
            }
            base += nch;
        }
        ps[(part * a.G + g) * 2] = a0;
        ps[(part * a.G + g) * 2 + 1] = a1;
    }
    __syncthreads();
    if (threadIdx.x < a.G) {
        float a0 = 0.f, a1 = 0.f;
        for (int q = 0; q < nparts; ++q) {
            a0 += ps[(q * a.G + g) * 2];
            a1 += ps[(q * a.G + g) * 2 + 1];
        }
        const float v0 = a0 / n;
        const float v1 = rsqrtf(fmaxf(a1 / n - v0 * v0, 0.f) + a.eps);
        st[g * 2] = v0;
        st[g * 2 + 1] = v1;
        if (out) { out[(b * a.G + g) * 2] = v0; out[(b * a.G + g) * 2 + 1] = v1; }
    }
    __syncthreads();
}

template <bool UNITS>
__global__ void gn_apply_kernel(GNArgs a, f16* y, const float* partial, float n, float* stats_out, GNUnits un) {
    FD_WG_TRACE(12);
    __shared__ float st[128];
    __shared__ float ps[UNITS ? 1024 : 1];
    const int C = a.C1 + a.C2, V = C >> 3, cg = C / a.G;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int v = threadIdx.x % V, rsub = threadIdx.x / V, rpb = blockDim.x / V;
    if (UNITS) gn_block_stats_units(a, un, b, n, st, ps, chunk == 0 ? stats_out : nullptr);
    else gn_block_stats<0>(a, partial, b, n, st, chunk == 0 ? stats_out : nullptr);
    if (rsub >= rpb) return;
    const int c0 = v * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (c0 + j) / cg;
        const float mu = st[g * 2], rs = st[g * 2 + 1];
        sc[j] = rs * a.gamma[c0 + j];
        sh[j] = a.beta[c0 + j] - mu * sc[j];
    }
    const int r0 = chunk * a.rows_per_chunk;
    const int r1 = min(r0 + a.rows_per_chunk, a.HW);
    constexpr int GN_U = 4;      // rows in flight per thread (see gn_reduce_kernel)
    for (int rb = r0 + rsub; rb < r1; rb += rpb * GN_U) {
        f16x8 xs[GN_U];
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int r = rb + u * rpb;
            if (r < r1) xs[u] = gn_load(a, (int64_t)b * a.HW + r, c0);
        }
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int r = rb + u * rpb;
            if (r >= r1) break;
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float z = (float)xs[u][j] * sc[j] + sh[j];
                if (a.silu) z = silu_f(z);
                o[j] = (f16)z;
            }
            *(f16x8*)(y + ((int64_t)b * a.HW + r) * C + c0) = o;
        }
    }
}

__global__ void gn_bwd_apply_kernel(GNArgs a, const float* partial, float n, const f16* add1, const f16* add2, f16* dx1, f16* dx2) {
    FD_WG_TRACE(13);
    __shared__ float s12[128];
    const int C = a.C1 + a.C2, V = C >> 3, cg = C / a.G;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int v = threadIdx.x % V, rsub = threadIdx.x / V, rpb = blockDim.x / V;
    gn_block_stats<1>(a, partial, b, n, s12, nullptr);
    if (rsub >= rpb) return;
    const int c0 = v * 8;
    float gm[8], bt[8], mu[8], rs[8], m1[8], m2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (c0 + j) / cg;
        gm[j] = a.gamma[c0 + j];
        bt[j] = a.beta[c0 + j];
        mu[j] = a.mean_rstd[(b * a.G + g) * 2];
        rs[j] = a.mean_rstd[(b * a.G + g) * 2 + 1];
        m1[j] = s12[g * 2];
        m2[j] = s12[g * 2 + 1];
    }
    const bool first = c0 < a.C1;
    const int cc = first ? c0 : c0 - a.C1;
    const int Cs = first ? a.C1 : a.C2;
    f16* dxp = first ? dx1 : dx2;
    const f16* addp = first ? add1 : add2;
    const int r0 = chunk * a.rows_per_chunk;
    const int r1 = min(r0 + a.rows_per_chunk, a.HW);
    constexpr int GN_U = 2;      // rows in flight per thread: three streams (x, dy, add) per row
    for (int rb = r0 + rsub; rb < r1; rb += rpb * GN_U) {
        f16x8 xs[GN_U], ds[GN_U], as[GN_U];
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int r = rb + u * rpb;
            as[u] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (r < r1) {
                const int64_t pix = (int64_t)b * a.HW + r;
                xs[u] = gn_load(a, pix, c0);
                ds[u] = *(const f16x8*)(a.dy + pix * C + c0);
                if (addp) as[u] = *(const f16x8*)(addp + pix * Cs + cc);
            }
        }
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int r = rb + u * rpb;
            if (r >= r1) break;
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = ((float)xs[u][j] - mu[j]) * rs[j];
                float dz = (float)ds[u][j];
                if (a.silu) dz *= silu_grad_f(xh * gm[j] + bt[j]);
                const float d = rs[j] * (dz * gm[j] - m1[j] - xh * m2[j]) + (float)as[u][j];
                o[j] = (f16)d;
            }
            if (dxp) *(f16x8*)(dxp + ((int64_t)b * a.HW + r) * Cs + cc) = o;
        }
    }
}

static int gn_geometry(int C, int HW, int& threads, int& rows_per_chunk, int& nchunks) {
    const int V = C / 8;
    if (V > 1024) return -1;
    const int rpb = V >= 256 ? 1 : (256 / V);
    threads = V * rpb;
    nchunks = HW < GN_MAX_CHUNKS * rpb ? (HW + rpb - 1) / rpb : GN_MAX_CHUNKS;
    if (nchunks < 1) nchunks = 1;
    rows_per_chunk = (HW + nchunks - 1) / nchunks;
    nchunks = (HW + rows_per_chunk - 1) / rows_per_chunk;
    return 0;
}

static int gn_check(int C1, int C2, int groups) {
    const int C = C1 + C2;
    if ((C1 & 7) || (C2 & 7) || C <= 0 || groups <= 0 || groups > 64 || C % groups) return -1;
    return 0;
}

// ------------------------------------------------------------------ single-launch GroupNorm for the small feature maps
// At the 32^2 / 16^2 / 8^2 levels a GroupNorm is 3-40 MB of traffic: the two-launch form above (reduce, then apply with the statistics
// finalised in its prologue) spends most of its 22-45 us on launch and dependency latency, not on HBM.  Here one workgroup owns
// (sample b, channel block): the smallest run of whole groups that is a multiple of 8 channels (40 channels = 1 group at C = 1280, 2 at 640;
// 80 at 2560; 120 = 2 groups at 1920), reads its [HW, CB] slab ONCE into registers (forward) or streams it twice through L2 (backward),
// reduces in a fixed order (per-thread channel sums -> LDS -> one wave per group -> shuffle tree: bit-reproducible, no atomics) and writes
// the result.  Chosen by (C, HW, groups) only -- never by the batch size -- so a sample's result does not depend on what it is batched with.
struct GNFused { int CB, VB, rpb, nv, ngb; };

static bool gn_fused_geometry(int C, int HW, int G, int maxv, GNFused& f) {
    static const bool off = bench_env("FD_GN_NOFUSED") != nullptr;        // A/B switch of the bench-hooks build
    const int cg = C / G;
    if (off || cg < 8) return false;
    int CB = cg;
    while (CB & 7) CB += cg;
    if (C % CB || CB / cg > 4 || CB / 8 > 16) return false;
    f.CB = CB; f.VB = CB / 8; f.rpb = 256 / f.VB; f.nv = (HW + f.rpb - 1) / f.rpb; f.ngb = CB / cg;
    return f.nv <= maxv;
}

// the (at most two) groups of this block that the 8 channels starting at block-local channel ``lc`` belong to
__device__ __forceinline__ void gn_vec_groups(int lc, int cg, int& ga, int& gb) { ga = lc / cg; gb = (lc + 7) / cg; }

// sums the per-thread (A0, A1, B0, B1) group partials left in ``red`` into (sum0, sum1) of block-local group ``w`` -- called by wave w
__device__ __forceinline__ void gn_group_total(const float* red, int w, int lane, int VB, int nact, int cg, float& t0, float& t1) {
    t0 = t1 = 0.f;
    for (int t = lane; t < nact; t += 64) {
        int ga, gb;
        gn_vec_groups((t % VB) * 8, cg, ga, gb);
        if (ga == w) { t0 += red[t * 4]; t1 += red[t * 4 + 1]; }
        if (gb == w && gb != ga) { t0 += red[t * 4 + 2]; t1 += red[t * 4 + 3]; }
    }
    t0 = wave_sum(t0);
    t1 = wave_sum(t1);
}

template <int MAXV>
__global__ __launch_bounds__(256) void gn_fused_fwd_kernel(GNArgs a, f16* y, float* stats_out, GNFused f) {
    FD_WG_TRACE(14);
    __shared__ float red[256 * 4];
    __shared__ float st[8];
    const int C = a.C1 + a.C2, cg = C / a.G;
    const int b = blockIdx.y, cb0 = blockIdx.x * f.CB;
    const int nact = f.VB * f.rpb;
    const bool act = threadIdx.x < nact;
    const int v = threadIdx.x % f.VB, rsub = threadIdx.x / f.VB;
    const int lc = v * 8, c0 = cb0 + lc;
    f16x8 xv[MAXV];
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int r = rsub + i * f.rpb;
        if (act && i < f.nv && r < a.HW) {
            xv[i] = gn_load(a, (int64_t)b * a.HW + r, c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = (float)xv[i][j];
                s0[j] += x;
                s1[j] += x * x;
            }
        }
    }
    int ga, gb;
    gn_vec_groups(lc, cg, ga, gb);
    float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool first = (lc + j) / cg == ga;
        p[first ? 0 : 2] += s0[j];
        p[first ? 1 : 3] += s1[j];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x * 4 + k] = p[k];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < f.ngb) {
        float t0, t1;
        gn_group_total(red, wave, lane, f.VB, nact, cg, t0, t1);
        if (lane == 0) {
            const float n = (float)a.HW * (float)cg;
            const float mu = t0 / n, rs = rsqrtf(fmaxf(t1 / n - mu * mu, 0.f) + a.eps);
            st[wave * 2] = mu;
            st[wave * 2 + 1] = rs;
            if (stats_out) {
                const int g = cb0 / cg + wave;
                stats_out[(b * a.G + g) * 2] = mu;
                stats_out[(b * a.G + g) * 2 + 1] = rs;
            }
        }
    }
    __syncthreads();
    if (!act) return;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (lc + j) / cg;
        sc[j] = st[g * 2 + 1] * a.gamma[c0 + j];
        sh[j] = a.beta[c0 + j] - st[g * 2] * sc[j];
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int r = rsub + i * f.rpb;
        if (i < f.nv && r < a.HW) {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float z = (float)xv[i][j] * sc[j] + sh[j];
                if (a.silu) z = silu_f(z);
                o[j] = (f16)z;
            }
            *(f16x8*)(y + ((int64_t)b * a.HW + r) * C + c0) = o;
        }
    }
}

__global__ __launch_bounds__(256) void gn_fused_bwd_kernel(GNArgs a, const f16* add1, const f16* add2, f16* dx1, f16* dx2, GNFused f) {
    FD_WG_TRACE(15);
    __shared__ float red[256 * 4];
    __shared__ float st[8];
    const int C = a.C1 + a.C2, cg = C / a.G;
    const int b = blockIdx.y, cb0 = blockIdx.x * f.CB;
    const int nact = f.VB * f.rpb;
    const bool act = threadIdx.x < nact;
    const int v = threadIdx.x % f.VB, rsub = threadIdx.x / f.VB;
    const int lc = v * 8, c0 = cb0 + lc;
    float gm[8], bt[8], mu[8], rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (c0 + j) / cg;
        gm[j] = a.gamma[c0 + j];
        bt[j] = a.beta[c0 + j];
        mu[j] = a.mean_rstd[(b * a.G + g) * 2];
        rs[j] = a.mean_rstd[(b * a.G + g) * 2 + 1];
    }
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.f;
    if (act)
        for (int r = rsub; r < a.HW; r += f.rpb) {
            const int64_t pix = (int64_t)b * a.HW + r;
            const f16x8 xv = gn_load(a, pix, c0);
            const f16x8 dv = *(const f16x8*)(a.dy + pix * C + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = ((float)xv[j] - mu[j]) * rs[j];
                float dz = (float)dv[j];
                if (a.silu) dz *= silu_grad_f(xh * gm[j] + bt[j]);
                const float t = dz * gm[j];
                s0[j] += t;
                s1[j] += t * xh;
            }
        }
    int ga, gb;
    gn_vec_groups(lc, cg, ga, gb);
    float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool first = (lc + j) / cg == ga;
        p[first ? 0 : 2] += s0[j];
        p[first ? 1 : 3] += s1[j];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x * 4 + k] = p[k];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < f.ngb) {
        float t0, t1;
        gn_group_total(red, wave, lane, f.VB, nact, cg, t0, t1);
        if (lane == 0) {
            const float n = (float)a.HW * (float)cg;
            st[wave * 2] = t0 / n;
            st[wave * 2 + 1] = t1 / n;
        }
    }
    __syncthreads();
    if (!act) return;
    float m1[8], m2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (lc + j) / cg;
        m1[j] = st[g * 2];
        m2[j] = st[g * 2 + 1];
    }
    const bool first = c0 < a.C1;
    const int cc = first ? c0 : c0 - a.C1;
    const int Cs = first ? a.C1 : a.C2;
    f16* dxp = first ? dx1 : dx2;
    const f16* addp = first ? add1 : add2;
    if (!dxp) return;
    for (int r = rsub; r < a.HW; r += f.rpb) {          // second sweep over the slab the workgroup has just read: served by L2
        const int64_t pix = (int64_t)b * a.HW + r;
        const f16x8 xv = gn_load(a, pix, c0);
        const f16x8 dv = *(const f16x8*)(a.dy + pix * C + c0);
        f16x8 av = {0, 0, 0, 0, 0, 0, 0, 0};
        if (addp) av = *(const f16x8*)(addp + pix * Cs + cc);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = ((float)xv[j] - mu[j]) * rs[j];
            float dz = (float)dv[j];
            if (a.silu) dz *= silu_grad_f(xh * gm[j] + bt[j]);
            o[j] = (f16)(rs[j] * (dz * gm[j] - m1[j] - xh * m2[j]) + (float)av[j]);
        }
        *(f16x8*)(dxp + pix * Cs + cc) = o;
    }
}

// GroupNorm forward: y = act(GN(x)); writes the (mean, rstd) it used to mean_rstd [B,groups,2] for the backward.
// scratch: B * 64 * groups * 2 floats.
extern "C" int fd_groupnorm_fwd(const void* x1, int C1, const void* x2, int C2, int B, int HW, int groups, float eps, const float* gamma,
                                const float* beta, int silu, void* y, float* mean_rstd, float* scratch, void* stream) {
    FD_REQUIRE(gn_check(C1, C2, groups) == 0, "fd_groupnorm_fwd: bad channels C1=%d C2=%d groups=%d", C1, C2, groups);
    GNArgs a = {};
    a.x1 = (const f16*)x1; a.x2 = (const f16*)x2; a.C1 = C1; a.C2 = C2; a.B = B; a.HW = HW; a.G = groups; a.eps = eps;
    a.gamma = gamma; a.beta = beta; a.silu = silu;
    int threads;
    FD_REQUIRE(gn_geometry(C1 + C2, HW, threads, a.rows_per_chunk, a.nchunks) == 0, "fd_groupnorm: C too large");
    hipStream_t s = (hipStream_t)stream;
    GNFused f;
    // measured, forward (reduce + apply -> one launch): 32x32x640 31 -> 21 us, 32x32x1280 44 -> 30, 16x16x1280 26 -> 16, 16x16x2560 38 -> 16, 8x8x1280 22 -> 15
    if (gn_fused_geometry(C1 + C2, HW, groups, 24, f)) {
        const dim3 grid((C1 + C2) / f.CB, B);
        if (f.nv <= 8) hipLaunchKernelGGL(gn_fused_fwd_kernel<8>, grid, dim3(256), 0, s, a, (f16*)y, mean_rstd, f);
        else hipLaunchKernelGGL(gn_fused_fwd_kernel<24>, grid, dim3(256), 0, s, a, (f16*)y, mean_rstd, f);
        return fd_check_launch("fd_groupnorm_fwd");
    }
    hipLaunchKernelGGL(gn_reduce_kernel<0>, dim3(a.nchunks, B), dim3(threads), threads * 16 * sizeof(float), s, a, scratch);
    const float n = (float)HW * (float)((C1 + C2) / groups);
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(a.nchunks, B), dim3(threads), 0, s, a, (f16*)y, (const float*)scratch, n, mean_rstd, GNUnits{});
    return fd_check_launch("fd_groupnorm_fwd");
}

// The apply pass alone, its statistics assembled from the producers' gn_stats buffers (the statistics pass of the two-launch
// form re-read x; here x is read once, by the pass that normalises it).
extern "C" int fd_groupnorm_fwd_stats_p(const void* x1, int C1, const void* x2, int C2, int B, int HW, int groups, float eps, const float* gamma,
                                        const float* beta, int silu, void* y, float* mean_rstd, const float* st1, int rows1, int per1,
                                        const float* st2, int rows2, int per2, void* stream);
extern "C" int fd_groupnorm_fwd_stats(const void* x1, int C1, const void* x2, int C2, int B, int HW, int groups, float eps, const float* gamma,
                                      const float* beta, int silu, void* y, float* mean_rstd, const float* st1, int rows1, const float* st2,
                                      int rows2, void* stream) {
    return fd_groupnorm_fwd_stats_p(x1, C1, x2, C2, B, HW, groups, eps, gamma, beta, silu, y, mean_rstd, st1, rows1, 0, st2, rows2, 0, stream);
}

extern "C" int fd_groupnorm_fwd_stats_p(const void* x1, int C1, const void* x2, int C2, int B, int HW, int groups, float eps, const float* gamma,
                                        const float* beta, int silu, void* y, float* mean_rstd, const float* st1, int rows1, int per1,
                                        const float* st2, int rows2, int per2, void* stream) {
    FD_REQUIRE(gn_check(C1, C2, groups) == 0, "fd_groupnorm_fwd_stats: bad channels C1=%d C2=%d groups=%d", C1, C2, groups);
    const int C = C1 + C2;
    FD_REQUIRE(C1 % 10 == 0 && C2 % 10 == 0 && (C / groups) % 10 == 0, "fd_groupnorm_fwd_stats: C1=%d, C2=%d and the group width %d must be multiples of 10",
               C1, C2, C / groups);
    FD_REQUIRE(st1 && rows1 > 0 && HW % rows1 == 0, "fd_groupnorm_fwd_stats: rows1=%d must divide HW=%d", rows1, HW);
    FD_REQUIRE(C2 == 0 || (x2 && st2 && rows2 > 0 && HW % rows2 == 0), "fd_groupnorm_fwd_stats: rows2=%d must divide HW=%d", rows2, HW);
    GNArgs a = {};
    a.x1 = (const f16*)x1; a.x2 = (const f16*)x2; a.C1 = C1; a.C2 = C2; a.B = B; a.HW = HW; a.G = groups; a.eps = eps;
    a.gamma = gamma; a.beta = beta; a.silu = silu;
    int threads;
    FD_REQUIRE(gn_geometry(C, HW, threads, a.rows_per_chunk, a.nchunks) == 0, "fd_groupnorm: C too large");
    FD_REQUIRE(threads / groups >= 1 && (threads / groups) * groups * 2 <= 1024, "fd_groupnorm_fwd_stats: %d threads for %d groups", threads, groups);
    FD_REQUIRE(per1 >= 0 && per2 >= 0 && (per1 == 0 || (HW / rows1) % per1 == 0) && (per2 == 0 || !C2 || (HW / rows2) % per2 == 0),
               "fd_groupnorm_fwd_stats: per1=%d / per2=%d must divide the chunks per image", per1, per2);
    GNUnits un = {st1, st2, HW / rows1, C2 ? HW / rows2 : 0, per1, C2 ? per2 : 0};
    const float n = (float)HW * (float)(C / groups);
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(a.nchunks, B), dim3(threads), 0, (hipStream_t)stream, a, (f16*)y, (const float*)nullptr, n,
                       mean_rstd, un);
    return fd_check_launch("fd_groupnorm_fwd_stats");
}

extern "C" int fd_groupnorm_bwd(const void* x1, int C1, const void* x2, int C2, const void* dy, int B, int HW, int groups,
                                const float* mean_rstd, const float* gamma, const float* beta, int silu, float* scratch,
                                const void* add1, const void* add2, void* dx1, void* dx2, void* stream) {
    FD_REQUIRE(gn_check(C1, C2, groups) == 0, "fd_groupnorm_bwd: bad channels");
    GNArgs a = {};
    a.x1 = (const f16*)x1; a.x2 = (const f16*)x2; a.C1 = C1; a.C2 = C2; a.B = B; a.HW = HW; a.G = groups;
    a.gamma = gamma; a.beta = beta; a.mean_rstd = mean_rstd; a.silu = silu; a.dy = (const f16*)dy;
    int threads;
    FD_REQUIRE(gn_geometry(C1 + C2, HW, threads, a.rows_per_chunk, a.nchunks) == 0, "fd_groupnorm: C too large");
    hipStream_t s = (hipStream_t)stream;
    GNFused f;
    // 16^2 and 8^2 only: at 32^2 (21 rows per thread, two sweeps) the single launch measured slower than the two-launch form
    // (32x32x1280: 74 vs 58 us; 16x16x1280: 22 vs 31 us, 16x16x2560: 28 vs 56 us, 8x8x1280: 13 vs 25 us)
    if (gn_fused_geometry(C1 + C2, HW, groups, 12, f)) {
        hipLaunchKernelGGL(gn_fused_bwd_kernel, dim3((C1 + C2) / f.CB, B), dim3(256), 0, s, a, (const f16*)add1, (const f16*)add2, (f16*)dx1,
                           (f16*)dx2, f);
        return fd_check_launch("fd_groupnorm_bwd");
    }
    hipLaunchKernelGGL(gn_reduce_kernel<1>, dim3(a.nchunks, B), dim3(threads), threads * 16 * sizeof(float), s, a, scratch);
    const float n = (float)HW * (float)((C1 + C2) / groups);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(a.nchunks, B), dim3(threads), 0, s, a, (const float*)scratch, n, (const f16*)add1,
                       (const f16*)add2, (f16*)dx1, (f16*)dx2);
    return fd_check_launch("fd_groupnorm_bwd");
}

// ------------------------------------------------------------------ LayerNorm: a wave owns R consecutive rows
// A row of C = 320 channels is 40 sixteen-byte vectors: one row per wave left 24 lanes idle, ONE load in flight per lane, and re-read gamma / beta
// (64 B per lane, four times the row data) for every row.  Here a wave issues the loads of R rows before it touches any of them (R x the bytes in
// flight) and keeps gamma / beta in registers across them.  The arithmetic of a row is unchanged -- lane-local sums over its own vectors, then
// the wave shuffle tree -- so the statistics are bit-identical to the one-row form (outputs too at C = 320; at wider C a few fp16 outputs move by
// one ulp with the compiler's FMA contraction).  32768 x 320: forward 41.8 -> 17.4 us, backward 26.4 -> 18.3 us;
// 16384 x 640: 41.9 -> 17.5 (profiles/r03_layernorm_rows_per_wave_ab.txt).  MAXV: vectors per lane (C <= 512 * MAXV).
template <bool BWD, int MAXV, int R>
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* x, const f16* dy, const float* gamma, const float* beta,
                                                        const f16* add, f16* out, float* mean_rstd, int M, int C, float eps) {
    FD_WG_TRACE(16);
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= M) return;
    const int V = C >> 3;
    float gm[MAXV][8], bt[MAXV][8];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int v = lane + i * 64;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gm[i][j] = v < V ? gamma[v * 8 + j] : 0.f;
            bt[i][j] = (!BWD && v < V) ? beta[v * 8 + j] : 0.f;
        }
    }
    f16x8 xv[R][MAXV], dv[BWD ? R : 1][MAXV];
#ifdef FD_LN_ZERO_INIT      // measurement build (scratch/r05_passes.sh h): rules "a lane reads an uninitialised register" in or out of the packed-fp32 hazard
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            xv[r][i] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (BWD) dv[r][i] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
#endif
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = lane + i * 64;
            if (row0 + r < M && v < V) {
                xv[r][i] = *(const f16x8*)(x + (int64_t)(row0 + r) * C + v * 8);
                if (BWD) dv[r][i] = *(const f16x8*)(dy + (int64_t)(row0 + r) * C + v * 8);
            }
        }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        if (row >= M) break;
        float s0 = 0.f, s1 = 0.f;
        if (!BWD) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int v = lane + i * 64;
                if (v < V) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) s0 += (float)xv[r][i][j];
                }
            }
            const float mean = wave_sum(s0) / C;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int v = lane + i * 64;
                if (v < V) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float d = (float)xv[r][i][j] - mean;
                        s1 += d * d;
                    }
                }
            }
            const float rstd = rsqrtf(wave_sum(s1) / C + eps);
            if (mean_rstd && lane == 0) {
                mean_rstd[row * 2] = mean;
                mean_rstd[row * 2 + 1] = rstd;
            }
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int v = lane + i * 64;
                if (v < V) {
                    f16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (f16)(((float)xv[r][i][j] - mean) * rstd * gm[i][j] + bt[i][j]);
                    *(f16x8*)(out + (int64_t)row * C + v * 8) = o;
                }
            }
        } else {
            const float mean = mean_rstd[row * 2], rstd = mean_rstd[row * 2 + 1];
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int v = lane + i * 64;
                if (v < V) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float t = (float)dv[r][i][j] * gm[i][j];
                        s0 += t;
                        s1 += t * ((float)xv[r][i][j] - mean) * rstd;
                    }
                }
            }
            const float m1 = wave_sum(s0) / C, m2 = wave_sum(s1) / C;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int v = lane + i * 64;
                if (v < V) {
                    f16x8 av = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (add) av = *(const f16x8*)(add + (int64_t)row * C + v * 8);
                    f16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xh = ((float)xv[r][i][j] - mean) * rstd;
                        const float t = (float)dv[r][i][j] * gm[i][j];
                        o[j] = (f16)(rstd * (t - m1 - xh * m2) + (float)av[j]);
                    }
                    *(f16x8*)(out + (int64_t)row * C + v * 8) = o;
                }
            }
        }
    }
}

template <bool BWD>
static void launch_layernorm(hipStream_t s, const f16* x, const f16* dy, const float* gamma, const float* beta, const f16* add, f16* out,
                             float* mean_rstd, int M, int C, float eps) {
#if defined(FD_LN_ONE_ROW)      // measurement: the one-row-per-wave form
    constexpr int R1 = 1, R2 = 1, R4 = 1;
#else
    // Both directions own 4 / 4 / 2 rows per wave.  (Round 3 shipped the backward at one row: its multi-row form made the three-stream
    // backward's gradients differ run to run.  Root cause, round 4: not a cross-stream race but packed-fp32 VALU code -- hipcc's SLP pass had
    // turned this kernel's per-element arithmetic into v_pk_{add,mul,fma}_f32, and those sequences returned wrong lanes whenever another
    // stream's kernel shared the SIMD; the library is built without packed-fp32 instructions since, csrc/Makefile.)
    constexpr int R1 = 4, R2 = 4, R4 = 2;
#endif
    if (C <= 512) hipLaunchKernelGGL((layernorm_kernel<BWD, 1, R1>), dim3((M + 4 * R1 - 1) / (4 * R1)), dim3(256), 0, s, x, dy, gamma, beta, add, out, mean_rstd, M, C, eps);
    else if (C <= 1024) hipLaunchKernelGGL((layernorm_kernel<BWD, 2, R2>), dim3((M + 4 * R2 - 1) / (4 * R2)), dim3(256), 0, s, x, dy, gamma, beta, add, out, mean_rstd, M, C, eps);
    else hipLaunchKernelGGL((layernorm_kernel<BWD, 4, R4>), dim3((M + 4 * R4 - 1) / (4 * R4)), dim3(256), 0, s, x, dy, gamma, beta, add, out, mean_rstd, M, C, eps);
}

extern "C" int fd_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd, int M, int C,
                                float eps, void* stream) {
    FD_REQUIRE((C & 7) == 0 && C <= 2048 && M > 0, "fd_layernorm_fwd: C=%d must be a multiple of 8 and <= 2048", C);
    launch_layernorm<false>((hipStream_t)stream, (const f16*)x, (const f16*)nullptr, gamma, beta, (const f16*)nullptr, (f16*)y, mean_rstd, M, C, eps);
    return fd_check_launch("fd_layernorm_fwd");
}

extern "C" int fd_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean_rstd, const void* add, void* dx,
                                int M, int C, void* stream) {
    FD_REQUIRE((C & 7) == 0 && C <= 2048 && M > 0 && mean_rstd, "fd_layernorm_bwd: bad args");
    launch_layernorm<true>((hipStream_t)stream, (const f16*)x, (const f16*)dy, gamma, (const float*)nullptr, (const f16*)add, (f16*)dx, (float*)mean_rstd, M, C, 0.f);
    return fd_check_launch("fd_layernorm_bwd");
}

FD_WGT_SETTER(norm)
