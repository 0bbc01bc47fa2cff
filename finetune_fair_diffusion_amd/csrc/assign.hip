// Device solver for the Monte-Carlo optimal-transport targets of the multi-attribute experiments (exp-3-debias-gender-race/
// 1-main-debias.py:1488-1536, exp-4 :1517-1569): for each of S draws of balanced cell counts the reference calls
// ``ot.emd(ones(N), counts, M)`` -- N unit-mass faces onto K cells with integer capacities that sum to N.  With unit sources and integer
// sinks the LP has an integral optimum, i.e. it is a linear assignment problem on the capacity-replicated cost matrix (column j stands
// for one seat of cell cell[j]).
//
// One 64-lane wave solves one draw with the shortest-augmenting-path (Hungarian, Jonker-Volgenant potentials) algorithm in fp64: rows
// are inserted one at a time; every step of the path search is a lane-parallel relaxation of the column slack ``minv`` and a wave-wide
// argmin.  All per-column state lives in LDS (N <= 1024 seats), the cost matrix (N x K doubles) is read through the cache.  The S
// draws are independent workgroups; a face's seat is added into the summed plan with a float atomic (integer-valued sums <= S: exact
// and order-independent).  Latency-bound by design (S = 100 waves on 256 CUs, ~N^2 dependent steps each): it exists to take the solve
// off the host, not to fill the chip.
#include "common.h"

namespace {

struct ArgMin {
    double v;
    int j;
};

__device__ __forceinline__ ArgMin wave_argmin(ArgMin a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(a.v, o, 64);
        const int oj = __shfl_xor(a.j, o, 64);
        if (ov < a.v || (ov == a.v && oj < a.j)) {
            a.v = ov;
            a.j = oj;
        }
    }
    return a;
}

__global__ __launch_bounds__(64) void ot_assign_kernel(const double* __restrict__ cost, const int* __restrict__ counts, float* plan,
                                                       int* seats /* [S,N] cell of face i in draw s, or null */, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_raw[];
    const int n1 = N + 1;
    double* u = (double*)sh_raw;          // row potentials, 1-based
    double* v = u + n1;                   // seat potentials, 1-based (0 = the virtual start column)
    double* minv = v + n1;
    int* p = (int*)(minv + n1);           // p[j] = row seated at j (0 = free)
    int* way = p + n1;
    int* cell = way + n1;                 // cell of seat j
    int* used = cell + n1;
    const int lane = threadIdx.x;
    const int* cnt = counts + (size_t)blockIdx.x * K;
    const double INF = 1e300;

    for (int j = lane; j <= N; j += 64) {
        u[j] = 0.0;
        v[j] = 0.0;
        p[j] = 0;
        way[j] = 0;
        int c = 0, acc = 0;
        if (j >= 1) {
            for (int k = 0; k < K; ++k) {     // seat j (1-based) belongs to the first cell whose cumulative capacity reaches j
                acc += cnt[k];
                if (j <= acc) {
                    c = k;
                    break;
                }
            }
        }
        cell[j] = c;
    }
    __syncthreads();

    for (int i = 1; i <= N; ++i) {
        for (int j = lane; j <= N; j += 64) {
            minv[j] = INF;
            used[j] = 0;
        }
        if (lane == 0) p[0] = i;
        __syncthreads();
        int j0 = 0;
        while (true) {
            if (lane == 0) used[j0] = 1;
            __syncthreads();
            const int i0 = p[j0];
            const double ui = u[i0];
            const double* crow = cost + (size_t)(i0 - 1) * K;
            ArgMin best = {INF, 0x7fffffff};
            for (int j = lane + 1; j <= N; j += 64) {
                if (used[j]) continue;
                const double cur = crow[cell[j]] - ui - v[j];
                double mv = minv[j];
                if (cur < mv) {
                    mv = cur;
                    minv[j] = cur;
                    way[j] = j0;
                }
                if (mv < best.v) {      // strict: the first minimal seat of this lane; lanes are merged towards the smaller index
                    best.v = mv;
                    best.j = j;
                }
            }
            best = wave_argmin(best);
            const double delta = best.v;
            const int j1 = best.j;
            __syncthreads();
            for (int j = lane; j <= N; j += 64) {
                if (used[j]) {
                    u[p[j]] += delta;       // seated rows are distinct: no two lanes touch one u
                    v[j] -= delta;
                } else {
                    minv[j] -= delta;
                }
            }
            __syncthreads();
            j0 = j1;
            if (p[j0] == 0) break;
        }
        if (lane == 0) {                    // flip the alternating path back to the virtual column
            int jj = j0;
            while (jj) {
                const int jp = way[jj];
                p[jj] = p[jp];
                jj = jp;
            }
        }
        __syncthreads();
    }
    for (int j = lane + 1; j <= N; j += 64) {
        const int row = p[j] - 1;
        atomicAdd(plan + (size_t)row * K + cell[j], 1.0f);
        if (seats) seats[(size_t)blockIdx.x * N + row] = cell[j];
    }
}

}  // namespace

extern "C" int fd_ot_assign_sum(const double* cost, const int32_t* counts, float* plan, int32_t* seats, int N, int K, int S, void* stream) {
    FD_REQUIRE(N >= 1 && N <= 1024 && K >= 1 && K <= 4096 && S >= 1, "fd_ot_assign_sum: N=%d (1..1024) K=%d S=%d", N, K, S);
    const size_t lds = (size_t)(N + 1) * (3 * sizeof(double) + 4 * sizeof(int));
    hipLaunchKernelGGL(ot_assign_kernel, dim3(S), dim3(64), lds, (hipStream_t)stream, cost, (const int*)counts, plan, (int*)seats, N, K);
    return fd_check_launch("fd_ot_assign_sum");
}
