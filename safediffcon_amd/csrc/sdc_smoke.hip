// Smoke score check: the fluid rollout the reference runs on every sampled control
// sequence of the 2-D smoke task --
//   solver               2d/dataset/apps/evaluate_solver.py:209-350   (one Python process per sample, 2d/inference_2d.py:389-447)
//   get_envolve          2d/dataset/apps/evaluate_solver.py:82-111     control ring + last interior velocity -> projection
//   divergence_free      phi/flow.py:317-326 ; StaggeredGrid.divergence / .gradient  phi/math/nd.py:333-344,581-592
//   sparse_pressure_matrix + conjugate_gradient   phi/solver/sparse.py:27-77, phi/solver/base.py:63-103
//   StaggeredGrid.advect phi/math/nd.py:407-428 ; SciPyBackend.resample (scipy interpn, linear)  phi/math/scipy_backend.py:55-75
//   bucket book-keeping  2d/dataset/apps/evaluate_solver.py:114-178,262-336
// (phi = the PhiFlow 1.x copy vendored beside evaluate_solver.py.)
//
// One workgroup of 512 threads carries one sample through all per_timelength - 1 steps: the 127 x 127 pressure problem lives
// on chip for the whole rollout.  The reference's CG runs in float64 (numpy), stops on |r|_max < 1e-8 or after 500 iterations
// -- on this domain it always takes the 500 -- so a step is 500 x {5-point mat-vec, three dot products, three vector
// updates} on 16 129 unknowns: thread t owns column t & 127 of the 32 rows (t >> 7) * 32 ...; x, r and A p sit in its
// registers (3 x 32 doubles = 192 VGPRs; four vectors would fill the CU's whole 512 KB register file), the search direction p
// in LDS (129 rows x 128 doubles = 129 KB of the 160 KB; rows -1 and 127 and column 127 stay zero, and column 127 of one
// row is column -1 of the next, so the open border needs no halo columns) where the neighbours read it.  CG scalars never
// leave the device: wave partials by DPP lane moves, every thread then forms alpha and beta from the same eight LDS
// partials in a fixed order (a sample's result does not depend on the batch it rides in).  Element-wise arithmetic keeps
// the reference's operation order without FMA contraction (scipy's CSC mat-vec adds a row's terms in ascending column
// order; numpy multiplies, then adds); only the dot products and the bucket sums associate differently from numpy's
// pairwise sums.  The CG's first direction update reads the NEW residual on both sides, as the reference's aliased
// in-place update does.
// Velocity (128 x 128 x 2 doubles) and the three float32 density fields (ping-pong) sit in a per-sample global workspace
// that stays in L2; they are touched once per step, the CG 500 times.
// Cost (MI355X, tools/smoke_solver_probe.py): 4.6 us per CG iteration = 0.59 s per 256-step rollout, the same for 1 or 256
// samples in the batch (one CU each).  Per iteration a thread issues ~820 VALU instructions (two waves per SIMD: ~2.7 us
// if nothing else stalled) and moves 162 doubles through LDS; the two block-wide reductions cost 0.8 us of it.  A variant
// that let A p and p share registers (p re-read from LDS under the residual update) needs ~20 VGPRs more than the 256 a
// thread has at this occupancy and ran slower on its scratch spills (6.5 us).
#include "sdc_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int SN = 127;            // cells per side
constexpr int SS = 128;            // staggered samples per side
constexpr int NTHR = 512;
constexpr int RPT = 32;            // rows per thread
constexpr int LP = 128;            // LDS pitch of p (doubles): column 127 is always zero and doubles as column -1 of the next row
constexpr int NWAVE = NTHR / 64;
constexpr int GR = 4;              // rows per prefetch group of the mat-vec
constexpr int MAXB = 8;            // buckets per label map
constexpr size_t VEL_BYTES = (size_t)SS * SS * 2 * sizeof(double);
constexpr size_t DEN_FLOATS = (size_t)SN * SS;      // pitch 128
constexpr size_t WORK_BYTES = VEL_BYTES + 6 * DEN_FLOATS * sizeof(float);

struct SmokeArgs {
    const float* c1; const float* c2; int64_t c_sb, c_sf;       // controls (B, nt, nx, nx): batch / frame strides, rows dense
    const float* dens0; int64_t d_sb;                           // (B, nx, nx)
    const float* vel0; int64_t v_sb;                            // (128, 128, 2) staggered initial velocity, stride 0 = shared
    const unsigned char* fluid;                                 // (127, 127): 1 fluid, 0 obstacle
    const unsigned char* label_n; const unsigned char* label_s; // (128, 128): 0 = none, k = bucket k - 1
    double* out;                                                // (B, nt, 7, nx, nx)
    double* out_zero;                                           // (B, nt, nx, nx) or null
    char* work; int nt, nx, T, ti, si, lo, hi, max_iter, nb_n, nb_s;
    double accuracy;
};

// wave64 reductions on DPP lane moves (no LDS crossbar round trips): xor 1, xor 2, half-row mirror, row mirror leave
// the sum of each row of 16 in all its lanes; the four rows are combined through v_readlane in a fixed order.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_get(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wsum(double v) {
    v += dpp_mov<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);         // row_half_mirror
    v += dpp_mov<0x140>(v);         // row_mirror
    return ((lane_get(v, 0) + lane_get(v, 16)) + lane_get(v, 32)) + lane_get(v, 48);
}
__device__ __forceinline__ double wmax(double v) {
    v = fmax(v, dpp_mov<0xB1>(v));
    v = fmax(v, dpp_mov<0x4E>(v));
    v = fmax(v, dpp_mov<0x141>(v));
    v = fmax(v, dpp_mov<0x140>(v));
    return fmax(fmax(lane_get(v, 0), lane_get(v, 16)), fmax(lane_get(v, 32), lane_get(v, 48)));
}

// numpy's np.sum over n <= 8 doubles: n < 8 a plain loop, n == 8 the unrolled pairwise block
__device__ __forceinline__ double np_sum8(const double (&v)[MAXB], int n) {
    const double pw = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < MAXB; ++q) s = (q < n) ? s + v[q] : s;
    return n == 8 ? pw : s;
}

__global__ __launch_bounds__(NTHR) void smoke_rollout_kernel(SmokeArgs A) {
    extern __shared__ double P[];                    // (SS + 1) x LP: rows -1 .. 127 (rows -1 and 127, column 127 stay zero)
    __shared__ double redA[NWAVE * 2];
    __shared__ double redB[NWAVE * 2];
    __shared__ double redK[NWAVE * 2 * (MAXB + 1)];
    __shared__ double outs[2 * MAXB];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int j = t & 127, i0 = (t >> 7) * RPT;
    const int b = blockIdx.x;
    double* V = reinterpret_cast<double*>(A.work + (size_t)b * WORK_BYTES);
    float* D = reinterpret_cast<float*>(A.work + (size_t)b * WORK_BYTES + VEL_BYTES);
    const float* c1 = A.c1 + (int64_t)b * A.c_sb;
    const float* c2 = A.c2 + (int64_t)b * A.c_sb;
    const int nx = A.nx, si = A.si, ti = A.ti;
    double* outb = A.out + (int64_t)b * A.nt * 7 * nx * nx;
    double* outz = A.out_zero ? A.out_zero + (int64_t)b * A.nt * nx * nx : nullptr;
    double* Pc = P + (i0 + 1) * LP + j;              // this thread's first cell

    // ---- domain coefficients of this thread's 32 slots (phi/solver/sparse.py:27-77, phi/flow.py:455-474)
    auto fl = [&](int i, int jj) -> int {            // fluid mask padded with ones (open border: pad_fluid)
        return (i < 0 || i >= SN || jj < 0 || jj >= SN) ? 1 : (int)A.fluid[i * SN + jj];
    };
    unsigned actm = 0, vmxm = 0, vmym = 0;
    unsigned nd4[4] = {0, 0, 0, 0};                  // 4 bits per cell: -diag (1 .. 4)
    for (int k = 0; k < RPT; ++k) {
        const int i = i0 + k;
        const int in = (i < SN && j < SN);
        const int f = fl(i, j);
        if (in && f) actm |= 1u << k;
        int nd = fl(i + 1, j) + fl(i - 1, j) + fl(i, j + 1) + fl(i, j - 1);
        if (nd < 1) nd = 1;                          // minimum(centre, -1)
        nd4[k >> 3] |= (unsigned)nd << (4 * (k & 7));
        if (min(f, fl(i, j - 1))) vmxm |= 1u << k;
        if (min(f, fl(i - 1, j))) vmym |= 1u << k;
    }
    for (int q = t; q < (SS + 1) * LP; q += NTHR) P[q] = 0.0;

    // ---- initial fields
    {
        const float* v0 = A.vel0 + (int64_t)b * A.v_sb;
#pragma unroll 4
        for (int k = 0; k < RPT; ++k) {
            const int i = i0 + k;
            const float2 v = reinterpret_cast<const float2*>(v0)[i * SS + j];
            reinterpret_cast<double2*>(V)[i * SS + j] = make_double2((double)v.x, (double)v.y);
            if (i < SN) {
                const float d = (j < SN) ? A.dens0[(int64_t)b * A.d_sb + (i / si) * nx + (j / si)] : 0.f;
                D[0 * DEN_FLOATS + i * SS + j] = d;
                D[2 * DEN_FLOATS + i * SS + j] = d;
                D[4 * DEN_FLOATS + i * SS + j] = d;
            }
        }
    }
    if (t < 2 * MAXB) outs[t] = 0.0;                 // smoke_outs / smoke_outs_safe: uniform state, kept out of the VGPRs
    int cur = 0;                                     // density ping-pong: field f at D[(2 f + cur) * DEN_FLOATS]
    __syncthreads();

    // book-keeping of the two absorbing density copies + frame output (evaluate_solver.py:262-336); the fields at `cur`
    auto book_and_record = [&](int step) {
        float* dzp = D + (2 + cur) * DEN_FLOATS;
        float* dsp = D + (4 + cur) * DEN_FLOATS;
        double sn[MAXB + 1], ss[MAXB + 1];
#pragma unroll
        for (int q = 0; q <= MAXB; ++q) { sn[q] = 0.0; ss[q] = 0.0; }
        if (j < SN) {
            for (int k = 0; k < RPT; ++k) {
                const int i = i0 + k;
                if (i >= SN) break;
                const int ln = A.label_n[i * SS + j], ls = A.label_s[i * SS + j];
                const double vz = (double)dzp[i * SS + j], vs = (double)dsp[i * SS + j];
#pragma unroll
                for (int q = 0; q <= MAXB; ++q) {
                    sn[q] += (ln == q) ? vz : 0.0;
                    ss[q] += (ls == q) ? vs : 0.0;
                }
            }
        }
#pragma unroll
        for (int q = 0; q <= MAXB; ++q) {
            const double a = wsum(sn[q]), c = wsum(ss[q]);
            if (lane == 0) { redK[(wave * 2 + 0) * (MAXB + 1) + q] = a; redK[(wave * 2 + 1) * (MAXB + 1) + q] = c; }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q <= MAXB; ++q) {
            double a = 0.0, c = 0.0;
            for (int w = 0; w < NWAVE; ++w) { a += redK[(w * 2 + 0) * (MAXB + 1) + q]; c += redK[(w * 2 + 1) * (MAXB + 1) + q]; }
            sn[q] = a; ss[q] = c;
        }
        double cat_n = 0.0, cat_s = 0.0;
#pragma unroll
        for (int q = 1; q <= MAXB; ++q) {
            cat_n += (q <= A.nb_n) ? sn[q] : 0.0;
            cat_s += (q <= A.nb_s) ? ss[q] : 0.0;
        }
        const bool hit_n = cat_n > 0.0, hit_s = cat_s > 0.0;
        if (t == 0) {
#pragma unroll
            for (int q = 0; q < MAXB; ++q) {
                outs[q] += (hit_n && q < A.nb_n) ? sn[q + 1] : 0.0;
                outs[MAXB + q] += (hit_s && q < A.nb_s) ? ss[q + 1] : 0.0;
            }
        }
        if ((hit_n || hit_s) && j < SN) {
            for (int k = 0; k < RPT; ++k) {
                const int i = i0 + k;
                if (i >= SN) break;
                if (hit_n && A.label_n[i * SS + j]) dzp[i * SS + j] = 0.f;
                if (hit_s && A.label_s[i * SS + j]) dsp[i * SS + j] = 0.f;
            }
        }
        const double tot_n = hit_n ? sn[0] : sn[0] + cat_n;
        const double tot_s = hit_s ? ss[0] : ss[0] + cat_s;
        __syncthreads();                             // zeroed cells and outs visible; redK free again
        if (step % ti == 0) {
            double outs_n[MAXB], outs_s[MAXB];
#pragma unroll
            for (int q = 0; q < MAXB; ++q) { outs_n[q] = outs[q]; outs_s[q] = outs[MAXB + q]; }
            const double rec_n = outs_n[1] / (np_sum8(outs_n, A.nb_n) + tot_n);
            const double rec_s = outs_s[0] / (np_sum8(outs_s, A.nb_s) + tot_s);
            const int f = step / ti;
            double* of = outb + (int64_t)f * 7 * nx * nx;
            const float* dp = D + (0 + cur) * DEN_FLOATS;
            if (j % si == 0) {
                for (int k = 0; k < RPT; ++k) {
                    const int i = i0 + k;
                    if (i % si) continue;
                    const int o = (i / si) * nx + (j / si);
                    const bool in = (i < SN && j < SN);
                    of[0 * nx * nx + o] = in ? (double)dp[i * SS + j] : 0.0;
                    const double2 v = reinterpret_cast<const double2*>(V)[i * SS + j];
                    of[1 * nx * nx + o] = v.x;
                    of[2 * nx * nx + o] = v.y;
                    of[3 * nx * nx + o] = (double)c1[(int64_t)f * A.c_sf + o];
                    of[4 * nx * nx + o] = (double)c2[(int64_t)f * A.c_sf + o];
                    of[5 * nx * nx + o] = rec_n;
                    of[6 * nx * nx + o] = rec_s;
                    if (outz) outz[(int64_t)f * nx * nx + o] = in ? (double)dzp[i * SS + j] : 0.0;
                }
            }
        }
    };

    book_and_record(0);

    double x[RPT], r[RPT], Ap[RPT];
    for (int step = 0; step < A.T - 1; ++step) {
        // ---- get_envolve: control ring + last interior velocity, masked (evaluate_solver.py:92-106, phi/flow.py:300-304)
        {
            const int fr = step / ti;
            const float* c1f = c1 + (int64_t)fr * A.c_sf;
            const float* c2f = c2 + (int64_t)fr * A.c_sf;
            const bool jin = (j >= A.lo && j < A.hi);
#pragma unroll 4
            for (int k = 0; k < RPT; ++k) {
                const int i = i0 + k;
                double2 v = reinterpret_cast<double2*>(V)[i * SS + j];
                if (!(jin && i >= A.lo && i < A.hi)) {
                    const int o = (i / si) * nx + (j / si);
                    v.x = (double)c1f[o];
                    v.y = (double)c2f[o];
                }
                v.x = ((vmxm >> k) & 1) ? v.x : 0.0;
                v.y = ((vmym >> k) & 1) ? v.y : 0.0;
                reinterpret_cast<double2*>(V)[i * SS + j] = v;
            }
        }
        __syncthreads();
        // ---- divergence -> k ; CG start: x = 0, p = r = k
        double rmax;
        {
            double m = 0.0;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int i = i0 + k;
                double d = 0.0;
                if (i < SN && j < SN) {
                    const double2 vc = reinterpret_cast<const double2*>(V)[i * SS + j];
                    const double vyn = V[((i + 1) * SS + j) * 2 + 1];
                    const double vxn = V[(i * SS + j + 1) * 2 + 0];
                    d = (vyn - vc.y) + (vxn - vc.x);
                }
                r[k] = d;
                x[k] = 0.0;
                Pc[k * LP] = d;
                m = fmax(m, fabs(d));
            }
            m = wmax(m);
            if (lane == 0) redB[wave] = m;
            __syncthreads();
            rmax = redB[0];
            for (int w = 1; w < NWAVE; ++w) rmax = fmax(rmax, redB[w]);
        }
        // ---- conjugate_gradient (phi/solver/base.py:63-103)
        for (int it = 0; rmax >= A.accuracy && it < A.max_iter; ++it) {
            __syncthreads();                         // p (and the previous partials' readers) settled
            asm volatile("" : "+v"(nd4[0]), "+v"(nd4[1]), "+v"(nd4[2]), "+v"(nd4[3]), "+v"(actm));   // keep the per-cell
                                                     // coefficients packed: no hoisted float64 copies in VGPRs
            double s1 = 0.0, s2 = 0.0;
            {
                // mat-vec in groups of four rows, the next group's neighbours in flight while this one is computed
                const double* Pl = Pc - 1;
                const double* Pr = Pc + 1;
                double L[2][GR], R[2][GR], Dn[2][GR];
#pragma unroll
                for (int q = 0; q < GR; ++q) { L[0][q] = Pl[q * LP]; R[0][q] = Pr[q * LP]; Dn[0][q] = Pc[(q + 1) * LP]; }
                double up = Pc[-LP], c = Pc[0];
#pragma unroll
                for (int g = 0; g < RPT / GR; ++g) {
                    if (g + 1 < RPT / GR) {
#pragma unroll
                        for (int q = 0; q < GR; ++q) {
                            const int k = (g + 1) * GR + q;
                            L[(g + 1) & 1][q] = Pl[k * LP];
                            R[(g + 1) & 1][q] = Pr[k * LP];
                            Dn[(g + 1) & 1][q] = Pc[(k + 1) * LP];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < GR; ++q) {
                        const int k = g * GR + q;
                        const double dn = Dn[g & 1][q];
                        const double dk = (double)((nd4[k >> 3] >> (4 * (k & 7))) & 15u);
                        double y = up + L[g & 1][q];   // a row's terms in ascending column index: (i-1,j), (i,j-1), (i,j), (i,j+1), (i+1,j)
                        y = y - c * dk;
                        y = y + R[g & 1][q];
                        y = y + dn;
                        const int m = ((int)(actm << (31 - k))) >> 31;
                        y = __hiloint2double(__double2hiint(y) & m, __double2loint(y) & m);
                        Ap[k] = y;
                        s1 = __builtin_fma(c, y, s1);
                        s2 = __builtin_fma(c, r[k], s2);
                        up = c;
                        c = dn;
                    }
                }
            }
            s1 = wsum(s1);
            s2 = wsum(s2);
            if (lane == 0) { redA[wave * 2] = s1; redA[wave * 2 + 1] = s2; }
            __syncthreads();
            double tmp = 0.0, pr = 0.0;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) { tmp += redA[w * 2]; pr += redA[w * 2 + 1]; }
            const double a = pr / tmp;
            double s3 = 0.0, m = 0.0;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const double rn = r[k] - a * Ap[k];
                r[k] = rn;
                s3 = __builtin_fma(rn, Ap[k], s3);
                m = fmax(m, fabs(rn));
            }
            s3 = wsum(s3);
            m = wmax(m);
            if (lane == 0) { redB[wave * 2] = s3; redB[wave * 2 + 1] = m; }
            __syncthreads();
            double rAp = 0.0;
            rmax = 0.0;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) { rAp += redB[w * 2]; rmax = fmax(rmax, redB[w * 2 + 1]); }
            const double bb = -rAp / tmp;
            if (it == 0) {                           // the reference's first update reads the new residual as momentum
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    x[k] = x[k] + a * Pc[k * LP];
                    Pc[k * LP] = r[k] + bb * r[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const double po = Pc[k * LP];
                    x[k] = x[k] + a * po;
                    Pc[k * LP] = r[k] + bb * po;
                }
            }
        }
        // ---- velocity -= mask * gradient(pressure), masked again (phi/flow.py:317-326, evaluate_solver.py:109)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RPT; ++k) Pc[k * LP] = x[k];
        __syncthreads();
        {
            const int cj = min(j, SN - 1), cjm = min(max(j - 1, 0), SN - 1);
#pragma unroll 4
            for (int k = 0; k < RPT; ++k) {
                const int i = i0 + k;
                const int ci = min(i, SN - 1), cim = min(max(i - 1, 0), SN - 1);
                const double pc = P[(ci + 1) * LP + cj];
                const double gx = pc - P[(ci + 1) * LP + cjm];
                const double gy = pc - P[(cim + 1) * LP + cj];
                double2 v = reinterpret_cast<double2*>(V)[i * SS + j];
                v.x = ((vmxm >> k) & 1) ? v.x - gx : 0.0;
                v.y = ((vmym >> k) & 1) ? v.y - gy : 0.0;
                reinterpret_cast<double2*>(V)[i * SS + j] = v;
            }
        }
        __syncthreads();
        // ---- three semi-Lagrangian advections through the new velocity (phi/math/nd.py:407-428, scipy interpn linear)
        if (j < SN) {
            const float* s0 = D + (0 + cur) * DEN_FLOATS; float* d0 = D + (0 + (cur ^ 1)) * DEN_FLOATS;
            const float* s1p = D + (2 + cur) * DEN_FLOATS; float* d1 = D + (2 + (cur ^ 1)) * DEN_FLOATS;
            const float* s2p = D + (4 + cur) * DEN_FLOATS; float* d2 = D + (4 + (cur ^ 1)) * DEN_FLOATS;
#pragma unroll 2
            for (int k = 0; k < RPT; ++k) {
                const int i = i0 + k;
                if (i >= SN) break;
                const double2 vc = reinterpret_cast<const double2*>(V)[i * SS + j];
                const double cy = (V[((i + 1) * SS + j) * 2 + 1] + vc.y) / 2;
                const double cx = (V[(i * SS + j + 1) * 2 + 0] + vc.x) / 2;
                double yy = (double)i - cy, xx = (double)j - cx;
                yy = fmax(0.0, fmin((double)SN, yy));
                xx = fmax(0.0, fmin((double)SN, xx));
                const bool oob = (yy > (double)(SN - 1)) || (xx > (double)(SN - 1));
                int iy = (int)floor(yy), ix = (int)floor(xx);
                iy = min(max(iy, 0), SN - 2);
                ix = min(max(ix, 0), SN - 2);
                const double ty = yy - (double)iy, tx = xx - (double)ix;
                const double sy = 1.0 - ty, sx = 1.0 - tx;
                const double w00 = sy * sx, w01 = sy * tx, w10 = ty * sx, w11 = ty * tx;
                const int o = iy * SS + ix;
                auto samp = [&](const float* s) -> float {
                    double v = (double)s[o] * w00;
                    v = v + (double)s[o + 1] * w01;
                    v = v + (double)s[o + SS] * w10;
                    v = v + (double)s[o + SS + 1] * w11;
                    return oob ? 0.f : (float)v;
                };
                d0[i * SS + j] = samp(s0);
                d1[i * SS + j] = samp(s1p);
                d2[i * SS + j] = samp(s2p);
            }
        }
        cur ^= 1;
        __syncthreads();
        book_and_record(step + 1);
    }
}

std::atomic<uint64_t> g_smoke_lds{0};

}  // namespace

extern "C" size_t sdc_smoke_rollout_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 0) * WORK_BYTES; }

extern "C" int sdc_smoke_rollout(const float* c1, const float* c2, int64_t ctrl_b_stride, int64_t ctrl_f_stride,
                                 const float* init_density, int64_t dens_b_stride, const float* init_velocity,
                                 int64_t vel_b_stride, const unsigned char* fluid_mask, const unsigned char* bucket_labels,
                                 const unsigned char* safe_labels, int n_buckets, int n_safe, double* out, double* out_zero,
                                 void* work, size_t work_bytes, int B, int nt, int nx, int per_timelength, int ring_lo,
                                 int ring_hi, double accuracy, int max_iterations, void* stream) {
    SDC_REQUIRE(c1 && c2 && init_density && init_velocity && fluid_mask && bucket_labels && safe_labels && out && work,
                SDC_ENULL, "sdc_smoke_rollout: null pointer");
    SDC_REQUIRE(B > 0 && nt > 0 && nx > 0 && per_timelength > 0, SDC_EINVAL, "sdc_smoke_rollout: bad sizes");
    SDC_REQUIRE(SS % nx == 0, SDC_EINVAL, "sdc_smoke_rollout: nx = %d does not divide 128 (evaluate_solver.py:228)", nx);
    SDC_REQUIRE(per_timelength % nt == 0, SDC_EINVAL, "sdc_smoke_rollout: per_timelength = %d is not a multiple of nt = %d",
                per_timelength, nt);
    SDC_REQUIRE(n_buckets >= 2 && n_buckets <= MAXB && n_safe >= 1 && n_safe <= MAXB, SDC_EINVAL,
                "sdc_smoke_rollout: bucket counts %d / %d outside [2, %d] / [1, %d]", n_buckets, n_safe, MAXB, MAXB);
    SDC_REQUIRE(ring_lo >= 0 && ring_lo <= ring_hi && ring_hi <= SS, SDC_EINVAL, "sdc_smoke_rollout: bad control ring");
    SDC_REQUIRE(max_iterations >= 0, SDC_EINVAL, "sdc_smoke_rollout: max_iterations < 0");
    SDC_REQUIRE(work_bytes >= (size_t)B * WORK_BYTES, SDC_EINVAL, "sdc_smoke_rollout: workspace %zu < %zu bytes", work_bytes,
                (size_t)B * WORK_BYTES);
    SDC_REQUIRE((reinterpret_cast<uintptr_t>(work) & 15) == 0 && (reinterpret_cast<uintptr_t>(init_velocity) & 7) == 0 &&
                    (vel_b_stride & 1) == 0,
                SDC_EALIGN, "sdc_smoke_rollout: workspace must be 16-byte, init_velocity 8-byte aligned");
    const int lds = (SS + 1) * LP * (int)sizeof(double);
    SDC_LDS_OPTIN(g_smoke_lds, smoke_rollout_kernel, lds, "sdc_smoke_rollout");
    SmokeArgs a;
    a.c1 = c1; a.c2 = c2; a.c_sb = ctrl_b_stride; a.c_sf = ctrl_f_stride;
    a.dens0 = init_density; a.d_sb = dens_b_stride;
    a.vel0 = init_velocity; a.v_sb = vel_b_stride;
    a.fluid = fluid_mask; a.label_n = bucket_labels; a.label_s = safe_labels;
    a.out = out; a.out_zero = out_zero; a.work = static_cast<char*>(work);
    a.nt = nt; a.nx = nx; a.T = per_timelength; a.ti = per_timelength / nt; a.si = SS / nx;
    a.lo = ring_lo; a.hi = ring_hi; a.max_iter = max_iterations; a.nb_n = n_buckets; a.nb_s = n_safe;
    a.accuracy = accuracy;
    hipLaunchKernelGGL(smoke_rollout_kernel, dim3(B), dim3(NTHR), lds, sdc::as_stream(stream), a);
    return sdc::check_launch("sdc_smoke_rollout");
}
